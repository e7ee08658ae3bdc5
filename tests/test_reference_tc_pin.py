"""Reference pin for the host side of the path: the REFERENCE'S OWN Surface classes (src/TC/src/MemoryInterfaces.cpp) and
converter dispatch (src/TC/src/TasksColorCvt.cpp) are compiled from /root/reference by oracle/Makefile (`ref_tc`) against
stand-in CUDA / NPP headers (oracle/ref_shim: "device" memory is host memory, NPP entry points record their names) into
oracle/_ref/libtc_ref.so.  These tests drive that library and this repo's PyNvCodec side by side:

  * surface geometry of every pixel format (planes, per-plane width / height / bytes, host frame size, plane offsets);
  * for every (src, dst) format pair: does the converter exist; for every ColorspaceConversionContext (and none at all):
    is the combination accepted, and WHICH NPP function — i.e. which colour model — does the reference call.

What stays unpinned is NPP's arithmetic (closed source): the table NPP-function -> (matrix, range) below is SURVEY §8c's
reading of NVIDIA's public NPP documentation.
"""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libtc_ref.so")
if not os.path.exists(LIB):
    pytest.skip("oracle/_ref/libtc_ref.so not built (needs /root/reference: make -C oracle ref_tc)", allow_module_level=True)
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
nvc = pytest.importorskip("PyNvCodec")

REF = C.CDLL(LIB)
REF.ref_convert_probe.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
REF.ref_surface_geometry.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_int64), C.c_int]
NAMES = "UNDEFINED Y RGB NV12 YUV420 RGB_PLANAR BGR YCBCR YUV444 RGB_32F RGB_32F_PLANAR YUV422 P10 P12 YUV444_10bit YUV420_10bit".split()
PF = nvc.PixelFormat
BT601, BT709, MPEG, JPEG = 0, 1, 0, 1

# NPP entry point -> colour model, per NVIDIA's NPP documentation (SURVEY §8c [A1][A4][A5]): the "YCbCr" family is the
# limited-range (MPEG) BT.601 model, the "YUV" family the full-range one, 709CSC limited / 709HDTV full range BT.709.
MODEL = {
    "nppiNV12ToRGB_8u_P2C3R_Ctx": (BT601, JPEG), "nppiNV12ToBGR_8u_P2C3R_Ctx": (BT601, JPEG),
    "nppiNV12ToRGB_709CSC_8u_P2C3R_Ctx": (BT709, MPEG), "nppiNV12ToBGR_709CSC_8u_P2C3R_Ctx": (BT709, MPEG),
    "nppiNV12ToRGB_709HDTV_8u_P2C3R_Ctx": (BT709, JPEG), "nppiNV12ToBGR_709HDTV_8u_P2C3R_Ctx": (BT709, JPEG),
    "nppiYCbCr420ToRGB_8u_P3C3R_Ctx": (BT601, MPEG), "nppiYCbCr420ToBGR_8u_P3C3R_Ctx": (BT601, MPEG),
    "nppiYUV420ToRGB_8u_P3C3R_Ctx": (BT601, JPEG), "nppiYUV420ToBGR_8u_P3C3R_Ctx": (BT601, JPEG),
    "nppiYCbCrToBGR_8u_P3C3R_Ctx": (BT601, MPEG), "nppiYUVToBGR_8u_P3C3R_Ctx": (BT601, JPEG), "nppiYUVToRGB_8u_P3C3R_Ctx": (BT601, JPEG),
    "nppiRGBToYCbCr420_8u_C3P3R_Ctx": (BT601, MPEG), "nppiRGBToYUV420_8u_C3P3R_Ctx": (BT601, JPEG),
    "nppiRGBToYCbCr_8u_C3R_Ctx": (BT601, MPEG), "nppiRGBToYUV_8u_C3P3R_Ctx": (BT601, JPEG),
    "nppiRGBToYCbCr_8u_P3R_Ctx": (BT601, MPEG), "nppiRGBToYUV_8u_P3R_Ctx": (BT601, JPEG),
    "nppiBGRToYCbCr_8u_C3P3R_Ctx": (BT601, MPEG), "nppiBGRToYUV_8u_C3P3R_Ctx": (BT601, JPEG),
    "nppiBGRToYCbCr420_8u_C3P3R_Ctx": (BT601, MPEG),
}
RELAYOUT = {  # no arithmetic on the samples: only accept / refuse is compared
    "nppiYCbCr420_8u_P2P3R_Ctx", "nppiNV12ToYUV420_8u_P2P3R_Ctx", "nppiYCbCr420_8u_P3P2R_Ctx", "nppiCopy_8u_C3P3R_Ctx",
    "nppiCopy_8u_P3C3R_Ctx", "nppiSwapChannels_8u_C3R_Ctx", "nppiSet_8u_C1R_Ctx", "nppiCopy_8u_C1R_Ctx", "nppiScale_8u32f_C3R_Ctx",
    "nppiCopy_32f_C3P3R_Ctx", "nppiDivC_16u_C1RSfs_Ctx", "nppiConvert_16u8u_C1R_Ctx", "nppiRGBToGray_8u_C3C1R_Ctx", ""}


def ref_probe(i, o, cs, cr):
    buf, seen = C.create_string_buffer(1024), C.c_int(-1)
    r = REF.ref_convert_probe(i, o, 64, 32, cs, cr, buf, 1024, C.byref(seen))
    return r, buf.value.decode(), seen.value


def test_converter_pairs_and_colour_model_selection_match_the_reference(capfd):
    nvc.SetExtendedColorspaces(False)
    n_pairs = n_cases = 0
    for i in range(1, 14):
        for o in range(1, 14):
            fi, fo = getattr(PF, NAMES[i]), getattr(PF, NAMES[o])
            r0, log0, _ = ref_probe(i, o, -1, -1)
            ours = nvc.ConverterPairSupport(fi, fo)
            assert (r0 != -1) == (ours == 1), f"{NAMES[i]}->{NAMES[o]}: reference ctor {'ok' if r0 != -1 else 'throws'}, ours level {ours}"
            if r0 == -1:
                assert "Unsupported pixel format conversion" in log0  # TasksColorCvt.cpp:1361-1366
                continue
            n_pairs += 1
            ctxs = [None] + [(cs, cr) for cs in range(3) for cr in range(3)]
            for c in ctxs:
                r, log, seen = ref_probe(i, o, *(c if c else (-1, -1)))
                cc = nvc.ColorspaceConversionContext(nvc.ColorSpace(c[0]), nvc.ColorRange(c[1])) if c else None
                mine = nvc.ConverterResolve(fi, fo, cc)
                what = f"{NAMES[i]}->{NAMES[o]} ctx {c}: reference {'accepts' if r == 1 else 'refuses'} ({log}), ours {mine}"
                assert (r == 1) == (mine is not None), what
                n_cases += 1
                if r != 1:
                    continue
                assert seen == o, what
                calls = set(log.split(","))
                models = {MODEL[f] for f in calls if f in MODEL}
                assert calls <= set(MODEL) | RELAYOUT, f"unknown NPP entry point in {log}"
                if models:
                    assert len(models) == 1 and tuple(mine) == models.pop(), what
    capfd.readouterr()  # both sides print the reference's diagnostics for refused combinations
    assert n_pairs == 24 and n_cases == 240  # the 24 converters of the reference ctor (TasksColorCvt.cpp:1313-1360)


@pytest.mark.parametrize("w,h", [(1920, 1080), (848, 464), (64, 32), (3840, 2160)])
def test_surface_geometry_matches_the_reference(w, h):
    nvc._UseHostAllocator(True)
    try:
        out = (C.c_int64 * 64)()
        for f in range(1, 16):
            name = NAMES[f]
            assert REF.ref_surface_geometry(f, w, h, out, 64) == 0, name
            s = nvc.Surface.Make(getattr(PF, name), w, h, context=0)
            n = int(out[0])
            assert s.NumPlanes() == n, name
            # reference quirks NOT replicated (asserted so that they stay documented):
            #   SurfaceP10 is a SurfaceNV12 with ElemSize() overridden to 2 (MemoryInterfaces.hpp:539-541): its own
            #   allocation is 8-bit sized; ours holds 16-bit samples.  SurfaceYUV444_10bit allocates 16-bit planes
            #   (MemoryInterfaces.cpp:1841-1845) but inherits ElemSize() == 1; ours reports 2.
            if name == "P10":
                assert out[1] == w * h * 3 // 2 and out[2] == 2 and s.HostSize() == w * h * 3
            elif name == "YUV444_10bit":
                assert out[2] == 1 and s.PlanePtr(0).ElemSize() == 2 and s.HostSize() == out[1]
            else:
                assert s.HostSize() == out[1], (name, s.HostSize(), out[1])
                assert s.PlanePtr(0).ElemSize() == out[2], name
            for p in range(n):
                rw, rh, rpitch, rwb, roff, pw, ph = (int(v) for v in out[3 + 7 * p:10 + 7 * p])
                assert (s.Width(p), s.Height(p)) == (rw, rh), (name, p)
                # plane p of a single-allocation format starts a whole number of rows after plane 0 in both libraries
                if roff and roff % rpitch == 0 and name not in ("YUV420", "YCBCR", "YUV422"):
                    mine = s.PlanePtr(p).GpuMem() - s.PlanePtr(0).GpuMem()
                    assert mine % s.Pitch(0) == 0 and mine // s.Pitch(0) == roff // rpitch, (name, p)
                if p == 0:  # raw plane object: width in elements of the plane (RGB: 3W), height of the whole allocation
                    assert (s.PlanePtr(0).Width(), s.PlanePtr(0).Height()) == (pw, ph), (name, "raw plane 0")
        for f in (16, 17):  # NV12_PLANAR, GRAY12: no Surface class in the reference's factory
            assert REF.ref_surface_geometry(f, w, h, out, 64) == -1
    finally:
        nvc._UseHostAllocator(False)


def test_odd_sizes_are_a_documented_divergence():
    """odd widths / heights: the reference truncates (a 7 x 5 NV12 surface reports Height(0) == 4 because it derives the luma
    height from the allocation's 7 rows as 7 * 2 / 3, and 4:2:0 chroma planes are 3 x 2, leaving the last column and row of
    the picture without chroma); this repo rounds chroma up so every luma sample has a chroma sample"""
    nvc._UseHostAllocator(True)
    try:
        out = (C.c_int64 * 64)()
        assert REF.ref_surface_geometry(3, 7, 5, out, 64) == 0      # NV12
        assert (out[3], out[4], out[10], out[11]) == (7, 4, 7, 2) and out[1] == 49
        s = nvc.Surface.Make(PF.NV12, 7, 5, context=0)
        assert (s.Width(0), s.Height(0), s.Width(1), s.Height(1)) == (7, 5, 7, 3)
        assert REF.ref_surface_geometry(4, 7, 5, out, 64) == 0      # YUV420
        assert (out[10], out[11]) == (3, 2) and out[1] == 35 + 2 * 6
        s = nvc.Surface.Make(PF.YUV420, 7, 5, context=0)
        assert (s.Width(1), s.Height(1)) == (4, 3) and s.HostSize() == 35 + 2 * 12
    finally:
        nvc._UseHostAllocator(False)
