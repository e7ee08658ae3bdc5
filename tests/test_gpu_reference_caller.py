"""The drop-in boundary demonstrated, not described: the REFERENCE'S OWN converter Task layer — src/TC/src/TasksColorCvt.cpp and
MemoryInterfaces.cpp compiled unmodified from /root/reference into oracle/_ref/libtc_ref_hip.so (oracle/Makefile `ref_tc_hip`) —
runs on the MI355X with its CUDA driver calls served by the HIP runtime and every `nppi*_Ctx` it calls forwarded to libvpfhip's C
ABI (oracle/ref_shim_hip/npp_over_vpf.h).  Its ConvertSurface::Execute, driven the way PySurfaceConverter::Execute drives it
(src/PyNvCodec/src/PySurfaceConverter.cpp:50-74), must produce oracle-equal pixels for every converter and every context it
accepts.  The library is built in the build container (the GPU box has no /root/reference) and travels with the snapshot."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libtc_ref_hip.so")

NAMES = {1: "Y", 2: "RGB", 3: "NV12", 4: "YUV420", 5: "RGB_PLANAR", 6: "BGR", 7: "YCBCR", 8: "YUV444", 9: "RGB_32F", 10: "RGB_32F_PLANAR"}
PAIRS = [(3, 4), (4, 3), (3, 2), (3, 6), (2, 5), (5, 2), (5, 8), (1, 8), (4, 2), (2, 4), (2, 8), (6, 7), (2, 6), (6, 2), (4, 6), (8, 6), (8, 2),
         (6, 8), (3, 1), (2, 9), (2, 1), (9, 10)]   # the reference ctor's pairs (TasksColorCvt.cpp:1313-1360) minus P10 / P12 -> NV12


@pytest.fixture(scope="module")
def ref(capi):
    capi.lib()  # one HIP runtime in the process, libvpfhip mapped first
    if not os.path.exists(SO):
        pytest.fail(f"{SO} missing: build it in the build container with `make -C oracle ref_tc_hip` (needs /root/reference)")
    L = C.CDLL(SO)
    L.ref_hip_convert.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_size_t), C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    return L


def _nvc():
    sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
    import PyNvCodec as nvc

    return nvc


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{NAMES[p[0]]}-{NAMES[p[1]]}")
@pytest.mark.parametrize("size", [(848, 464), (64, 32)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_reference_convertsurface_runs_on_mi355x_with_oracle_equal_pixels(ref, oracle, capfd, pair, size):
    o, nvc = oracle, _nvc()
    nvc.SetExtendedColorspaces(False)
    (fi, fo), (w, h) = pair, size
    src = o.synth(fi, w, h, 4000 + 16 * fi + fo)
    flat = np.concatenate([p.reshape(-1).view(np.uint8) for p in src])
    n_ok = 0
    for ctx in [None] + [(cs, cr) for cs in range(3) for cr in range(3)]:
        out = np.zeros(sum(r * rb * np.dtype(dt).itemsize for r, rb, dt in o.plane_shapes(fo, w, h)), np.uint8)
        nbytes, calls, log = C.c_size_t(0), C.c_int(0), C.create_string_buffer(512)
        rc = ref.ref_hip_convert(fi, fo, w, h, *(ctx if ctx else (-1, -1)), flat.ctypes.data, flat.size, out.ctypes.data, out.size, C.byref(nbytes), log, 512,
                                 C.byref(calls))
        what = f"{NAMES[fi]}->{NAMES[fo]} {w}x{h} ctx {ctx}: rc {rc}, adapters [{log.value.decode()}]"
        assert rc in (0, 1), what
        cc = nvc.ColorspaceConversionContext(nvc.ColorSpace(ctx[0]), nvc.ColorRange(ctx[1])) if ctx else None
        mine = nvc.ConverterResolve(getattr(nvc.PixelFormat, NAMES[fi]), getattr(nvc.PixelFormat, NAMES[fo]), cc)
        if (fi, fo) == (2, 8) and mine is not None and tuple(mine)[1] == o.MPEG:
            # rgb_yuv444's MPEG branch calls nppiRGBToYCbCr_8u_C3R: a PACKED result written into plane 0 of a planar surface
            # (TasksColorCvt.cpp:758), a reference bug this repo does not replicate; the adapter refuses it
            assert rc == 0 and b"nppiRGBToYCbCr_8u_C3R_Ctx:not-forwarded" in log.value, what
            continue
        assert (rc == 1) == (mine is not None), what + f" | this repo's Task layer resolves {mine}"
        if rc != 1:
            continue
        assert nbytes.value == out.size, what
        st, want = o.convert(fi, fo, mine[0], mine[1], w, h, src, o.FP32)
        assert st == 0
        want = np.concatenate([p.reshape(-1).view(np.uint8) for p in want])
        if not np.array_equal(out, want):
            d = np.flatnonzero(out != want)
            raise AssertionError(f"{what}: {d.size} of {out.size} bytes differ from the oracle, first at {d[0]}: {out[d[0]]} vs {want[d[0]]}")
        if (fi, fo) not in ((1, 8), (3, 1)):          # y_yuv444 / nv12_y are plain copies in the reference (no NPP colour call)
            assert calls.value >= 1 and b":ok" in log.value, what
        n_ok += 1
    capfd.readouterr()  # the reference prints a diagnostic for every refused combination
    assert n_ok >= 1
