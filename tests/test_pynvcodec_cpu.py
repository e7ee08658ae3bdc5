"""CPU-side tests of the drop-in Python API (PyNvCodec) and the C++ Task layer underneath it: enum values,
surface geometry per pixel format, converter pair table, error behaviour.  Surfaces are backed by HOST memory
through the `_UseHostAllocator` test hook, so nothing here launches a kernel; the pixel results are covered by the
-m gpu tests.  Reference lines are cited per test."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
nvc = pytest.importorskip("PyNvCodec")


@pytest.fixture(autouse=True)
def host_alloc():
    nvc._UseHostAllocator(True)
    yield
    nvc._UseHostAllocator(False)


def test_enum_values_are_the_reference_abi():
    # src/TC/inc/MemoryInterfaces.hpp:30-61 (values are visible to Python through int())
    PF = nvc.PixelFormat
    want = dict(UNDEFINED=0, Y=1, RGB=2, NV12=3, YUV420=4, RGB_PLANAR=5, BGR=6, YCBCR=7, YUV444=8, RGB_32F=9,
                RGB_32F_PLANAR=10, YUV422=11, P10=12, P12=13, YUV444_10bit=14, YUV420_10bit=15)
    for k, v in want.items():
        assert int(getattr(PF, k)) == v and getattr(nvc, k) == getattr(PF, k)  # export_values(): module-level names
    assert [int(nvc.ColorSpace.BT_601), int(nvc.ColorSpace.BT_709), int(nvc.ColorSpace.UNSPEC)] == [0, 1, 2]
    assert [int(nvc.ColorRange.MPEG), int(nvc.ColorRange.JPEG), int(nvc.ColorRange.UDEF)] == [0, 1, 2]
    c = nvc.ColorspaceConversionContext()
    assert c.color_space == nvc.ColorSpace.UNSPEC and c.color_range == nvc.ColorRange.UDEF  # :67
    c = nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_709, nvc.ColorRange.JPEG)
    c.color_range = nvc.ColorRange.MPEG
    assert (c.color_space, c.color_range) == (nvc.ColorSpace.BT_709, nvc.ColorRange.MPEG)


W, H = 1920, 1080
GEOMETRY = {
    # fmt: (num_planes, [(width, height) per plane], host_size, raw plane-0 (width, height), elem)
    "Y": (1, [(W, H)], W * H, (W, H), 1),
    "NV12": (2, [(W, H), (W, H // 2)], W * H * 3 // 2, (W, H * 3 // 2), 1),            # MemoryInterfaces.cpp:817-820,862-873
    "YUV420": (3, [(W, H), (W // 2, H // 2), (W // 2, H // 2)], W * H * 3 // 2, (W, H), 1),  # :924-929
    "YCBCR": (3, [(W, H), (W // 2, H // 2), (W // 2, H // 2)], W * H * 3 // 2, (W, H), 1),
    "YUV422": (3, [(W, H), (W // 2, H), (W // 2, H)], W * H * 2, (W, H), 1),
    "RGB": (1, [(W, H)], W * H * 3, (3 * W, H), 1),                                         # :1367-1370
    "BGR": (1, [(W, H)], W * H * 3, (3 * W, H), 1),
    "RGB_PLANAR": (3, [(W, H)] * 3, W * H * 3, (W, 3 * H), 1),                              # :1530-1534
    "YUV444": (3, [(W, H)] * 3, W * H * 3, (W, 3 * H), 1),
    "RGB_32F": (1, [(W, H)], W * H * 12, (3 * W, H), 4),
    "RGB_32F_PLANAR": (3, [(W, H)] * 3, W * H * 12, (W, 3 * H), 4),
    "P10": (2, [(W, H), (W, H // 2)], W * H * 3, (W, H * 3 // 2), 2),
    "P12": (2, [(W, H), (W, H // 2)], W * H * 3, (W, H * 3 // 2), 2),
}


@pytest.mark.parametrize("name", sorted(GEOMETRY))
def test_surface_geometry(name):
    nplanes, dims, host, raw0, elem = GEOMETRY[name]
    s = nvc.Surface.Make(getattr(nvc.PixelFormat, name), W, H, context=0)
    assert not s.Empty() and s.OwnMemory() and s.Format() == getattr(nvc.PixelFormat, name)
    assert s.NumPlanes() == nplanes and s.HostSize() == host
    for p, (w, h) in enumerate(dims):
        assert (s.Width(p), s.Height(p)) == (w, h), p
        assert s.Pitch(p) % 256 == 0 and s.Pitch(p) >= w * elem  # rows are dwordx4-aligned for the gfx950 kernels
    with pytest.raises(ValueError):
        s.Width(nplanes)  # std::invalid_argument("Invalid plane number")
    p0 = s.PlanePtr()
    assert (p0.Width(), p0.Height(), p0.ElemSize()) == (*raw0, elem) and p0.GpuMem() != 0
    assert p0.HostFrameSize() == raw0[0] * raw0[1] * elem and p0.Pitch() == s.Pitch(0)
    assert "Width:" in repr(s) and name in repr(s) and "Pitch" in repr(p0)


def test_plane_pointers_follow_reference_layout():
    s = nvc.Surface.Make(nvc.PixelFormat.NV12, W, H, context=0)
    base = s.PlanePtr(0).GpuMem()
    assert s.PlanePtr(1).GpuMem() == base + H * s.Pitch()            # :889-896 PlanePtr(1) = base + Height*pitch
    assert (s.PlanePtr(1).Width(), s.PlanePtr(1).Height()) == (W, H // 2)
    s = nvc.Surface.Make(nvc.PixelFormat.RGB_PLANAR, W, H, context=0)
    base = s.PlanePtr(0).GpuMem()
    assert [s.PlanePtr(i).GpuMem() for i in range(3)] == [base + i * H * s.Pitch() for i in range(3)]  # :1593-1600
    s = nvc.Surface.Make(nvc.PixelFormat.YUV420, W, H, context=0)
    ptrs = [s.PlanePtr(i).GpuMem() for i in range(3)]
    assert len(set(ptrs)) == 3 and s.Pitch(1) == s.Pitch(2) and s.Pitch(1) != s.Pitch(0)  # three separate allocations


def test_odd_sizes_round_chroma_up():
    s = nvc.Surface.Make(nvc.PixelFormat.YUV420, 7, 5, context=0)
    assert (s.Width(1), s.Height(1)) == (4, 3) and s.HostSize() == 35 + 2 * 12
    s = nvc.Surface.Make(nvc.PixelFormat.NV12, 7, 5, context=0)
    assert (s.Height(0), s.Height(1)) == (5, 3) and s.PlanePtr(0).Height() == 8


def test_unsupported_surface_formats():
    with pytest.raises(ValueError):
        nvc.Surface.Make(nvc.PixelFormat.UNDEFINED, 16, 16, context=0)


REFERENCE_PAIRS = [  # ConvertSurface ctor, src/TC/src/TasksColorCvt.cpp:1313-1360
    ("NV12", "YUV420"), ("YUV420", "NV12"), ("P10", "NV12"), ("P12", "NV12"), ("NV12", "RGB"), ("NV12", "BGR"),
    ("RGB", "RGB_PLANAR"), ("RGB_PLANAR", "RGB"), ("RGB_PLANAR", "YUV444"), ("Y", "YUV444"), ("YUV420", "RGB"),
    ("RGB", "YUV420"), ("RGB", "YUV444"), ("BGR", "YCBCR"), ("RGB", "BGR"), ("BGR", "RGB"), ("YUV420", "BGR"),
    ("YUV444", "BGR"), ("YUV444", "RGB"), ("BGR", "YUV444"), ("NV12", "Y"), ("RGB", "RGB_32F"), ("RGB", "Y"),
    ("RGB_32F", "RGB_32F_PLANAR")]


def test_converter_pair_table():
    PF = nvc.PixelFormat
    for s, d in REFERENCE_PAIRS:
        assert nvc.ConverterPairSupport(getattr(PF, s), getattr(PF, d)) == 1, (s, d)
        c = nvc.PySurfaceConverter(64, 32, getattr(PF, s), getattr(PF, d), 0, 0)  # (context, stream) overload
        assert c.Format() == getattr(PF, d)
    assert nvc.ConverterPairSupport(PF.NV12, PF.RGB_PLANAR) == 2  # additive: fused nv12_rgb + rgb8_deinterleave
    for s, d in [("Y", "RGB"), ("NV12", "NV12"), ("YUV444", "NV12"), ("RGB_32F", "RGB"), ("YCBCR", "RGB"), ("NV12", "YUV444")]:
        assert nvc.ConverterPairSupport(getattr(PF, s), getattr(PF, d)) == 0
        with pytest.raises(ValueError, match="Unsupported pixel format conversion"):  # :1361-1366 invalid_argument
            nvc.PySurfaceConverter(64, 32, getattr(PF, s), getattr(PF, d), 0, 0)


def test_fused_convert_resizer_pairs_and_host_side_checks():
    """additive PySurfaceConvertResizer: only NV12 / YUV420 -> RGB / BGR / RGB_PLANAR are fusable; without a GPU the launch
    fails and the result is an Empty() surface (no CPU fallback)"""
    PF = nvc.PixelFormat
    for s in ("NV12", "YUV420"):
        for d in ("RGB", "BGR", "RGB_PLANAR"):
            f = nvc.PySurfaceConvertResizer(640, 360, getattr(PF, s), 224, 224, getattr(PF, d), 0, 0)
            assert f.Format() == getattr(PF, d)
    for s, d in [("RGB", "RGB_PLANAR"), ("NV12", "YUV420"), ("YUV444", "RGB"), ("NV12", "Y")]:
        with pytest.raises(ValueError, match="Unsupported fused conversion"):
            nvc.PySurfaceConvertResizer(640, 360, getattr(PF, s), 224, 224, getattr(PF, d), 0, 0)
    with pytest.raises(ValueError):
        nvc.PySurfaceConvertResizer(640, 360, PF.NV12, 0, 224, PF.RGB, 0, 0)
    f = nvc.PySurfaceConvertResizer(64, 32, PF.NV12, 32, 16, PF.RGB, 0, 0)
    assert f.Execute(None, None).Empty()
    assert f.Execute(nvc.Surface.Make(PF.NV12, 32, 32, context=0), None).Empty()        # wrong size
    assert f.Execute(nvc.Surface.Make(PF.YUV420, 64, 32, context=0), None).Empty()      # wrong format
    assert not f.ExecuteBatch([], [], None)
    if nvc.GetNumGpus() == 0:
        cc = nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_709, nvc.ColorRange.MPEG)
        assert f.Execute(nvc.Surface.Make(PF.NV12, 64, 32, context=0), cc).Empty()


def test_converter_failure_is_an_empty_surface_not_a_cpu_fallback(capfd):
    """no GPU here: the HIP launch fails, Execute returns an Empty() surface of the output format
    (PySurfaceConverter.cpp:54-73) — and nothing computes the pixels on the CPU instead"""
    PF = nvc.PixelFormat
    conv = nvc.PySurfaceConverter(64, 32, PF.NV12, PF.RGB, 0, 0)
    src = nvc.Surface.Make(PF.NV12, 64, 32, context=0)
    if nvc.GetNumGpus() > 0:
        pytest.skip("GPU present: covered by the gpu tests")
    out = conv.Execute(src, nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_709, nvc.ColorRange.MPEG))
    assert out.Empty() and out.Format() == PF.RGB
    assert "Failed to convert surface" in capfd.readouterr().err
    assert conv.Execute(None, None).Empty()                                  # null input -> empty surface (:54-56)
    wrong = nvc.Surface.Make(PF.NV12, 32, 32, context=0)
    assert conv.Execute(wrong, None).Empty()                                 # size mismatch -> empty


def test_nv12_rgb_601_mpeg_is_rejected_like_the_reference(capfd):
    PF = nvc.PixelFormat
    conv = nvc.PySurfaceConverter(64, 32, PF.NV12, PF.RGB, 0, 0)
    src = nvc.Surface.Make(PF.NV12, 64, 32, context=0)
    out = conv.Execute(src, nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_601, nvc.ColorRange.MPEG))
    assert out.Empty()
    assert "Rec. 601 NV12 -> RGB MPEG range conversion isn't supported yet." in capfd.readouterr().err  # :156-163
    out = conv.Execute(src, None)  # default context = BT_601 + MPEG (:67-68) -> same rejection
    assert out.Empty() and "isn't supported yet" in capfd.readouterr().err
    out = conv.Execute(src, nvc.ColorspaceConversionContext())  # UNSPEC -> "unsupported color space" (:166-168)
    assert out.Empty() and "unsupported color space" in capfd.readouterr().err
    y2r = nvc.PySurfaceConverter(64, 32, PF.YUV420, PF.RGB, 0, 0)
    out = y2r.Execute(nvc.Surface.Make(PF.YUV420, 64, 32, context=0),
                      nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_709, nvc.ColorRange.MPEG))
    assert out.Empty() and "Rec.709 YUV -> RGB conversion isn't supported yet." in capfd.readouterr().err  # :346-348
    y4 = nvc.PySurfaceConverter(64, 32, PF.YUV444, PF.RGB, 0, 0)
    out = y4.Execute(nvc.Surface.Make(PF.YUV444, 64, 32, context=0), None)  # default MPEG: yuv444_rgb only has JPEG (:534-541)
    assert out.Empty() and "unsupported color range" in capfd.readouterr().err


def test_resizer_and_remaper_argument_errors():
    PF = nvc.PixelFormat
    for bad in (PF.P10, PF.YUV422, PF.UNDEFINED):
        with pytest.raises(RuntimeError):
            nvc.PySurfaceResizer(64, 32, bad, 0, 0)  # Tasks.cpp:1470-1475
    for ok in (PF.RGB, PF.BGR, PF.YUV420, PF.YCBCR, PF.YUV444, PF.RGB_PLANAR, PF.RGB_32F, PF.RGB_32F_PLANAR, PF.NV12):
        assert nvc.PySurfaceResizer(64, 32, ok, 0, 0).Format() == ok  # the reference ctor's list (:1458-1469)
    rs = nvc.PySurfaceResizer(64, 32, PF.RGB, 0, 0)
    assert rs.Format() == PF.RGB
    assert rs.Execute(nvc.Surface.Make(PF.BGR, 128, 64, context=0)).Empty()  # format mismatch -> TASK_EXEC_FAIL (:1166-1168)
    assert rs.Execute(None).Empty()
    # additive knobs: defaults are the reference's behaviour (blocking Run, the Lanczos filter it asks NPP for); ExecuteBatch validates before it launches
    assert rs.GetAsync() is False and rs.GetInterpolation() == 2
    rs.SetAsync(True); rs.SetInterpolation(1)
    assert rs.GetAsync() is True and rs.GetInterpolation() == 1
    a = [nvc.Surface.Make(PF.RGB, 128, 64, context=0) for _ in range(2)]
    assert not rs.ExecuteBatch(a, [nvc.Surface.Make(PF.RGB, 64, 32, context=0)])                       # length mismatch
    assert not rs.ExecuteBatch(a, [nvc.Surface.Make(PF.RGB, 66, 32, context=0) for _ in range(2)])    # wrong destination size
    assert not rs.ExecuteBatch(a, [nvc.Surface.Make(PF.BGR, 64, 32, context=0) for _ in range(2)])    # wrong destination format
    xm = np.zeros((8, 8), np.float32)
    with pytest.raises(RuntimeError):
        nvc.PySurfaceRemaper(xm, xm, PF.NV12, 0, 0)  # :1615-1620
    with pytest.raises(RuntimeError):
        nvc.PySurfaceRemaper(xm, np.zeros((4, 8), np.float32), PF.RGB, 0, 0)


def test_gpu_id_overloads_fail_loudly_without_that_gpu():
    n = nvc.GetNumGpus()
    with pytest.raises(RuntimeError, match="GPU ordinal out of range"):
        nvc.Surface.Make(nvc.PixelFormat.NV12, 64, 64, n + 3)
    with pytest.raises(RuntimeError):
        nvc.PySurfaceConverter(64, 64, nvc.PixelFormat.NV12, nvc.PixelFormat.RGB, n + 3)


def test_nvdec_nvenc_placeholders_explain_themselves():
    for cls in (nvc.PyNvDecoder, nvc.PyNvEncoder, nvc.PyFFmpegDemuxer, nvc.PyFfmpegDecoder):
        with pytest.raises(RuntimeError, match="not available on MI355X"):
            cls("x.mp4", 0)
