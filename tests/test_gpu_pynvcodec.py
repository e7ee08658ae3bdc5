"""-m gpu tests of the drop-in Python API: the chains user code actually builds with the reference
(samples/*.py), checked against the CPU oracle.  Everything goes PyNvCodec -> C++ Task layer ->
C ABI -> HIP kernels."""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no GPU visible", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
import PyNvCodec as nvc  # noqa: E402
import PytorchNvCodec as pnvc  # noqa: E402

PF, CS, CR = nvc.PixelFormat, nvc.ColorSpace, nvc.ColorRange
GPU = 0


def host_frame(planes):
    return np.concatenate([p.reshape(-1).view(np.uint8) for p in planes])


def upload(fmt, w, h, planes):
    up = nvc.PyFrameUploader(w, h, fmt, GPU)
    return up.UploadSingleFrame(host_frame(planes)).Clone(GPU)  # Clone: the uploader's surface is recycled


def download(surf, dtype=np.uint8):
    dl = nvc.PySurfaceDownloader(surf.Width(), surf.Height(), surf.Format(), GPU)
    out = np.zeros(1, dtype)
    assert dl.DownloadSingleSurface(surf, out)
    return out


def test_upload_download_roundtrip(oracle):
    for name in ("NV12", "YUV420", "RGB", "BGR", "RGB_PLANAR", "YUV444", "Y", "YCBCR", "YUV422"):
        fmt = getattr(PF, name)
        w, h = 322, 146
        s = nvc.Surface.Make(fmt, w, h, GPU)
        frame = np.random.default_rng(1).integers(0, 256, s.HostSize(), dtype=np.uint8)
        up = nvc.PyFrameUploader(w, h, fmt, GPU)
        surf = up.UploadSingleFrame(frame)
        assert surf.Format() == fmt and not surf.Empty() and not surf.OwnMemory()  # alias of the uploader's surface
        assert np.array_equal(download(surf), frame), name
    f32 = np.random.default_rng(2).random(64 * 32 * 3, dtype=np.float32)
    s = nvc.PyFrameUploader(64, 32, PF.RGB_32F, GPU).UploadSingleFrame(f32)
    assert np.array_equal(download(s, np.float32), f32)
    u16 = np.random.default_rng(3).integers(0, 65536, 64 * 32 * 3 // 2, dtype=np.uint16)
    s = nvc.PyFrameUploader(64, 32, PF.P10, GPU).UploadSingleFrame(u16)
    assert np.array_equal(download(s, np.uint16), u16)


def test_chain_segmentation_sample(oracle):
    """samples/SampleTorchSegmentation.py:207-253 — NV12 -> RGB -> RGB_PLANAR -> torch tensor, BT.709 + JPEG"""
    w, h = 848, 464  # the reference's test clip resolution (tests/test_PySurface.py:55-64)
    src = oracle.synth(oracle.NV12, w, h, 11, "B")
    nv12 = upload(PF.NV12, w, h, src)
    to_rgb = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, GPU)
    to_pln = nvc.PySurfaceConverter(w, h, PF.RGB, PF.RGB_PLANAR, GPU)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.JPEG)
    rgb = to_rgb.Execute(nv12, cc)
    pln = to_pln.Execute(rgb, cc)
    assert not rgb.Empty() and not pln.Empty() and pln.Format() == PF.RGB_PLANAR
    _, want_rgb = oracle.convert(oracle.NV12, oracle.RGB, 1, 1, w, h, src)
    _, want_pln = oracle.convert(oracle.RGB, oracle.RGB_PLANAR, 1, 1, w, h, want_rgb)
    assert np.array_equal(download(pln), host_frame(want_pln))
    # the sample's tensor hand-off: plane = W x 3H, then resize_(3, H, W)
    p = pln.PlanePtr()
    t = pnvc.makefromDevicePtrUint8(p.GpuMem(), p.Width(), p.Height(), p.Pitch(), p.ElemSize())
    assert t.shape == (3 * h, w) and t.dtype == torch.uint8 and t.is_contiguous()
    t.resize_(3, h, w)
    assert np.array_equal(t.cpu().numpy(), np.stack(want_pln))
    # zero-copy view of the same surface: same pixels, no copy (data_ptr inside the surface allocation)
    v = pnvc.view_surface_planar(pln)
    assert v.shape == (3, h, w) and v.data_ptr() == p.GpuMem() and np.array_equal(v.cpu().numpy(), np.stack(want_pln))
    # and the additive fused converter gives the identical planar picture in one pass
    fused = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB_PLANAR, GPU).Execute(nv12, cc)
    assert np.array_equal(download(fused), host_frame(want_pln))


def test_chain_resnet_sample(oracle):
    """samples/SampleTorchResnet.py:1073-1138 — NV12 -> YUV420 -> Resize(224x224) -> RGB -> RGB_PLANAR, BT.601 + MPEG
    (goes through YUV420 because nv12_rgb rejects 601+MPEG)"""
    w, h, tw, th = 848, 464, 224, 224
    src = oracle.synth(oracle.NV12, w, h, 12, "B")
    cc = nvc.ColorspaceConversionContext(CS.BT_601, CR.MPEG)
    nv12 = upload(PF.NV12, w, h, src)
    yuv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.YUV420, GPU).Execute(nv12, cc)
    small = nvc.PySurfaceResizer(tw, th, PF.YUV420, GPU).Execute(yuv)
    rgb = nvc.PySurfaceConverter(tw, th, PF.YUV420, PF.RGB, GPU).Execute(small, cc)
    pln = nvc.PySurfaceConverter(tw, th, PF.RGB, PF.RGB_PLANAR, GPU).Execute(rgb, cc)
    assert (small.Width(), small.Height()) == (tw, th) and not pln.Empty()
    _, a = oracle.convert(oracle.NV12, oracle.YUV420, 0, 0, w, h, src)
    _, b = oracle.resize(oracle.YUV420, oracle.LANCZOS3, w, h, a, tw, th)  # the resizer's default is the reference's filter
    _, c = oracle.convert(oracle.YUV420, oracle.RGB, 0, 0, tw, th, b)
    _, d = oracle.convert(oracle.RGB, oracle.RGB_PLANAR, 0, 0, tw, th, c)
    assert np.array_equal(download(pln), host_frame(d))
    # the direct route is refused exactly like the reference refuses it (TasksColorCvt.cpp:156-163) ...
    direct = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, GPU)
    assert direct.Execute(nv12, cc).Empty()
    # ... unless the caller opts into the superset the kernels implement
    nvc.SetExtendedColorspaces(True)
    try:
        out = direct.Execute(nv12, cc)
        _, want = oracle.convert(oracle.NV12, oracle.RGB, 0, 0, w, h, src)
        assert not out.Empty() and np.array_equal(download(out), want[0].reshape(-1))
    finally:
        nvc.SetExtendedColorspaces(False)


def test_fused_convert_resizer_equals_the_two_step_chain(oracle):
    """additive PySurfaceConvertResizer: NV12 -> (bilinear) -> RGB_PLANAR in one pass must equal PySurfaceConverter followed
    by PySurfaceResizer bit for bit (and the oracle's convert-then-resize); same refusals as the unfused converter"""
    w, h, tw, th = 1280, 720, 416, 234
    src = oracle.synth(oracle.NV12, w, h, 14)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
    nv12 = upload(PF.NV12, w, h, src)
    for fmt, ofmt in ((PF.RGB, oracle.RGB), (PF.RGB_PLANAR, oracle.RGB_PLANAR)):
        fused = nvc.PySurfaceConvertResizer(w, h, PF.NV12, tw, th, fmt, GPU)
        out = fused.Execute(nv12, cc)
        assert not out.Empty() and (out.Width(), out.Height(), out.Format()) == (tw, th, fmt) and fused.Format() == fmt
        conv, rs = nvc.PySurfaceConverter(w, h, PF.NV12, fmt, GPU), nvc.PySurfaceResizer(tw, th, fmt, GPU)
        rs.SetInterpolation(1)  # the fused task is the bilinear chain (the resizer alone defaults to Lanczos like the reference)
        two_step = rs.Execute(conv.Execute(nv12, cc))
        assert np.array_equal(download(out), download(two_step))
        _, want = oracle.convert_resize(oracle.NV12, ofmt, 1, 0, w, h, src, tw, th)
        assert np.array_equal(download(out), host_frame(want))
        # batch: caller-owned destinations, one dispatch
        dsts = [nvc.Surface.Make(fmt, tw, th, GPU) for _ in range(3)]
        assert fused.ExecuteBatch([nv12] * 3, dsts, cc)
        torch.cuda.synchronize()
        for d in dsts:
            assert np.array_equal(download(d), host_frame(want))
        assert not fused.ExecuteBatch([nv12], [nvc.Surface.Make(fmt, tw + 2, th, GPU)], cc)  # wrong destination size
    # refusals: the reference's nv12_rgb rejects BT.601 + MPEG; wrong input size / format -> Empty()
    fused = nvc.PySurfaceConvertResizer(w, h, PF.NV12, tw, th, PF.RGB, GPU)
    assert fused.Execute(nv12, nvc.ColorspaceConversionContext(CS.BT_601, CR.MPEG)).Empty()
    assert fused.Execute(upload(PF.NV12, 640, 360, oracle.synth(oracle.NV12, 640, 360, 15)), cc).Empty()
    assert fused.Execute(None, cc).Empty()
    with pytest.raises(ValueError):
        nvc.PySurfaceConvertResizer(w, h, PF.RGB, tw, th, PF.RGB_PLANAR, GPU)  # not a fusable pair


def test_resizer_accepts_float_surfaces_like_the_reference(oracle):
    """ResizeSurface accepts RGB_32F and RGB_32F_PLANAR (Tasks.cpp:1458-1476): RGB -> RGB_32F -> resize -> RGB_32F_PLANAR"""
    w, h, tw, th = 640, 360, 224, 126
    src = oracle.synth(oracle.RGB, w, h, 16)
    rgb = upload(PF.RGB, w, h, src)
    f32 = nvc.PySurfaceConverter(w, h, PF.RGB, PF.RGB_32F, GPU).Execute(rgb, None)
    small = nvc.PySurfaceResizer(tw, th, PF.RGB_32F, GPU).Execute(f32)
    pln = nvc.PySurfaceConverter(tw, th, PF.RGB_32F, PF.RGB_32F_PLANAR, GPU).Execute(small, None)
    small_pln = nvc.PySurfaceResizer(tw // 2, th // 2, PF.RGB_32F_PLANAR, GPU).Execute(pln)
    assert (small.Width(), small.Height(), small.Format()) == (tw, th, PF.RGB_32F) and not small_pln.Empty()
    _, a = oracle.convert(oracle.RGB, oracle.RGB_32F, 0, 0, w, h, src)
    _, b = oracle.resize(oracle.RGB_32F, oracle.LANCZOS3, w, h, a, tw, th)
    _, c = oracle.convert(oracle.RGB_32F, oracle.RGB_32F_PLANAR, 0, 0, tw, th, b)
    _, d = oracle.resize(oracle.RGB_32F_PLANAR, oracle.LANCZOS3, tw, th, c, tw // 2, th // 2)
    assert np.array_equal(download(small, np.float32), host_frame(b).view(np.float32))
    assert np.array_equal(download(small_pln, np.float32), host_frame(d).view(np.float32))


def test_output_reuse_hint(oracle):
    """additive PySurfaceConverter.SetOutputReuseHint: same pixels, the flag sticks"""
    w, h = 1280, 720
    src = oracle.synth(oracle.NV12, w, h, 17)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
    conv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, GPU)
    assert conv.GetOutputReuseHint() is False
    a = download(conv.Execute(upload(PF.NV12, w, h, src), cc)).copy()
    conv.SetOutputReuseHint(True)
    assert conv.GetOutputReuseHint() is True
    b = download(conv.Execute(upload(PF.NV12, w, h, src), cc))
    _, want = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
    assert np.array_equal(a, b) and np.array_equal(a, host_frame(want))


def test_chain_remap_sample(oracle):
    """samples/SampleRemap.py:74-101 — NV12 -> RGB -> Remap(RGB) -> download, BT.709 + JPEG"""
    w, h = 640, 360
    src = oracle.synth(oracle.NV12, w, h, 13)
    xm, ym = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    xm = (xm + 3.25 * np.sin(ym / 17)).astype(np.float32)
    ym = (ym + 2.5 * np.cos(xm / 23)).astype(np.float32)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.JPEG)
    rgb = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, GPU).Execute(upload(PF.NV12, w, h, src), cc)
    rm = nvc.PySurfaceRemaper(xm, ym, PF.RGB, GPU)
    out = rm.Execute(rgb)
    assert (out.Width(), out.Height(), out.Format()) == (w, h, PF.RGB)
    _, want_rgb = oracle.convert(oracle.NV12, oracle.RGB, 1, 1, w, h, src)
    _, want = oracle.remap(oracle.RGB, w, h, want_rgb, xm, ym)  # unmapped pixels stay 0 (surface starts black)
    assert np.array_equal(download(out), want[0].reshape(-1))
    assert rm.Execute(nvc.Surface.Make(PF.BGR, w, h, GPU)).Empty()  # format mismatch


def test_chain_multithread_sample(oracle):
    """samples/SampleDecodeMultiThread.py:50-152 — per-thread {stream, converter, resizer}: NV12 -> RGB -> Resize(W/2 x H/2)"""
    w, h = 1280, 720
    results, errors = {}, []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            ctx, s = nvc.GetContext(GPU), st.cuda_stream
            src = oracle.synth(oracle.NV12, w, h, 100 + i)
            up = nvc.PyFrameUploader(w, h, PF.NV12, ctx, s)
            conv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, ctx, s)
            rs = nvc.PySurfaceResizer(w // 2, h // 2, PF.RGB, ctx, s)
            dl = nvc.PySurfaceDownloader(w // 2, h // 2, PF.RGB, ctx, s)
            cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
            out = np.zeros(1, np.uint8)
            for _ in range(3):
                small = rs.Execute(conv.Execute(up.UploadSingleFrame(host_frame(src)), cc))
                assert dl.DownloadSingleSurface(small, out)
            results[i] = (src, out)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for i, (src, out) in results.items():
        _, rgb = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
        _, want = oracle.resize(oracle.RGB, oracle.LANCZOS3, w, h, rgb, w // 2, h // 2)
        assert np.array_equal(out, want[0].reshape(-1)), i


def test_encode_return_path(oracle):
    """samples/SamplePyTorch.py:150-158 return path — RGB_PLANAR -> RGB -> YUV420 -> NV12 (BT.601 MPEG)"""
    w, h = 320, 180
    pln = oracle.synth(oracle.RGB_PLANAR, w, h, 14)
    cc = nvc.ColorspaceConversionContext(CS.BT_601, CR.MPEG)
    s = upload(PF.RGB_PLANAR, w, h, pln)
    rgb = nvc.PySurfaceConverter(w, h, PF.RGB_PLANAR, PF.RGB, GPU).Execute(s, cc)
    yuv = nvc.PySurfaceConverter(w, h, PF.RGB, PF.YUV420, GPU).Execute(rgb, cc)
    nv12 = nvc.PySurfaceConverter(w, h, PF.YUV420, PF.NV12, GPU).Execute(yuv, cc)
    _, a = oracle.convert(oracle.RGB_PLANAR, oracle.RGB, 0, 0, w, h, pln)
    _, b = oracle.convert(oracle.RGB, oracle.YUV420, 0, 0, w, h, a)
    _, c = oracle.convert(oracle.YUV420, oracle.NV12, 0, 0, w, h, b)
    assert np.array_equal(download(nv12), host_frame(c))


def test_converter_output_aliases_internal_surface(oracle):
    """§3.1 consequence (a): Execute returns a non-owning alias of ONE internal surface, overwritten by the next
    Execute; Clone() is the deep copy samples use (samples/SamplePyTorch.py:83)"""
    w, h = 128, 64
    a, b = oracle.synth(oracle.NV12, w, h, 15), oracle.synth(oracle.NV12, w, h, 16)
    conv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, GPU)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
    out_a = conv.Execute(upload(PF.NV12, w, h, a), cc)
    keep = out_a.Clone(GPU)
    out_b = conv.Execute(upload(PF.NV12, w, h, b), cc)
    assert not out_a.OwnMemory() and keep.OwnMemory()
    assert out_a.PlanePtr().GpuMem() == out_b.PlanePtr().GpuMem() != keep.PlanePtr().GpuMem()
    _, wa = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, a)
    _, wb = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, b)
    assert np.array_equal(download(out_a), wb[0].reshape(-1))   # the alias now shows frame b
    assert np.array_equal(download(keep), wa[0].reshape(-1))    # the clone kept frame a


def test_surface_copy_clone_crop(oracle):
    w, h = 96, 48
    for name in ("NV12", "RGB", "YUV420", "RGB_PLANAR"):
        fmt, ofmt = getattr(PF, name), getattr(oracle, name)
        planes = oracle.synth(ofmt, w, h, 17)
        s = upload(fmt, w, h, planes)
        d = nvc.Surface.Make(fmt, w, h, GPU)
        s.CopyFrom(d, GPU)  # SELF -> OTHER: the reference binding's direction (PySurface.cpp:54-81,361), kept for drop-in behaviour
        assert np.array_equal(download(d), host_frame(planes))
        assert np.array_equal(download(s), host_frame(planes))   # the source is untouched
        d2 = nvc.Surface.Make(fmt, w, h, GPU)
        d2.UpdateFrom(s, GPU)  # additive, unambiguous: src -> self
        assert np.array_equal(download(d2), host_frame(planes))
        d3 = nvc.Surface.Make(fmt, w, h, GPU)
        s.CopyFrom(d3, nvc.GetContext(GPU), nvc.GetStream(GPU))
        assert np.array_equal(download(d3), host_frame(planes))
        st = torch.cuda.Stream()
        assert np.array_equal(download(s.Clone(nvc.GetContext(GPU), st.cuda_stream)), host_frame(planes))
        assert np.array_equal(download(s.Clone()), host_frame(planes))
        with pytest.raises(RuntimeError, match="different size"):
            nvc.Surface.Make(fmt, w // 2, h, GPU).CopyFrom(s, GPU)
        x, y, cw, ch = 16, 8, 32, 20
        c = s.Crop(x, y, cw, ch, GPU)
        assert (c.Width(), c.Height(), c.Format()) == (cw, ch, fmt)
        got = download(c)
        if name == "RGB":
            want = planes[0].reshape(h, w, 3)[y:y + ch, x:x + cw].reshape(-1)
        elif name == "NV12":
            want = np.concatenate([planes[0][y:y + ch, x:x + cw].reshape(-1), planes[1][y // 2:(y + ch) // 2, x:x + cw].reshape(-1)])
        elif name == "YUV420":
            want = np.concatenate([planes[0][y:y + ch, x:x + cw].reshape(-1)] +
                                  [planes[k][y // 2:(y + ch) // 2, x // 2:(x + cw) // 2].reshape(-1) for k in (1, 2)])
        else:
            want = np.concatenate([planes[k][y:y + ch, x:x + cw].reshape(-1) for k in range(3)])
        assert np.array_equal(got, want), name
    with pytest.raises(RuntimeError, match="different pixel formats"):
        nvc.Surface.Make(PF.RGB, w, h, GPU).CopyFrom(nvc.Surface.Make(PF.BGR, w, h, GPU), GPU)


def test_surface_plane_import_export():
    w, h = 200, 50
    s = nvc.Surface.Make(PF.Y, w, h, GPU)
    t = torch.randint(0, 256, (h, w), dtype=torch.uint8, device="cuda")
    back = torch.zeros_like(t)
    torch.cuda.synchronize()  # torch kernels (randint / zeros) run on torch's stream, Import/Export on ours
    p = s.PlanePtr()
    p.Import(t.data_ptr(), w, GPU)
    p.Export(back.data_ptr(), w, GPU)
    assert torch.equal(t, back)
    t2 = pnvc.DptrToTensor(p.GpuMem(), p.Width(), p.Height(), p.Pitch(), p.ElemSize())
    assert torch.equal(t, t2)
    pnvc.TensorToDptr(255 - t, p.GpuMem(), p.Width(), p.Height(), p.Pitch(), p.ElemSize())
    assert torch.equal(pnvc.view_plane(p.GpuMem(), w, h, p.Pitch()), 255 - t)
    with pytest.raises(RuntimeError, match="only torch.uint8"):
        pnvc.makefromDevicePtrUint8(p.GpuMem(), w, h, p.Pitch(), 4)


def test_cuda_buffer_roundtrip():
    n = 1000
    a = np.random.default_rng(4).integers(0, 256, n * 4, dtype=np.uint8)
    up = nvc.PyBufferUploader(4, n, GPU)
    buf = up.UploadSingleBuffer(a)
    assert (buf.GetElemSize(), buf.GetNumElems(), buf.GetRawMemSize()) == (4, n, 4 * n) and buf.GpuMem()
    out = np.zeros(1, np.uint8)
    assert nvc.PyCudaBufferDownloader(4, n, GPU).DownloadSingleCudaBuffer(buf, out) and np.array_equal(out, a)
    other = nvc.CudaBuffer.Make(4, n, GPU)
    other.CopyFrom(buf, GPU)
    assert nvc.PyCudaBufferDownloader(4, n, GPU).DownloadSingleCudaBuffer(other.Clone(), out) and np.array_equal(out, a)
    with pytest.raises(RuntimeError, match="different size"):
        nvc.CudaBuffer.Make(4, n + 1, GPU).CopyFrom(buf, GPU)


def test_execute_batch(oracle):
    w, h, n = 640, 360, 40
    conv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, GPU)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
    srcs = [oracle.synth(oracle.NV12, w, h, 300 + i) for i in range(n)]
    ins = [upload(PF.NV12, w, h, s) for s in srcs]
    outs = [nvc.Surface.Make(PF.RGB, w, h, GPU) for _ in range(n)]
    assert conv.ExecuteBatch(ins, outs, cc)
    torch.cuda.synchronize()
    for i in (0, 17, 31, 32, 39):
        _, want = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, srcs[i])
        assert np.array_equal(download(outs[i]), want[0].reshape(-1)), i
    assert not conv.ExecuteBatch(ins, outs[:-1], cc)
    assert not conv.ExecuteBatch(ins, [nvc.Surface.Make(PF.BGR, w, h, GPU) for _ in range(n)], cc)


def test_num_gpus():
    assert nvc.GetNumGpus() == torch.cuda.device_count() >= 1


def test_convert_straight_into_a_torch_tensor(oracle):
    """BASELINE.json configs[4]: NV12 -> RGB_PLANAR written DIRECTLY into a torch tensor (true zero copy: the tensor is
    wrapped as a non-owning Surface and handed to the converter as its output), then a remap warp of the packed picture."""
    w, h = 1280, 720
    src = oracle.synth(oracle.NV12, w, h, 21, "B")
    nv12 = upload(PF.NV12, w, h, src)
    t = torch.zeros((3, h, w), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # torch's fill kernel is on torch's stream, the converter on its own: order them
    out = pnvc.surface_from_tensor(t)
    assert out.Format() == PF.RGB_PLANAR and not out.OwnMemory() and out.PlanePtr().GpuMem() == t.data_ptr()
    conv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB_PLANAR, GPU)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
    assert conv.ExecuteBatch([nv12], [out], cc)
    torch.cuda.synchronize()
    _, want = oracle.convert(oracle.NV12, oracle.RGB_PLANAR, 1, 0, w, h, src)
    assert np.array_equal(t.cpu().numpy(), np.stack(want))                 # bit-exact (FP32 oracle)
    _, ex = oracle.convert(oracle.NV12, oracle.RGB_PLANAR, 1, 0, w, h, src, oracle.EXACT)
    assert np.abs(t.cpu().numpy().astype(int) - np.stack(ex).astype(int)).max() <= 1   # +-1 LSB vs the specification level
    # packed tensor as converter output + remap of it (samples/SampleRemap.py)
    tp = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, GPU).ExecuteBatch([nv12], [pnvc.surface_from_tensor(tp)], cc)
    xm, ym = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    xm = xm * 0.97 + 11.5
    warped = nvc.PySurfaceRemaper(xm, ym, PF.RGB, GPU).Execute(pnvc.surface_from_tensor(tp))
    _, rgb = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
    _, wantw = oracle.remap(oracle.RGB, w, h, rgb, xm, ym)
    assert np.array_equal(download(warped), wantw[0].reshape(-1))
    with pytest.raises(ValueError):
        nvc.Surface.Wrap(PF.YUV420, w, h, w, t.data_ptr())               # three allocations cannot wrap one pointer


def test_download_into_pinned_and_pageable_arrays(oracle):
    """PySurfaceDownloader: a page-locked destination (AllocPinned) receives the DMA directly, a pageable one goes through
    the staging buffer; wrong-size pageable arrays are resized like the reference does (PySurfaceDownloader.cpp:119-189);
    threads download concurrently with the GIL released"""
    w, h = 640, 360
    src = oracle.synth(oracle.NV12, w, h, 21)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
    rgb = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, GPU).Execute(upload(PF.NV12, w, h, src), cc)
    _, want = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
    dl = nvc.PySurfaceDownloader(w, h, PF.RGB, GPU)
    pinned = nvc.AllocPinned(w * h * 3)
    pinned[:] = 0
    assert dl.DownloadSingleSurface(rgb, pinned) and np.array_equal(pinned, want[0].reshape(-1))
    small = np.zeros(7, np.uint8)
    assert dl.DownloadSingleSurface(rgb, small) and small.size == w * h * 3 and np.array_equal(small, want[0].reshape(-1))
    assert not dl.DownloadSingleSurface(nvc.Surface.Make(PF.RGB, 2 * w, 2 * h, GPU), np.zeros(1, np.uint8))  # larger than built for
    # NV12 surface: planes concatenated at tight width
    nv = upload(PF.NV12, w, h, src)
    out = nvc.AllocPinned(w * h * 3 // 2)
    assert nvc.PySurfaceDownloader(w, h, PF.NV12, GPU).DownloadSingleSurface(nv, out) and np.array_equal(out, host_frame(src))
    res = [None] * 4

    def work(i):
        d = nvc.PySurfaceDownloader(w, h, PF.RGB, nvc.GetContext(GPU), torch.cuda.Stream().cuda_stream)
        buf = nvc.AllocPinned(w * h * 3) if i % 2 else np.empty(w * h * 3, np.uint8)
        ok = all(d.DownloadSingleSurface(rgb, buf) for _ in range(20))
        res[i] = ok and np.array_equal(buf, want[0].reshape(-1))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(res)


@pytest.mark.parametrize("fmt,ofmt,w,h", [(PF.RGB, "RGB", 3840, 2160), (PF.NV12, "NV12", 3840, 2160), (PF.YUV420, "YUV420", 3840, 2160),
                                          (PF.RGB, "RGB", 2502, 1407), (PF.NV12, "NV12", 4098, 1026), (PF.RGB, "RGB", 1400, 1000)])  # ragged last pieces; just above 4 MB
def test_download_of_a_large_frame_into_a_pageable_array_goes_in_pieces(oracle, fmt, ofmt, w, h):
    """Frames of 4 MB and more reach a pageable array piece by piece (DMA of piece k+1 under the host copy of piece k,
    Tasks.cpp DownloadInto); the bytes are those of the direct DMA into AllocPinned memory and of the host frame, repeatedly"""
    src = oracle.synth(getattr(oracle, ofmt), w, h, 77)
    surf = upload(fmt, w, h, src)
    want = host_frame(src)
    dl = nvc.PySurfaceDownloader(w, h, fmt, GPU)
    pinned = nvc.AllocPinned(want.size)
    assert dl.DownloadSingleSurface(surf, pinned) and np.array_equal(pinned, want)
    for i in range(4):
        out = np.full(want.size if i % 2 else 5, 0xA5, np.uint8)
        assert dl.DownloadSingleSurface(surf, out) and out.size == want.size and np.array_equal(out, want)


def test_upload_from_pinned_memory():
    """AllocPinned: numpy array over page-locked memory; the uploader DMAs from it directly and the result is identical"""
    w, h = 640, 360
    n = w * h * 3 // 2
    a = np.random.default_rng(9).integers(0, 256, n, dtype=np.uint8)
    p = nvc.AllocPinned(n)
    assert p.dtype == np.uint8 and p.shape == (n,)
    p[:] = a
    up = nvc.PyFrameUploader(w, h, PF.NV12, GPU)
    for _ in range(3):
        assert np.array_equal(download(up.UploadSingleFrame(p)), a)
        assert np.array_equal(download(up.UploadSingleFrame(a)), a)


def test_uploaded_surface_is_complete_for_a_consumer_on_another_stream():
    """the reference's upload blocks until the copy is done (src/TC/src/Tasks.cpp:617-618), so user code reads the returned surface from any
    stream.  Default here: the same — a downloader on a DIFFERENT stream, which nothing orders behind the uploader's private copy stream,
    sees the whole new frame every time (4K frames: the copy takes a few hundred microseconds, long enough to lose the race if the call
    returned early).  SetAsync(True) is the opt-in to stream-ordered completion; it still yields the right frame to a consumer on the
    uploader's own stream."""
    w, h = 3840, 2160
    n = w * h * 3 // 2
    ctx = nvc.GetContext(GPU)
    s_up, s_dl = torch.cuda.Stream(), torch.cuda.Stream()
    up = nvc.PyFrameUploader(w, h, PF.NV12, ctx, s_up.cuda_stream)
    assert up.GetAsync() is False
    dl_other = nvc.PySurfaceDownloader(w, h, PF.NV12, ctx, s_dl.cuda_stream)
    dl_same = nvc.PySurfaceDownloader(w, h, PF.NV12, ctx, s_up.cuda_stream)
    rng = np.random.default_rng(77)
    out = np.zeros(n, np.uint8)
    pinned = nvc.AllocPinned(n)
    for i in range(6):
        frame = rng.integers(0, 256, n, dtype=np.uint8)
        if i & 1:  # page-locked source: DMA'd in place
            pinned[:] = frame
            surf = up.UploadSingleFrame(pinned)
        else:      # pageable source: staged
            surf = up.UploadSingleFrame(frame)
        assert dl_other.DownloadSingleSurface(surf, out)
        assert np.array_equal(out, frame), f"frame {i}: a consumer on another stream read an incomplete upload"
    up.SetAsync(True)
    assert up.GetAsync() is True
    for i in range(3):
        frame = rng.integers(0, 256, n, dtype=np.uint8)
        surf = up.UploadSingleFrame(frame)
        assert dl_same.DownloadSingleSurface(surf, out)
        assert np.array_equal(out, frame)


def test_concurrent_threads_mixed_operations(oracle):
    """8 threads, each with its own stream and task objects, hammer converters / fused resize / resizer / remaper /
    uploader / downloader concurrently (GIL released inside the calls): every result must stay bit-exact.  Guards the
    process-wide pieces (device guard, pair table, HIP error clearing, pinned staging) against races."""
    w, h, tw, th = 640, 360, 320, 120
    src = oracle.synth(oracle.NV12, w, h, 31)
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
    _, rgb = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
    _, pln = oracle.convert(oracle.NV12, oracle.RGB_PLANAR, 1, 0, w, h, src)
    _, yuv = oracle.convert(oracle.NV12, oracle.YUV420, 1, 0, w, h, src)
    _, small = oracle.resize(oracle.RGB, oracle.LANCZOS3, w, h, rgb, tw, th)
    _, fused = oracle.convert_resize(oracle.NV12, oracle.RGB_PLANAR, 1, 0, w, h, src, tw, th)
    xm, ym = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    xm, ym = (xm + 2.5 * np.sin(ym / 11)).astype(np.float32), (ym + 1.5 * np.cos(xm / 13)).astype(np.float32)
    _, warped = oracle.remap(oracle.RGB, w, h, rgb, xm, ym, dst=oracle.alloc(oracle.RGB, w, h))
    frame = host_frame(src)
    errors = []

    def work(tid):
        try:
            stream = torch.cuda.Stream()
            ctx, st = nvc.GetContext(GPU), stream.cuda_stream
            up = nvc.PyFrameUploader(w, h, PF.NV12, ctx, st)
            to_rgb = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, ctx, st)
            to_pln = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB_PLANAR, ctx, st)
            to_yuv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.YUV420, ctx, st)
            rs = nvc.PySurfaceResizer(tw, th, PF.RGB, ctx, st)
            fz = nvc.PySurfaceConvertResizer(w, h, PF.NV12, tw, th, PF.RGB_PLANAR, ctx, st)
            rm = nvc.PySurfaceRemaper(xm, ym, PF.RGB, ctx, st)
            dls = {f: nvc.PySurfaceDownloader(w, h, f, ctx, st) for f in (PF.RGB, PF.RGB_PLANAR, PF.YUV420)}
            dl_small = nvc.PySurfaceDownloader(tw, th, PF.RGB, ctx, st)
            dl_small_pln = nvc.PySurfaceDownloader(tw, th, PF.RGB_PLANAR, ctx, st)
            out = np.empty(1, np.uint8)
            for it in range(25):
                nv12 = up.UploadSingleFrame(frame)
                op = (it + tid) % 6
                if op == 0:
                    got, want, d = to_rgb.Execute(nv12, cc), host_frame(rgb), dls[PF.RGB]
                elif op == 1:
                    got, want, d = to_pln.Execute(nv12, cc), host_frame(pln), dls[PF.RGB_PLANAR]
                elif op == 2:
                    got, want, d = to_yuv.Execute(nv12, cc), host_frame(yuv), dls[PF.YUV420]
                elif op == 3:
                    got, want, d = rs.Execute(to_rgb.Execute(nv12, cc)), host_frame(small), dl_small
                elif op == 4:
                    got, want, d = fz.Execute(nv12, cc), host_frame(fused), dl_small_pln
                else:
                    got, want, d = rm.Execute(to_rgb.Execute(nv12, cc)), host_frame(warped), dls[PF.RGB]
                assert not got.Empty(), (tid, it, op)
                assert d.DownloadSingleSurface(got, out) and np.array_equal(out, want), (tid, it, op)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:3]


def test_tracing_hooks_kernel_selection_log_and_roctx_ranges():
    """the reference's NvtxMark (src/TC/inc/Tasks.hpp:27-52) has a counterpart — VPF_HIP_LOG=2 names the kernel every
    launch selected, VPF_HIP_ROCTX=1 wraps ABI entries / Task::Run in roctx ranges (the library is dlopen()ed; no link dependency)"""
    import subprocess
    import sys

    code = f"""
import sys
sys.path.insert(0, {os.path.join(ROOT, 'videoprocessingframework_amd')!r})
import numpy as np
import PyNvCodec as nvc
w, h = 256, 64
up = nvc.PyFrameUploader(w, h, nvc.PixelFormat.NV12, 0)
conv = nvc.PySurfaceConverter(w, h, nvc.PixelFormat.NV12, nvc.PixelFormat.RGB, 0)
rs = nvc.PySurfaceResizer(96, 32, nvc.PixelFormat.RGB, 0)
s = conv.Execute(up.UploadSingleFrame(np.arange(w * h * 3 // 2, dtype=np.uint8)), nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_709, nvc.ColorRange.MPEG))
assert not s.Empty() and not rs.Execute(s).Empty()
print("done")
"""
    env = dict(os.environ, VPF_HIP_LOG="2", VPF_HIP_ROCTX="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "done" in r.stdout, r.stdout + r.stderr
    assert "libvpfhip: launch (k_nv12_rgb_p16_one<" in r.stderr and "libvpfhip: launch (k_resize" in r.stderr, r.stderr
    assert "no roctx library" not in r.stderr
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, VPF_HIP_LOG="0"), timeout=300)
    assert r.returncode == 0 and "libvpfhip" not in r.stderr


def test_resizer_and_remaper_execute_batch(oracle):
    """additive PySurfaceResizer.ExecuteBatch / PySurfaceRemaper.ExecuteBatch: n surfaces -> n caller-owned surfaces, same pixels as n
    Execute() calls (and as the oracle)"""
    w, h, dw, dh, n = 320, 180, 200, 120, 5
    for name in ("NV12", "RGB", "YUV420"):
        fmt, ofmt = getattr(PF, name), getattr(oracle, name)
        planes = [oracle.synth(ofmt, w, h, 50 + i) for i in range(n)]
        srcs = [upload(fmt, w, h, p) for p in planes]
        for interp in (1, 2):
            rs = nvc.PySurfaceResizer(dw, dh, fmt, GPU)
            rs.SetInterpolation(interp)
            outs = [nvc.Surface.Make(fmt, dw, dh, GPU) for _ in range(n)]
            assert rs.ExecuteBatch(srcs, outs)
            torch.cuda.synchronize()
            for i in range(n):
                want = host_frame(oracle.resize(ofmt, interp, w, h, planes[i], dw, dh, oracle.FP32)[1])
                assert np.array_equal(download(outs[i]), want), (name, interp, i)
                assert np.array_equal(download(rs.Execute(srcs[i])), want)
        assert not nvc.PySurfaceResizer(dw, dh, fmt, GPU).ExecuteBatch(srcs, outs[:2])                           # length mismatch
        assert not nvc.PySurfaceResizer(dw, dh, fmt, GPU).ExecuteBatch(srcs, [nvc.Surface.Make(fmt, dw + 2, dh, GPU) for _ in range(n)])
    yy, xx = np.meshgrid(np.arange(dh, dtype=np.float32), np.arange(dw, dtype=np.float32), indexing="ij")
    xm, ym = (xx * (w / dw) + 0.3).astype(np.float32), (yy * (h / dh) + 0.6).astype(np.float32)
    planes = [oracle.synth(oracle.RGB, w, h, 80 + i) for i in range(n)]
    srcs = [upload(PF.RGB, w, h, p) for p in planes]
    rm = nvc.PySurfaceRemaper(xm, ym, PF.RGB, GPU)
    outs = [nvc.Surface.Make(PF.RGB, dw, dh, GPU) for _ in range(n)]
    assert rm.ExecuteBatch(srcs, outs)
    torch.cuda.synchronize()
    inside = ((xm <= w - 1) & (ym <= h - 1))[:, :, None].repeat(3, 2).reshape(-1)
    for i in range(n):
        want = oracle.remap(oracle.RGB, w, h, planes[i], xm, ym)[1][0].reshape(-1)
        assert np.array_equal(download(outs[i])[inside], want[inside]), i
        assert np.array_equal(download(rm.Execute(srcs[i]))[inside], want[inside])


def test_one_resizer_fed_many_source_shapes_with_a_shared_width(oracle):
    """ResizeSurface keeps ONE table workspace for its destination size, whatever the sources.  Sources that share a width
    share their column tables (a HIT in the workspace record) while every new height adds row tables — until the eight-entry record is
    full and the launch that has just hit the column tables misses on the row tables: round 4 started the record over there and handed the
    hit tables' bytes to the new build (one corrupted frame).  Now that table goes to the arena.  Every frame equals the oracle."""
    dw, dh = 640, 360
    for name in ("NV12", "YUV420", "RGB"):
        fmt, ofmt = getattr(PF, name), getattr(oracle, name)
        rs = nvc.PySurfaceResizer(dw, dh, fmt, GPU)   # default filter: Lanczos-3, the one with tables
        for rnd, (w, h) in enumerate([(960, 540), (960, 544), (960, 536), (960, 528), (960, 520), (960, 512), (960, 540), (968, 540), (960, 544), (960, 504)]):
            planes = oracle.synth(ofmt, w, h, 700 + rnd)
            got = download(rs.Execute(upload(fmt, w, h, planes)))
            want = host_frame(oracle.resize(ofmt, 2, w, h, planes, dw, dh, oracle.FP32)[1])
            assert np.array_equal(got, want), (name, rnd, w, h)


def test_resizer_and_remaper_async_opt_out(oracle):
    """additive SetAsync(True): Execute() stops waiting for the stream (default: blocking, like the reference's cuda_stream_sync callback);
    same pixels once the stream is synchronised"""
    w, h, dw, dh = 640, 360, 300, 200
    planes = oracle.synth(oracle.RGB, w, h, 91)
    src = upload(PF.RGB, w, h, planes)
    rs = nvc.PySurfaceResizer(dw, dh, PF.RGB, GPU)
    assert rs.GetAsync() is False
    rs.SetAsync(True)
    assert rs.GetAsync() is True
    out = rs.Execute(src)
    torch.cuda.synchronize()
    assert np.array_equal(download(out), host_frame(oracle.resize(oracle.RGB, 2, w, h, planes, dw, dh, oracle.FP32)[1]))
    yy, xx = np.meshgrid(np.arange(dh, dtype=np.float32), np.arange(dw, dtype=np.float32), indexing="ij")
    rm = nvc.PySurfaceRemaper((xx * 2).astype(np.float32), (yy * 1.5).astype(np.float32), PF.RGB, GPU)
    rm.SetAsync(True)
    out = rm.Execute(src)
    torch.cuda.synchronize()
    assert np.array_equal(download(out), oracle.remap(oracle.RGB, w, h, planes, (xx * 2).astype(np.float32), (yy * 1.5).astype(np.float32))[1][0].reshape(-1))
