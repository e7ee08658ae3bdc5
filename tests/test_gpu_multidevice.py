"""-m gpu tests parametrised over EVERY device the box has: the reference's multi-GPU model is "pass a different gpu_id"
(/root/reference/src/PyNvCodec/src/PyNvCodec.cpp:57-111, samples/SampleDecodeMultiThread.py:50-115), so everything device-scoped in this
library — the per-device Lanczos weight tables, the per-device big-LDS attribute bit, DeviceGuard, the uploader's private copy stream and
events — has to work for gpu_id = d while some OTHER device is current.  On a 1-GPU lease the parametrisation collapses to device 0 (and
"another device is current" to the same one); on the driver's 8-GPU node the same tests touch devices 0..7 without any edit."""
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

PF, CS, CR = nvc.PixelFormat, nvc.ColorSpace, nvc.ColorRange
NDEV = torch.cuda.device_count()
DEVICES = list(range(NDEV))


def _other(d):
    """a device that is NOT d when the box has one (the point of these tests), else d"""
    return (d + 1) % NDEV


def _maps(w, h):
    xm, ym = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    return (xm + 2.25 * np.sin(ym / 13)).astype(np.float32), (ym + 1.5 * np.cos(xm / 19)).astype(np.float32)


def _chain_on(d, oracle, seed, w=640, h=360, dw=426, dh=240):
    """uploader -> NV12->RGB converter -> Lanczos resizer (the default filter: per-device weight tables) -> remaper -> downloader, every task
    built with gpu_id = d; returns (got bytes, wanted bytes)"""
    src = oracle.synth(oracle.NV12, w, h, seed)
    frame = np.concatenate([p.reshape(-1) for p in src])
    cc = nvc.ColorspaceConversionContext(CS.BT_709, CR.MPEG)
    up = nvc.PyFrameUploader(w, h, PF.NV12, d)
    conv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, d)
    rs = nvc.PySurfaceResizer(dw, dh, PF.RGB, d)
    xm, ym = _maps(dw, dh)
    rm = nvc.PySurfaceRemaper(xm, ym, PF.RGB, d)
    dl = nvc.PySurfaceDownloader(dw, dh, PF.RGB, d)
    out = np.zeros(1, np.uint8)
    for _ in range(2):  # twice: the second pass finds the shape's weight tables already built on this device
        surf = rm.Execute(rs.Execute(conv.Execute(up.UploadSingleFrame(frame), cc)))
        assert not surf.Empty()
        assert dl.DownloadSingleSurface(surf, out)
    _, rgb = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
    _, small = oracle.resize(oracle.RGB, oracle.LANCZOS3, w, h, rgb, dw, dh)
    _, want = oracle.remap(oracle.RGB, dw, dh, small, xm, ym)
    return out.copy(), want[0].reshape(-1)


@pytest.mark.parametrize("d", DEVICES)
def test_python_chain_with_gpu_id(oracle, d):
    """the drop-in classes with gpu_id = d while ANOTHER device is the current one (PyNvCodec.cpp:57-111 keeps one context + stream per GPU)"""
    torch.cuda.set_device(_other(d))
    try:
        got, want = _chain_on(d, oracle, 500 + d)
        assert np.array_equal(got, want), f"gpu_id {d}"
        assert torch.cuda.current_device() == _other(d)  # the library restored the caller's device
    finally:
        torch.cuda.set_device(0)


@pytest.mark.parametrize("d", DEVICES)
def test_c_abi_with_exec_device(capi, oracle, d):
    """the C ABI with vpf_exec.device = d and a stream of device d, while a different device is current: convert, Lanczos resize (weight tables
    and the > 64 KB LDS attribute are per device), bilinear resize, remap — each against the oracle"""
    w, h, dw, dh = 1920, 128, 1280, 86
    src = oracle.synth(oracle.NV12, w, h, 900 + d)
    dev = torch.device("cuda", d)
    with torch.cuda.device(d):
        stream = torch.cuda.Stream(device=dev)
        y, uv = torch.from_numpy(src[0]).to(dev), torch.from_numpy(src[1]).to(dev)
        rgb = torch.zeros((h, 3 * w), dtype=torch.uint8, device=dev)
        lz = torch.zeros((dh, 3 * dw), dtype=torch.uint8, device=dev)
        bl = torch.zeros((dh, 3 * dw), dtype=torch.uint8, device=dev)
        xm, ym = _maps(dw, dh)
        dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
        warped = torch.zeros((dh, 3 * dw), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
    torch.cuda.set_device(_other(d))
    try:
        ex = capi.make_exec(stream.cuda_stream, device=d)
        capi.convert(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, w, h, [(y.data_ptr(), w), (uv.data_ptr(), w)], [(rgb.data_ptr(), 3 * w)])
        capi.resize(ex, capi.RGB, capi.INTERP_LANCZOS3, w, h, [(rgb.data_ptr(), 3 * w)], dw, dh, [(lz.data_ptr(), 3 * dw)])
        capi.resize(ex, capi.RGB, capi.INTERP_LINEAR, w, h, [(rgb.data_ptr(), 3 * w)], dw, dh, [(bl.data_ptr(), 3 * dw)])
        capi.remap(ex, capi.RGB, dw, dh, (lz.data_ptr(), 3 * dw), dxm.data_ptr(), 4 * dw, dym.data_ptr(), 4 * dw, dw, dh, (warped.data_ptr(), 3 * dw))
        assert torch.cuda.current_device() == _other(d)
        stream.synchronize()
    finally:
        torch.cuda.set_device(0)
    _, want_rgb = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
    _, want_lz = oracle.resize(oracle.RGB, oracle.LANCZOS3, w, h, want_rgb, dw, dh)
    _, want_bl = oracle.resize(oracle.RGB, 1, w, h, want_rgb, dw, dh)
    _, want_warp = oracle.remap(oracle.RGB, dw, dh, want_lz, xm, ym)
    assert np.array_equal(rgb.cpu().numpy(), want_rgb[0])
    assert np.array_equal(lz.cpu().numpy(), want_lz[0]), f"Lanczos on device {d}"
    assert np.array_equal(bl.cpu().numpy(), want_bl[0])
    assert np.array_equal(warped.cpu().numpy(), want_warp[0])


def test_one_thread_per_device(oracle):
    """samples/SampleDecodeMultiThread.py:50-115: one worker thread per GPU, each building its own task objects with its gpu_id, all running
    at once (on a 1-GPU box: two threads on device 0)"""
    results, errors = {}, []

    def worker(i, d):
        try:
            results[i] = _chain_on(d, oracle, 700 + i, w=1280, h=96, dw=854, dh=64)
        except Exception as e:  # noqa: BLE001
            errors.append((i, d, repr(e)))

    jobs = [(i, DEVICES[i % NDEV]) for i in range(max(2, NDEV))]
    ts = [threading.Thread(target=worker, args=j) for j in jobs]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for i, (got, want) in results.items():
        assert np.array_equal(got, want), f"thread {i}"
