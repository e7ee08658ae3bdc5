"""End-to-end feed path: host NV12 frames (what a software decoder hands over) -> PyFrameUploader
(pinned staging + side copy stream) -> PySurfaceConverter NV12->RGB, through the drop-in Python API.  Reports frames/s for
upload only, convert only and the pipeline, single thread and N threads (one stream + task chain per thread, the pattern of
samples/SampleDecodeMultiThread.py).  PCIe-inclusive numbers: never the headline `value`."""
import os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
import PyNvCodec as nvc

W, H, N = 3840, 2160, 200
ASYNC = "--async" in sys.argv  # uploaders return once their copy is queued (SetAsync(True)); default: the reference's blocking upload
PF = nvc.PixelFormat
frames = [np.random.default_rng(i).integers(0, 256, W * H * 3 // 2, dtype=np.uint8) for i in range(8)]
if "--pinned" in sys.argv:  # frames decoded straight into page-locked memory: no staging memcpy
    pinned = [nvc.AllocPinned(f.size) for f in frames]
    for p_, f in zip(pinned, frames):
        p_[:] = f
    frames = pinned
cc = nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_709, nvc.ColorRange.MPEG)


def run(kind, n=N, ctx=None, stream=None):
    ctx = nvc.GetContext(0) if ctx is None else ctx
    stream = nvc.GetStream(0) if stream is None else stream
    up = nvc.PyFrameUploader(W, H, PF.NV12, ctx, stream)
    up.SetAsync(ASYNC)  # default: wait for every copy like the reference; --async: return once the copy is queued
    conv = nvc.PySurfaceConverter(W, H, PF.NV12, PF.RGB, ctx, stream)
    s = up.UploadSingleFrame(frames[0])
    conv.Execute(s, cc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if kind != "convert":
            s = up.UploadSingleFrame(frames[i % 8])
        if kind != "upload":
            conv.Execute(s, cc)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def run_download(pinned_dst, n=N, multi=1):
    """convert -> PySurfaceDownloader into a numpy array (pageable) or an AllocPinned() array (direct DMA)"""
    ctx, stream = nvc.GetContext(0), nvc.GetStream(0)
    up = nvc.PyFrameUploader(W, H, PF.NV12, ctx, stream)
    conv = nvc.PySurfaceConverter(W, H, PF.NV12, PF.RGB, ctx, stream)
    dl = nvc.PySurfaceDownloader(W, H, PF.RGB, ctx, stream)
    rgb = conv.Execute(up.UploadSingleFrame(frames[0]), cc)
    out = nvc.AllocPinned(W * H * 3) if pinned_dst else np.empty(W * H * 3, np.uint8)
    assert dl.DownloadSingleSurface(rgb, out)
    t0 = time.perf_counter()
    for _ in range(n):
        dl.DownloadSingleSurface(rgb, out)
    return n / (time.perf_counter() - t0)


for kind in ("upload", "convert", "pipeline"):
    print(f"[pipeline] 1 thread  {kind:8s}: {run(kind):9.1f} frames/s", flush=True)
for nt in (2, 4, 8, 16):
    res = [0.0] * nt
    streams = [torch.cuda.Stream() for _ in range(nt)]

    def work(i):
        res[i] = run("pipeline", N, nvc.GetContext(0), streams[i].cuda_stream)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.perf_counter() - t0
    print(f"[pipeline] {nt:2d} threads pipeline: {nt * N / dt:9.1f} frames/s aggregate = {nt * N / dt * W * H / 1e9:6.2f} Gpix/s, "
          f"{nt * N / dt * W * H * 1.5 / 1e9:5.1f} GB/s over PCIe", flush=True)

for pinned_dst in (False, True):
    fps = run_download(pinned_dst)
    print(f"[pipeline] download 4K RGB into {'AllocPinned() array (direct DMA)' if pinned_dst else 'pageable numpy array (staging + 1 copy)'}: "
          f"{fps:8.1f} frames/s = {fps * W * H * 3 / 1e9:5.1f} GB/s", flush=True)
