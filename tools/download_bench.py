"""PySurfaceDownloader rate for one frame size: into a pageable numpy array (pinned staging + host copy, in pieces for frames >= 4 MB)
and into an AllocPinned() array (direct DMA).  VPF_HIP_SYNC_SPIN_US=0 gives the two-step form of the pageable case for comparison.
python tools/download_bench.py [W H [frames]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "videoprocessingframework_amd"))
import PyNvCodec as nvc  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 400
PF = nvc.PixelFormat
for fmt, name, bpp in ((PF.RGB, "RGB", 3.0), (PF.NV12, "NV12", 1.5)):
    size = int(W * H * bpp)
    surf = nvc.PyFrameUploader(W, H, fmt, 0).UploadSingleFrame(np.random.default_rng(1).integers(0, 256, size, dtype=np.uint8)).Clone(0)
    dl = nvc.PySurfaceDownloader(W, H, fmt, 0)
    for pinned in (False, True):
        out = nvc.AllocPinned(size) if pinned else np.empty(size, np.uint8)
        for _ in range(10):
            assert dl.DownloadSingleSurface(surf, out)
        t0 = time.perf_counter()
        for _ in range(N):
            dl.DownloadSingleSurface(surf, out)
        fps = N / (time.perf_counter() - t0)
        print(f"[download] {name:5s} {W}x{H} into {'AllocPinned()' if pinned else 'pageable    '}: {fps:8.1f} frames/s = {fps * size / 1e9:5.1f} GB/s "
              f"(VPF_HIP_SYNC_SPIN_US={os.environ.get('VPF_HIP_SYNC_SPIN_US', 'default')})", flush=True)
