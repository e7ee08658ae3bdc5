"""Stress: caller buffers that SHARE PAGES (small numpy arrays cut from the heap one after the other) registered one by one by HostPinCache, released,
then the same memory used by torch's pageable copies.  python tools/lab/probes/pin_cache_adjacent_buffers.py [iterations]"""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
import PyNvCodec as nvc
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w, h = 640, 360
n = w * h * 3 // 2
up = nvc.PyFrameUploader(w, h, nvc.PixelFormat.NV12, 0)
dl = nvc.PySurfaceDownloader(w, h, nvc.PixelFormat.NV12, 0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
big = [np.empty(40 << 20, np.uint8) for _ in range(2)]  # raise glibc's mmap threshold: what follows comes from the heap, page-sharing neighbours
del big
for it in range(iters):
    pool = [rng.integers(0, 256, n, dtype=np.uint8) for _ in range(40)]
    shared = sum(1 for a, b in zip(pool, pool[1:]) if (a.ctypes.data + n - 1) // 4096 == b.ctypes.data // 4096)
    for rep in range(2):
        for f in pool:
            s = up.UploadSingleFrame(f)
    out = np.zeros(1, np.uint8)
    assert dl.DownloadSingleSurface(s, out) and np.array_equal(out, pool[-1])
    st = dict(nvc.PinCacheStats())
    del pool[::2]                     # every other buffer dies (unregistered) while its page-sharing neighbours stay registered
    gc.collect()
    for f in pool:
        s = up.UploadSingleFrame(f)   # the survivors are read in place again
    assert dl.DownloadSingleSurface(s, out) and np.array_equal(out, pool[-1])
    del pool, f, s
    gc.collect()
    for k in range(20):               # the same heap memory through torch's pageable copies
        t = torch.randint(0, 255, (h * 3 // 2, w), dtype=torch.uint8, device=dev)
        c = t.cpu()
        assert torch.equal(c.to(dev), t)
    if it % 10 == 0:
        print(f"[adjacent] iteration {it}: {shared} of 39 neighbours share a page; cache {st}", flush=True)
print("[adjacent] ok", dict(nvc.PinCacheStats()))
