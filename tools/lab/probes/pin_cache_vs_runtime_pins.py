"""Stress: does HostPinCache's hipHostRegister / hipHostUnregister of numpy memory collide with the HIP runtime's own on-the-fly pinning of pageable
host memory (what hipMemcpy from / to numpy and torch CPU tensors does)?  Addresses are reused on purpose: a frame-sized array is copied by the
runtime (pageable path), freed, reallocated (usually at the same address), registered by two blocking uploads, read in place, freed (unregistered),
reallocated and copied by the runtime again.  Several worker processes share the GPU like the test suite's xdist workers.
python tools/lab/probes/pin_cache_vs_runtime_pins.py [iterations [processes]]"""
import gc, os, sys, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def work(rank, iters):
    import numpy as np, torch
    sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
    import PyNvCodec as nvc
    w, h = 848, 464
    n = w * h * 3 // 2
    up = nvc.PyFrameUploader(w, h, nvc.PixelFormat.NV12, 0)
    dl = nvc.PySurfaceDownloader(w, h, nvc.PixelFormat.NV12, 0)
    rng = np.random.default_rng(rank)
    dev = torch.device("cuda", 0)
    for i in range(iters):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        t = torch.from_numpy(a).to(dev)                       # the runtime's pageable H2D path on a's pages
        back = t.cpu().numpy()                                # ... and its pageable D2H path
        assert np.array_equal(back, a)
        del a, back
        b = np.empty(n, np.uint8)                             # usually a's address again
        b[:] = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(3):                                    # registered on second sight, read in place on the third
            s = up.UploadSingleFrame(b)
        out = np.zeros(1, np.uint8)
        assert dl.DownloadSingleSurface(s, out) and np.array_equal(out, b)
        del b, s
        gc.collect()                                          # owner gone: unregistered
        c = np.empty(n, np.uint8)                             # the same pages once more, through the runtime's path
        c[:] = 7
        assert int(torch.from_numpy(c).to(dev).sum().item()) == 7 * n
        v = torch.randint(0, 255, (3 * h, w), dtype=torch.uint8, device=dev)
        assert v.cpu().numpy().shape == (3 * h, w)
        del c, v
    print(f"[pin-vs-runtime] worker {rank}: {iters} iterations ok, cache {dict(nvc.PinCacheStats())}", flush=True)


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    mp.set_start_method("spawn")
    ps = [mp.Process(target=work, args=(r, iters)) for r in range(procs)]
    [p.start() for p in ps]
    [p.join() for p in ps]
    codes = [p.exitcode for p in ps]
    print(f"[pin-vs-runtime] exit codes {codes}")
    sys.exit(0 if all(c == 0 for c in codes) else 1)
