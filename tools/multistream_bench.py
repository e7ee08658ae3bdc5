"""One launch per frame (the unmodified per-Execute API), frames dealt round-robin to K HIP streams from one host thread:
do kernels of different streams overlap enough to hide a single frame's load / compute / store bursts?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
for (w, h, N) in ((3840, 2160, 32), (1920, 1080, 64)):
    p1, p3 = (w + 255) // 256 * 256, (3 * w + 255) // 256 * 256
    src = [torch.randint(0, 256, (h * 3 // 2, p1), dtype=torch.uint8, device=dev) for _ in range(N)]
    dst = [torch.zeros((h, p3), dtype=torch.uint8, device=dev) for _ in range(N)]
    for K in (1, 2, 4, 8):
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        exs = [capi.make_exec(s.cuda_stream) for s in streams]
        def step():
            for i, (s, d) in enumerate(zip(src, dst)):
                capi.convert(exs[i % K], capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, w, h, [(s.data_ptr(), p1), (s.data_ptr() + h * p1, p1)], [(d.data_ptr(), p3)])
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 10
        for _ in range(R):
            step()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) * 1e6 / (R * N)
        print(f"[streams] {w}x{h} NV12->RGB per-frame launches over {K} stream(s): {us:6.2f} us/frame  {w * h / us / 1e3:7.0f} Gpix/s  {4.5 * w * h / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
