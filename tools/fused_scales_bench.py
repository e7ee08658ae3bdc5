"""Fused NV12 -> bilinear -> RGB (vpf_convert_resize_batch) at several scale factors: exact odd integer ratios take the exact-alignment
shortcuts, the others convert all four taps.  32 frames per dispatch (FUSED_N to change), rings of frames past the 256 MiB Infinity Cache,
median of three passes — the method of tools/resize_batch_bench.py (until round 4 this tool re-dispatched ONE 16-frame batch: half the
frames per dispatch, and at 1080p a working set inside the cache).  Fractions are algorithmic bytes (whole NV12 source + RGB destination)
per time against 8 TB/s; down-scales beyond 2 x skip source rows, so theirs can pass the bytes they really move."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from videoprocessingframework_amd import capi
from resize_batch_bench import timed

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = int(os.environ.get("FUSED_N", "32"))
for (sw, sh, dw, dh) in ((3840, 2160, 1280, 720), (3840, 2160, 1920, 1080), (3840, 2160, 1600, 900), (1920, 1080, 1280, 720),
                         (1920, 1080, 640, 360), (1920, 1080, 224, 224), (1920, 1080, 3840, 2160), (1280, 720, 1920, 1080)):
    sp, dp = (sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256
    nbytes = sw * sh * 3 // 2 + 3 * dw * dh
    ring = max(N, min(256, int(600e6 // nbytes) // N * N))
    src = [torch.randint(0, 256, (sh * 3 // 2, sp), dtype=torch.uint8, device=dev) for _ in range(ring)]
    dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(ring)]
    io = [([(s.data_ptr(), sp), (s.data_ptr() + sh * sp, sp)], [(d.data_ptr(), dp)]) for s, d in zip(src, dst)]
    batches = [capi.make_batch(io[i:i + N]) for i in range(0, ring, N)]
    us = timed(lambda: [capi.convert_resize_batch(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, sw, sh, dw, dh, b) for b in batches], 5) / ring
    print(f"[fused] {sw}x{sh} -> {dw}x{dh} (x{sw / dw:.3g}): {us:6.2f} us/frame = {nbytes / us / 8e6:.2f} of 8 TB/s  {sw * sh / us / 1e3:7.0f} Gpix/s(src)  "
          f"{dw * dh / us / 1e3:6.0f} Gpix/s(dst)  ({N} frames per dispatch, ring {ring})", flush=True)
    del src, dst, batches
    torch.cuda.empty_cache()
