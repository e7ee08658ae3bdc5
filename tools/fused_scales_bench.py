"""Fused NV12 -> bilinear -> RGB (vpf_convert_resize_batch, 16 frames per dispatch) at several scale factors: exact odd
integer ratios take the exact-alignment shortcuts, the others convert all four taps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = 16
for (sw, sh, dw, dh) in ((3840, 2160, 1280, 720), (3840, 2160, 1920, 1080), (3840, 2160, 1600, 900), (1920, 1080, 1280, 720),
                         (1920, 1080, 640, 360), (1920, 1080, 224, 224), (1920, 1080, 3840, 2160)):
    sp, dp = (sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256
    src = [torch.randint(0, 256, (sh * 3 // 2, sp), dtype=torch.uint8, device=dev) for _ in range(N)]
    dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(N)]
    batch = capi.make_batch([([(s.data_ptr(), sp), (s.data_ptr() + sh * sp, sp)], [(d.data_ptr(), dp)]) for s, d in zip(src, dst)])
    fn = lambda: capi.convert_resize_batch(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, sw, sh, dw, dh, batch)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * N)
    print(f"[fused] {sw}x{sh} -> {dw}x{dh} (x{sw / dw:.3g}): {us:6.2f} us/frame  {sw * sh / us / 1e3:7.0f} Gpix/s(src)  {dw * dh / us / 1e3:6.0f} Gpix/s(dst)", flush=True)
