"""Times the gather-form resize kernels: vpf_resize on RGB_32F surfaces (the float resizer of PySurfaceResizer) and the
8-bit Lanczos gather form (tuning 9 = generic kernels), per-frame dispatch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
RING, STEPS = 8, 5


def timed(step):
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(STEPS):
        step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (STEPS * RING)


for (sw, sh, dw, dh) in ((1920, 1080, 1280, 720), (1280, 720, 1920, 1080)):
    sp, dp = (12 * sw + 255) // 256 * 256, (12 * dw + 255) // 256 * 256
    src = [torch.rand((sh, sp // 4), dtype=torch.float32, device=dev) for _ in range(RING)]
    dst = [torch.zeros((dh, dp // 4), dtype=torch.float32, device=dev) for _ in range(RING)]
    for name, interp in (("nearest", capi.INTERP_NEAREST), ("bilinear", capi.INTERP_LINEAR), ("lanczos3", capi.INTERP_LANCZOS3)):
        us = timed(lambda: [capi.resize(ex, capi.RGB_32F, interp, sw, sh, [(s.data_ptr(), sp)], dw, dh, [(d.data_ptr(), dp)]) for s, d in zip(src, dst)])
        print(f"[resize f32] {sw}x{sh} -> {dw}x{dh} RGB_32F {name:9s}: {us:7.1f} us/frame  {dw * dh / us / 1e3:7.2f} Gpix/s(dst)", flush=True)
    sp8, dp8 = (3 * sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256
    s8 = [torch.randint(0, 256, (sh, sp8), dtype=torch.uint8, device=dev) for _ in range(RING)]
    d8 = [torch.zeros((dh, dp8), dtype=torch.uint8, device=dev) for _ in range(RING)]
    prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 9)
    us = timed(lambda: [capi.resize(ex, capi.RGB, capi.INTERP_LANCZOS3, sw, sh, [(s.data_ptr(), sp8)], dw, dh, [(d.data_ptr(), dp8)]) for s, d in zip(s8, d8)])
    capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
    print(f"[resize u8 ] {sw}x{sh} -> {dw}x{dh} RGB lanczos3 gather form: {us:7.1f} us/frame  {dw * dh / us / 1e3:7.2f} Gpix/s(dst)", flush=True)
