"""Times vpf_resize for the three interpolation modes on 4K -> 720p and 1080p -> 720p packed RGB (per-frame dispatch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
RING, STEPS = 8, 5
if len(sys.argv) > 1:
    capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, int(sys.argv[1]))  # 43: tiled kernel for every bilinear resize, 40: never
for (sw, sh, dw, dh) in ((3840, 2160, 1280, 720), (1920, 1080, 1280, 720), (1920, 1080, 3840, 2160)):
    sp, dp = (3 * sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256
    src = [torch.randint(0, 256, (sh, sp), dtype=torch.uint8, device=dev) for _ in range(RING)]
    dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(RING)]
    for name, interp in (("nearest", capi.INTERP_NEAREST), ("bilinear", capi.INTERP_LINEAR), ("lanczos3", capi.INTERP_LANCZOS3)):
        def step():
            for s, d in zip(src, dst):
                capi.resize(ex, capi.RGB, interp, sw, sh, [(s.data_ptr(), sp)], dw, dh, [(d.data_ptr(), dp)])
        step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(STEPS):
            step()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (STEPS * RING)
        print(f"[resize] {sw}x{sh} -> {dw}x{dh} RGB {name:9s}: {us:7.1f} us/frame  {dw * dh / us / 1e3:7.2f} Gpix/s(dst)", flush=True)
