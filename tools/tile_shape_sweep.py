"""Sweep of the tiled resize kernels' shape (rows per tile x waves per workgroup, VPF_TUNE_RESIZE_TILE) for a few (size pair, filter)
cases.  Run under `rocprofv3 --kernel-trace`: tools/tile_shape_post.py pairs the launch log this script writes with the trace's kernel
durations (event timing of back-to-back launches cannot see below the ~5 us launch floor).  Usage: tile_shape_sweep.py out.json [quick]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = 6
CASES = [(1920, 1080, 1280, 720, 2), (1920, 1080, 3840, 2160, 2), (1920, 1080, 3840, 2160, 1), (1280, 720, 1920, 1080, 2), (1920, 1080, 416, 416, 2), (3840, 2160, 1920, 1088, 2)]
SHAPES = [(0, 0)] + [(ty, wpb) for wpb in (4, 8) for ty in (8, 16, 24, 32, 48, 64)]
log = []
for (sw, sh, dw, dh, interp) in CASES:
    sp, dp = (3 * sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256
    src = [torch.randint(0, 256, (sh, sp), dtype=torch.uint8, device=dev) for _ in range(N)]
    dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(N)]
    ref = None
    for (ty, wpb) in SHAPES:
        if capi.set_tuning(capi.TUNE_RESIZE_TILE, ty | (wpb << 8)) < 0:
            continue
        for s, d in zip(src, dst):
            capi.resize(ex, capi.RGB, interp, sw, sh, [(s.data_ptr(), sp)], dw, dh, [(d.data_ptr(), dp)])
        torch.cuda.synchronize()
        out = dst[0].clone()
        if ref is None:
            ref = out
        same = bool(torch.equal(out, ref))
        log.append({"case": f"{sw}x{sh}->{dw}x{dh} {'lanczos' if interp == 2 else 'bilinear'}", "ty": ty, "wpb": wpb, "launches": N, "same_pixels": same})
capi.set_tuning(capi.TUNE_RESIZE_TILE, 0)
json.dump(log, open(sys.argv[1], "w"))
print("launch groups:", len(log), "all identical pixels:", all(l["same_pixels"] for l in log))
