"""Workload for rocprofv3 (tools/gpu_pmc_resize.sh): vpf_resize on packed RGB, per-frame dispatch over a ring of 8.
Usage: pmc_resize_run.py [sw sh dw dh [interp]]   (default 3840 2160 2560 1440 bilinear; interp 0 nearest, 1 bilinear, 2 lanczos3)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

sw, sh, dw, dh = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (3840, 2160, 2560, 1440)
interp = {0: capi.INTERP_NEAREST, 1: capi.INTERP_LINEAR, 2: capi.INTERP_LANCZOS3}[int(sys.argv[5]) if len(sys.argv) > 5 else 1]
dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = 8
sp, dp = (3 * sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256
src = [torch.randint(0, 256, (sh, sp), dtype=torch.uint8, device=dev) for _ in range(N)]
dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(N)]
for _ in range(4):
    for s, d in zip(src, dst):
        capi.resize(ex, capi.RGB, interp, sw, sh, [(s.data_ptr(), sp)], dw, dh, [(d.data_ptr(), dp)])
torch.cuda.synchronize()
