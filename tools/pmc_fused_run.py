"""Workload for rocprofv3 (tools/gpu_pmc_fused.sh): the fused NV12 -> bilinear -> RGB batch, 32 frames per dispatch.
Usage: pmc_fused_run.py [sw sh dw dh]   (default 3840 2160 1280 720)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

sw, sh, dw, dh = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (3840, 2160, 1280, 720)
dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = 32
sp, dp = (sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256
src = [torch.randint(0, 256, (sh * 3 // 2, sp), dtype=torch.uint8, device=dev) for _ in range(N)]
dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(N)]
batch = capi.make_batch([([(s.data_ptr(), sp), (s.data_ptr() + sh * sp, sp)], [(d.data_ptr(), dp)]) for s, d in zip(src, dst)])
for _ in range(5):
    capi.convert_resize_batch(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, sw, sh, dw, dh, batch)
torch.cuda.synchronize()
