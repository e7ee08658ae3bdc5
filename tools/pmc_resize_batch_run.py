"""Workload for rocprofv3 (tools/gpu_pmc_resize_batch.sh): vpf_resize_batch on packed RGB, 32 frames per dispatch.
Usage: pmc_resize_batch_run.py [sw sh dw dh [interp]]   (default 1920 1080 1280 720 lanczos)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

sw, sh, dw, dh = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (1920, 1080, 1280, 720)
interp = int(sys.argv[5]) if len(sys.argv) > 5 else 2
if os.environ.get("VPF_PMC_MFMA"):  # 1: keep the tiled Lanczos kernel
    capi.set_tuning(capi.TUNE_RESIZE_MFMA, int(os.environ["VPF_PMC_MFMA"], 0))
dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = 32
CHN = 1 if os.environ.get("VPF_PMC_FMT") == "Y" else 3  # VPF_PMC_FMT=Y: one 1-channel plane instead of packed RGB
sp, dp = (CHN * sw + 255) // 256 * 256, (CHN * dw + 255) // 256 * 256
src = [torch.randint(0, 256, (sh, sp), dtype=torch.uint8, device=dev) for _ in range(N)]
dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(N)]
batch = capi.make_batch([([(s.data_ptr(), sp)], [(d.data_ptr(), dp)]) for s, d in zip(src, dst)])
for _ in range(6):
    capi.resize_batch(ex, capi.Y if CHN == 1 else capi.RGB, interp, sw, sh, dw, dh, batch)
torch.cuda.synchronize()
