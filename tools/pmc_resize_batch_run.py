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
# a ring of batches past the 256 MiB Infinity Cache (one 32-frame batch of 1080p -> 720p RGB is 287 MB: dispatched again and again it is
# served partly from that cache and runs ~15 % faster than in a real stream of frames — the round-2 / early round-3 profiles had that flaw)
NBATCH = max(1, int(900e6 // (N * (sh * sp + dh * dp))) + 1)
batches = []
keep = []
for _ in range(NBATCH):
    src = [torch.randint(0, 256, (sh, sp), dtype=torch.uint8, device=dev) for _ in range(N)]
    dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(N)]
    keep.append((src, dst))
    batches.append(capi.make_batch([([(s.data_ptr(), sp)], [(d.data_ptr(), dp)]) for s, d in zip(src, dst)]))
for i in range(max(6, 2 * NBATCH)):
    capi.resize_batch(ex, capi.Y if CHN == 1 else capi.RGB, interp, sw, sh, dw, dh, batches[i % NBATCH])
torch.cuda.synchronize()
