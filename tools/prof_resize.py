"""Launches the resize / fused kernels (LDS-staged and gather variants) for rocprofv3 --kernel-trace --stats."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
for variant in (0, 9):
    for name in ("resize_4k_720p", "fused_4k_720p"):
        wl = bench.Workload(name, dev, 8, variant, "single")
        capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
        for _ in range(5):
            wl.step()
        torch.cuda.synchronize()
        del wl
        torch.cuda.empty_cache()
capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0)
