"""NV12 -> RGB_PLANAR, one launch per frame, kernel variants 37 (r16, one row per wave) and 8 (p16, one row pair per wave):
run under rocprofv3 --kernel-trace --stats to compare kernel durations (tools/gpu_planar_single.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
for (w, h) in ((3840, 2160), (1920, 1080), (1280, 720)):
    N = 16
    p1 = (w + 255) // 256 * 256
    src = [torch.randint(0, 256, (h * 3 // 2, p1), dtype=torch.uint8, device=dev) for _ in range(N)]
    dst = [torch.zeros((3 * h, p1), dtype=torch.uint8, device=dev) for _ in range(N)]
    for variant in (37, 8):
        capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
        for _ in range(4):
            for s, d in zip(src, dst):
                capi.convert(ex, capi.NV12, capi.RGB_PLANAR, capi.BT_709, capi.MPEG, w, h, [(s.data_ptr(), p1), (s.data_ptr() + h * p1, p1)],
                             [(d.data_ptr() + i * h * p1, p1) for i in range(3)])
        torch.cuda.synchronize()
capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0)
