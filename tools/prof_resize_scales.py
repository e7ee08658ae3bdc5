"""Workload for `rocprofv3 --kernel-trace`: vpf_resize (bilinear, packed RGB) at the size pairs given as sw,sh,dw,dh
arguments after the tuning value; kernel durations from the trace are the honest comparison (per-frame Python loops sit on
the ~5 us host floor)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi
dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, int(sys.argv[1]))
interp = {"nearest": capi.INTERP_NEAREST, "bilinear": capi.INTERP_LINEAR, "lanczos": capi.INTERP_LANCZOS3}[sys.argv[2]]
sizes = [tuple(int(t) for t in a.split(",")) for a in sys.argv[3:]] or [(3840, 2160, 1280, 720)]
for (sw, sh, dw, dh) in sizes:
    sp, dp = (3 * sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256
    src = [torch.randint(0, 256, (sh, sp), dtype=torch.uint8, device=dev) for _ in range(4)]
    dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(4)]
    for _ in range(3):
        for s, d in zip(src, dst):
            capi.resize(ex, capi.RGB, interp, sw, sh, [(s.data_ptr(), sp)], dw, dh, [(d.data_ptr(), dp)])
    torch.cuda.synchronize()
