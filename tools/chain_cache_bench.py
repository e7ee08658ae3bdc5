"""Does the intermediate of a per-frame chain (NV12 -> RGB -> RGB_PLANAR, the reference's sample chain) benefit from being
left in the 256 MiB Infinity Cache?  k1 = NV12->RGB with non-temporal stores (variant 8, the per-frame default) vs plain
stores (variant 12: allocating stores); k2 = RGB->RGB_PLANAR reads the intermediate right after.  Ring of 32 distinct frame sets."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
for (w, h) in ((3840, 2160), (1920, 1080)):
    N = 32
    p1, p3 = (w + 255) // 256 * 256, (3 * w + 255) // 256 * 256
    src = [torch.randint(0, 256, (h * 3 // 2, p1), dtype=torch.uint8, device=dev) for _ in range(N)]
    mid = [torch.zeros((h, p3), dtype=torch.uint8, device=dev) for _ in range(N)]
    out = [torch.zeros((3 * h, p1), dtype=torch.uint8, device=dev) for _ in range(N)]
    exr = capi.make_exec(torch.cuda.current_stream().cuda_stream, flags=capi.EXEC_DST_REUSED)
    for variant in (8, 12, -1):  # -1: default kernel policy with the VPF_EXEC_DST_REUSED hint on k1
        def step():
            for s, m, o in zip(src, mid, out):
                capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, max(variant, 0))
                capi.convert(exr if variant < 0 else ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, w, h, [(s.data_ptr(), p1), (s.data_ptr() + h * p1, p1)], [(m.data_ptr(), p3)])
                capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0)
                capi.convert(ex, capi.RGB, capi.RGB_PLANAR, capi.BT_709, capi.MPEG, w, h, [(m.data_ptr(), p3)], [(o.data_ptr() + i * h * p1, p1) for i in range(3)])
        step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * N)
        name = {8: "k1 NT loads + NT stores", 7: "k1 plain loads + plain stores", 12: "k1 NT loads, plain stores", 11: "k1 plain loads, NT stores",
                -1: "k1 default + VPF_EXEC_DST_REUSED"}[variant]
        print(f"[chain] {w}x{h} {name:32s}: {us:6.2f} us per frame (both kernels)", flush=True)
