"""fused_nb.py: the fused convert + resize strip kernel with 1 .. 4 bands per wave (VPF_LAB_FUSED_NB, read by the launcher at every launch): us/frame, medians of five passes, interleaved"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
from resize_batch_bench import surf, timed
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (sw, sh, dw, dh) in ((1920, 1080, 1280, 720), (1920, 1080, 3840, 2160), (1280, 720, 1920, 1080), (3840, 2160, 2560, 1440)):
    ring = 64 if dw < 2000 else 32
    S = [surf(capi.NV12, sw, sh, True) for _ in range(ring)]
    D = [surf(capi.RGB, dw, dh, False) for _ in range(ring)]
    batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + N]]) for i in range(0, ring, N)]
    res = {}
    for rep in range(3):
        for nb in ("policy", 1, 2, 3, 4):
            if nb == "policy":
                os.environ.pop("VPF_LAB_FUSED_NB", None)
            else:
                os.environ["VPF_LAB_FUSED_NB"] = str(nb)
            t = timed(lambda: [capi.convert_resize_batch(ex, capi.NV12, capi.RGB, 1, 0, sw, sh, dw, dh, b) for b in batches], 5, 3) / ring
            res[nb] = min(res.get(nb, 1e9), t)
    os.environ.pop("VPF_LAB_FUSED_NB", None)
    print(f"[fused-nb] {sw}x{sh}->{dw}x{dh} n={N}: " + " | ".join(f"nb={k}: {v:.2f}" for k, v in res.items()), flush=True)
    del S, D, batches
    torch.cuda.empty_cache()
