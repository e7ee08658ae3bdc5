"""time_one.py LIB [frames]: us/frame of the batched Lanczos resize for three cases with an ablated kernel library (capi.LIB_PATH swapped before the first call)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
LIBP = os.path.abspath(sys.argv[1])
capi.LIB_PATH = LIBP
sys.argv = sys.argv[:1]
from resize_batch_bench import surf, timed
N = 32
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
shape = int(os.environ.get("VPF_BENCH_MFMA", "0"), 0)
capi.set_tuning(capi.TUNE_RESIZE_MFMA, shape)
out = []
for fmt, fname, (sw, sh, dw, dh) in ((capi.RGB, "RGB", (1920, 1080, 1280, 720)), (capi.RGB, "RGB", (3840, 2160, 1920, 1080)), (capi.NV12, "NV12", (1920, 1080, 1280, 720)), (capi.RGB, "RGB", (1280, 720, 1920, 1080))):
    ring = 64
    S = [surf(fmt, sw, sh, True) for _ in range(ring)]
    D = [surf(fmt, dw, dh, False) for _ in range(ring)]
    batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + N]]) for i in range(0, ring, N)]
    t = timed(lambda: [capi.resize_batch(ex, fmt, 2, sw, sh, dw, dh, b) for b in batches], 5) / ring
    out.append(f"{fname} {sw}->{dw}: {t:.2f}")
    del S, D, batches
    torch.cuda.empty_cache()
print(f"[ablate] {os.path.basename(LIBP):24s} shape {shape:#x}  " + " | ".join(out), flush=True)
