"""time_lib.py LIB: us/frame of batched bilinear resizes, the fused convert + resize and batched RGB -> YUV444 with the kernel library LIB
(capi.LIB_PATH swapped before the first call) — for same-box A/B runs of two builds: run it alternately with both libraries."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
LIBP = os.path.abspath(sys.argv[1])
capi.LIB_PATH = LIBP
sys.argv = sys.argv[:1]
from resize_batch_bench import surf, timed
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
out = []
for fmt, fname, (sw, sh, dw, dh) in ((capi.RGB, "RGB", (1920, 1080, 1280, 720)), (capi.RGB, "RGB", (1920, 1080, 3840, 2160)), (capi.RGB, "RGB", (1280, 720, 1920, 1080)),
                                     (capi.NV12, "NV12", (1920, 1080, 1280, 720)), (capi.YUV420, "YUV420", (1920, 1080, 1280, 720))):
    ring = 64 if dw < 3000 else 32
    S = [surf(fmt, sw, sh, True) for _ in range(ring)]
    D = [surf(fmt, dw, dh, False) for _ in range(ring)]
    batch = capi.make_batch([(s[1], d[1]) for s, d in zip(S, D)])
    t = timed(lambda: capi.resize_batch(ex, fmt, 1, sw, sh, dw, dh, batch), 5, 5) / ring
    out.append(f"{fname} {sw}->{dw}: {t:.2f}")
    del S, D, batch
    torch.cuda.empty_cache()
for (sw, sh, dw, dh) in ((1920, 1080, 1280, 720), (1920, 1080, 3840, 2160)):
    ring = 32
    S = [surf(capi.NV12, sw, sh, True) for _ in range(ring)]
    D = [surf(capi.RGB, dw, dh, False) for _ in range(ring)]
    batch = capi.make_batch([(s[1], d[1]) for s, d in zip(S, D)])
    t = timed(lambda: capi.convert_resize_batch(ex, capi.NV12, capi.RGB, 1, 0, sw, sh, dw, dh, batch), 5, 5) / ring
    out.append(f"fused {sw}->{dw}: {t:.2f}")
    del S, D, batch
    torch.cuda.empty_cache()
print(f"[ab] {os.path.basename(LIBP):28s} " + " | ".join(out), flush=True)
