"""Sweep of the matrix-core Lanczos kernel's launch shape (VPF_TUNE_RESIZE_MFMA = N-tiles per wave << 8 | 16-row tiles per band) over the
batched cases the planner has to get right: us per frame for every (format, size pair, nt, tiles per band), 32 frames per dispatch.
python tools/lanczos_shape_sweep.py [frames [passes]]   (SWEEP_SIZES="1280x720:1920x1080,..." overrides the size pairs, SWEEP_Y=1 sweeps a 1-channel plane)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from resize_batch_bench import surf, timed  # noqa: E402  (its module-level sweep is guarded below)

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 3
FMTS = ((capi.RGB, "RGB"), (capi.NV12, "NV12"), (capi.YUV420, "YUV420")) if not os.environ.get("SWEEP_Y") else ((capi.Y, "Y"),)  # SWEEP_Y=1: one 1-channel plane
SIZES = ((1920, 1080, 1280, 720), (3840, 2160, 1920, 1080), (1280, 720, 1920, 1080))
if os.environ.get("SWEEP_SIZES"):
    SIZES = tuple(tuple(int(v) for v in c.replace(":", "x").split("x")) for c in os.environ["SWEEP_SIZES"].split(","))
for fmt, fname in FMTS:
    for (sw, sh, dw, dh) in SIZES:
        ring = max(N, min(128, int(600e6 // (sw * sh * 3 + dw * dh * 3)) // N * N))
        S = [surf(fmt, sw, sh, True) for _ in range(ring)]
        D = [surf(fmt, dw, dh, False) for _ in range(ring)]
        batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + N]]) for i in range(0, ring, N)]
        tiles = (dh + 15) // 16
        res = {}
        shapes = [0]
        for nt in (8, 4):
            for r in sorted({2, 3, 4, 6, 8, 12, 16, (tiles + 7) // 8, (tiles + 5) // 6, (tiles + 4) // 5, (tiles + 3) // 4, (tiles + 2) // 3, (tiles + 1) // 2, tiles} - {1}):
                if r <= 64:
                    shapes.append((nt << 8) | r)
        for rep in range(PASSES):  # whole passes over the shapes, the minimum per shape: clock / thermal drift over a pass is several percent
            for shape in (shapes if rep % 2 == 0 else shapes[::-1]):
                capi.set_tuning(capi.TUNE_RESIZE_MFMA, shape)
                t = timed(lambda: [capi.resize_batch(ex, fmt, 2, sw, sh, dw, dh, b) for b in batches], 3, 1) / ring
                key = "policy" if shape == 0 else f"nt{shape >> 8} r{shape & 0xff}"
                res[key] = min(res.get(key, 1e9), t)
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
        best = min(res, key=res.get)
        print(f"[lzm-sweep] {fname:6s} {sw}x{sh}->{dw}x{dh} n={N}: policy {res['policy']:.2f} us/frame | best {best} {res[best]:.2f} | " +
              " ".join(f"{k}={v:.2f}" for k, v in res.items() if k != "policy"), flush=True)
        del S, D, batches
        torch.cuda.empty_cache()
