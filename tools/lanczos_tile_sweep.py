"""Tile-shape sweep of the tiled Lanczos kernel (LanczosTileTask) on strong down-scales: us per frame, 32 frames per dispatch and one frame per dispatch, over
VPF_TUNE_RESIZE_TILE = rows per tile | waves per workgroup << 8.  python tools/lanczos_tile_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from resize_batch_bench import surf, timed  # noqa: E402

ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = 32
for fmt, fname, (sw, sh, dw, dh) in ((capi.RGB, "RGB", (1920, 1080, 416, 416)), (capi.RGB, "RGB", (1920, 1080, 608, 608)), (capi.RGB, "RGB", (1920, 1080, 224, 224)),
                                     (capi.YUV420, "YUV420", (1920, 1080, 224, 224)), (capi.RGB, "RGB", (3840, 2160, 1280, 704)), (capi.NV12, "NV12", (1920, 1080, 416, 416))):
    ring = 64
    S = [surf(fmt, sw, sh, True) for _ in range(ring)]
    D = [surf(fmt, dw, dh, False) for _ in range(ring)]
    batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + N]]) for i in range(0, ring, N)]
    ones = [capi.make_batch([(s[1], d[1])]) for s, d in zip(S, D)]
    res = {}
    for wpb in (8, 4):
        for ty in (0, 4, 8, 12, 16, 24, 32):
            shape = (ty | (wpb << 8)) if ty else 0
            if ty == 0 and wpb == 4:
                continue
            if capi.set_tuning(capi.TUNE_RESIZE_TILE, shape) < 0:
                continue
            tb = timed(lambda: [capi.resize_batch(ex, fmt, 2, sw, sh, dw, dh, b) for b in batches], 3) / ring
            t1 = timed(lambda: [capi.resize_batch(ex, fmt, 2, sw, sh, dw, dh, b) for b in ones], 2) / ring
            res["policy" if not shape else f"ty{ty}w{wpb}"] = (tb, t1)
    capi.set_tuning(capi.TUNE_RESIZE_TILE, 0)
    print(f"[lz-tile-sweep] {fname:6s} {sw}x{sh}->{dw}x{dh}: " + " ".join(f"{k}={v[0]:.2f}/{v[1]:.2f}" for k, v in res.items()), flush=True)
    del S, D, batches, ones
    torch.cuda.empty_cache()
