"""Strong down-scales (horizontal factors ~2.2 .. 6: the resize in front of a network; SWEEP_THUMBS=1: both axes 3 .. 6, thumbnails), batched Lanczos-3: the matrix-core kernel with two-chunk
windows (policy; forced band heights 4 << 8 | r) against the tile kernel (VPF_TUNE_RESIZE_MFMA = 1).  us/frame, 32 frames per dispatch, rings past
the Infinity Cache, medians of three passes.  python tools/lanczos_k2_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from videoprocessingframework_amd import capi
from resize_batch_bench import surf, timed

ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
SHAPES = ((1920, 1080, 416, 416), (1920, 1080, 640, 480), (1920, 1080, 480, 480), (1280, 720, 416, 416), (1280, 720, 320, 320), (3840, 2160, 1440, 810),
          (3840, 2160, 1080, 1080), (2560, 1440, 640, 640))
if os.environ.get("SWEEP_THUMBS"):  # vertical factors of ~2.9 .. 6: half tiles
    SHAPES = ((1920, 1080, 480, 270), (1920, 1080, 384, 216), (3840, 2160, 960, 540), (3840, 2160, 1024, 576), (1920, 1080, 512, 288), (1280, 720, 224, 224),
              (2560, 1440, 640, 360), (1920, 1080, 416, 234))
if os.environ.get("SWEEP_SHAPES"):  # "sw,sh,dw,dh ..."
    SHAPES = tuple(tuple(int(v) for v in t.split(",")) for t in os.environ["SWEEP_SHAPES"].split())
RS = [int(v) for v in os.environ.get("SWEEP_R", "1 2 3 4 6 8").split()]
for fmt, fname in ((capi.RGB, "RGB"), (capi.Y, "Y"), (capi.NV12, "NV12")):
    for sw, sh, dw, dh in SHAPES:
        ring = max(32, min(128, int(600e6 // ((sw * sh + dw * dh) * 3)) // 32 * 32))
        S = [surf(fmt, sw, sh, True) for _ in range(ring)]
        D = [surf(fmt, dw, dh, False) for _ in range(ring)]
        batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + 32]]) for i in range(0, ring, 32)]
        nbytes = S[0][2] + D[0][2]
        out = []
        for name, knob in [("policy", 0), ("tile", 1)] + [(f"r{r}", (4 << 8) | r) for r in RS]:
            capi.set_tuning(capi.TUNE_RESIZE_MFMA, knob)
            t = timed(lambda: [capi.resize_batch(ex, fmt, 2, sw, sh, dw, dh, b) for b in batches], 5) / ring
            out.append(f"{name} {t:5.2f}")
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
        print(f"[lz-k2] {fname:4s} {sw}x{sh}->{dw}x{dh} ({sw / dw:.2f} x {sh / dh:.2f}), floor {nbytes / 8e6:.2f}: " + " | ".join(out), flush=True)
        del S, D, batches
        torch.cuda.empty_cache()
