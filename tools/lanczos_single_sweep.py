"""One Lanczos resize dispatch per frame (what an unmodified PySurfaceResizer.Execute loop issues): the matrix-core kernel (VPF_TUNE_RESIZE_MFMA
0x800 / 0x400 forced) against the tile kernel (1 = matrix-core kernel off) and the policy (0), over source sizes and scale factors, RGB and Y.
Rings past the Infinity Cache, medians of three passes.  The launch policy's single-frame rule (k_resize.hip: launch_resize) is fitted to this
table.  python tools/lanczos_single_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from videoprocessingframework_amd import capi
from resize_batch_bench import surf, timed

ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
SRC = ((640, 360), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160))
FACTORS = (0.4, 0.5, 2.0 / 3.0, 0.8, 1.25, 1.5, 2.0)   # destination / source
KNOBS = (("policy", 0), ("mfma", 0x40000), ("tile", 1)) if not os.environ.get("SWEEP_POLICY_ONLY") else (("policy", 0),)
for fmt, fname in ((capi.RGB, "RGB"), (capi.Y, "Y")):
    ch = 3 if fmt == capi.RGB else 1
    for sw, sh in SRC:
        for f in FACTORS:
            dw, dh = int(sw * f) // 2 * 2, int(sh * f) // 2 * 2
            if dw * dh > 3840 * 2160 or dw < 64:
                continue
            per = (sw * sh + dw * dh) * ch
            ring = max(8, min(128, int(600e6 // per)))
            S = [surf(fmt, sw, sh, True) for _ in range(ring)]
            D = [surf(fmt, dw, dh, False) for _ in range(ring)]
            planes = [(capi.planes(s[1]), capi.planes(d[1])) for s, d in zip(S, D)]
            out = []
            for name, knob in KNOBS:
                capi.set_tuning(capi.TUNE_RESIZE_MFMA, knob)
                t = timed(lambda: [capi.resize(ex, fmt, 2, sw, sh, s, dw, dh, d) for s, d in planes], 3) / ring
                out.append(f"{name} {t:6.2f}")
            capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
            print(f"[lz-single] {fname:3s} {sw}x{sh}->{dw}x{dh} ({f:.2f}) {per / 1e6:6.2f} MB: " + " | ".join(out) + " us/frame", flush=True)
            del S, D, planes
            torch.cuda.empty_cache()
