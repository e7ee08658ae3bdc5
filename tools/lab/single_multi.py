"""one dispatch per frame, multi-plane formats, Lanczos: policy against the tile kernel (VPF_TUNE_RESIZE_MFMA = 1) and the matrix cores forced (0x40000)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
from resize_batch_bench import surf, timed
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
SH = ((1920, 1080, 224, 224), (1920, 1080, 416, 416), (1920, 1080, 480, 270), (1280, 720, 224, 224), (3840, 2160, 960, 540), (3840, 2160, 416, 416), (1920, 1080, 1280, 720),
      (3840, 2160, 1920, 1080), (1920, 1080, 640, 480), (1280, 720, 416, 416), (3840, 2160, 1440, 810))
for fmt, name in ((capi.NV12, "NV12"), (capi.YUV420, "YUV420"), (capi.RGB, "RGB"), (capi.Y, "Y")):
    for sw, sh, dw, dh in SH:
        ring = max(8, min(64, int(600e6 // ((sw * sh + dw * dh) * 3))))
        S = [surf(fmt, sw, sh, True) for _ in range(ring)]; D = [surf(fmt, dw, dh, False) for _ in range(ring)]
        planes = [(capi.planes(s[1]), capi.planes(d[1])) for s, d in zip(S, D)]
        out = []
        for nm, knob in (("policy", 0), ("tile", 1), ("mfma", 0x40000)):
            capi.set_tuning(capi.TUNE_RESIZE_MFMA, knob)
            t = timed(lambda: [capi.resize(ex, fmt, 2, sw, sh, s, dw, dh, d) for s, d in planes], 3) / ring
            out.append(f"{nm} {t:6.2f}")
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
        print(f"[single-multi] {name:6s} {sw}x{sh}->{dw}x{dh}: " + " | ".join(out), flush=True)
        del S, D, planes; torch.cuda.empty_cache()
