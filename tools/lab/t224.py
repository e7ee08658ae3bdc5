import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from videoprocessingframework_amd import capi
from resize_batch_bench import surf, timed
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
for fmt, name in ((capi.YUV420, "YUV420"), (capi.NV12, "NV12"), (capi.RGB, "RGB"), (capi.Y, "Y")):
    for (sw, sh, dw, dh) in ((1920, 1080, 224, 224), (1280, 720, 224, 224), (3840, 2160, 416, 416)):
        ring = 64
        S = [surf(fmt, sw, sh, True) for _ in range(ring)]; D = [surf(fmt, dw, dh, False) for _ in range(ring)]
        batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + 32]]) for i in range(0, ring, 32)]
        nb = S[0][2] + D[0][2]
        out = []
        for interp in (1, 2):
            t = timed(lambda: [capi.resize_batch(ex, fmt, interp, sw, sh, dw, dh, b) for b in batches], 5) / ring
            out.append(f"{'bilinear' if interp == 1 else 'lanczos'} {t:.2f} us ({nb / t / 8e6:.2f})")
        print(f"[t224] {name} {sw}x{sh}->{dw}x{dh} floor {nb / 8e6:.2f}: " + " | ".join(out), flush=True)
        del S, D, batches; torch.cuda.empty_cache()
