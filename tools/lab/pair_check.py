"""first contact of the two-role Lanczos kernel (VPF_TUNE_RESIZE_MFMA | 0x20000): a few shapes against the oracle, then timings against the one-role kernel"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle
from videoprocessingframework_amd import capi
from gpu_util import DevPlanes
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
bad = 0
for fmt, sw, sh, dw, dh in [("RGB", 640, 360, 427, 240), ("RGB", 1920, 200, 1280, 133), ("Y", 997, 161, 333 * 2, 107), ("NV12", 1280, 720, 854, 480), ("RGB", 320, 180, 640, 360),
                            ("RGB", 1283, 211, 857, 140), ("YUV420", 642, 362, 500, 270), ("RGB", 1280, 200, 640, 100), ("RGB", 1920, 1080, 1280, 720), ("RGB", 700, 90, 2000, 257)]:
    for knob in (0x20000, 0x20000 | 2, 0x20000 | 5, 0x20000 | 1, 0x30000 | 3):
        for n in (1, 3):
            f, of = getattr(capi, fmt), getattr(oracle, fmt)
            srcs = [oracle.synth(of, sw, sh, 4400 + i) for i in range(n)]
            S, D = [DevPlanes(p) for p in srcs], [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
            capi.set_tuning(capi.TUNE_RESIZE_MFMA, knob)
            capi.resize_batch(ex, f, 2, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
            capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
            torch.cuda.synchronize()
            for i in range(n):
                got, intact = D[i].download()
                want = oracle.resize(of, 2, sw, sh, srcs[i], dw, dh, oracle.FP32)[1]
                ok = intact and all(np.array_equal(g, w) for g, w in zip(got, want))
                if not ok:
                    bad += 1
                    diffs = [(int((g != w).sum()), np.argwhere(g != w)[:3].tolist()) for g, w in zip(got, want)]
                    print("MISMATCH", fmt, sw, sh, dw, dh, hex(knob), "n", n, "frame", i, "intact", intact, diffs, flush=True)
print("pair-check mismatches:", bad, flush=True)
