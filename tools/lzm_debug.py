"""First-contact diagnostics for the matrix-core Lanczos kernel (k_lanczos_mfma.hip): a few small cases against the oracle with a
structured report of WHERE mismatches are (by N-tile, byte within the tile, row within the 16-row tile), so that one GPU visit localises
a layout / constant error.  python tools/lzm_debug.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from videoprocessingframework_amd import capi
from gpu_util import DevPlanes, stream_handle

ok = True
for shape in ((8 << 8) | 2, (4 << 8) | 1):
    capi.set_tuning(capi.TUNE_RESIZE_MFMA, shape)
    for fmt, sw, sh, dw, dh in (("Y", 256, 64, 256, 64), ("Y", 96, 54, 64, 36), ("RGB", 96, 54, 64, 36), ("RGB", 64, 36, 96, 54), ("NV12", 128, 72, 64, 36), ("RGB", 640, 360, 427, 240)):
        f, of = getattr(capi, fmt), getattr(oracle, fmt)
        for kind in ("flat", "rand"):
            src = oracle.synth(of, sw, sh, 5)
            if kind == "flat":
                for p in src:
                    p[...] = 200
            s, d = DevPlanes(src), DevPlanes(oracle.alloc(of, dw, dh))
            os.environ["VPF_HIP_LOG"] = "2"
            capi.resize_batch(capi.make_exec(stream_handle()), f, 2, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc())]))
            torch.cuda.synchronize()
            got, intact = d.download()
            _, want = oracle.resize(of, 2, sw, sh, src, dw, dh, oracle.FP32)
            for pi, (g, w) in enumerate(zip(got, want)):
                bad = np.argwhere(g != w)
                tag = f"shape {shape:#x} {fmt} {sw}x{sh}->{dw}x{dh} {kind} plane {pi}"
                if len(bad) == 0 and intact:
                    print("OK  ", tag)
                    continue
                ok = False
                print("FAIL", tag, "intact", intact, "mismatches", len(bad), "of", g.size)
                ys, xs = bad[:, 0], bad[:, 1]
                print("   rows mod 16 histogram:", np.bincount(ys % 16, minlength=16).tolist())
                print("   byte mod 16 histogram:", np.bincount(xs % 16, minlength=16).tolist())
                print("   N-tile (byte // 16) histogram (first 24):", np.bincount(xs // 16)[:24].tolist())
                print("   row tile (y // 16) histogram:", np.bincount(ys // 16).tolist())
                d8 = (g.astype(int) - w.astype(int))
                print("   diff min / max / mean:", d8.min(), d8.max(), float(d8.mean()))
                for (y, x) in bad[:12]:
                    print(f"   y {y} byte {x}: got {g[y, x]} want {w[y, x]}")
capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
print("ALL OK" if ok else "SOME FAILED")
