"""band_knob_sweep.py KNOB [KNOB ...]: us/frame of the batched BILINEAR resize (32 frames per dispatch, rings past the Infinity Cache) under
VPF_TUNE_RESIZE_BAND values (hex ok: rows | nb << 8 | 0x10000 = the persistent launch), passes interleaved over the knobs, minimum and median
of the passes per cell.  SWEEP_CASES="Y:1920x1080:1280x720,NV12:..." overrides the cases; SWEEP_INTERP=2 sweeps VPF_TUNE_RESIZE_MFMA on the
Lanczos kernels instead; SWEEP_LIB=path runs another build of the kernel library."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
if os.environ.get("SWEEP_LIB"):  # another build of the kernel library (same-box A/B of two builds: run alternately)
    capi.LIB_PATH = os.path.abspath(os.environ["SWEEP_LIB"])
argv, sys.argv = sys.argv, sys.argv[:1]
from resize_batch_bench import surf  # noqa: E402
sys.argv = argv

INTERP = int(os.environ.get("SWEEP_INTERP", "1"))
KEY = capi.TUNE_RESIZE_BAND if INTERP == 1 else capi.TUNE_RESIZE_MFMA
CASES = [("Y", 1920, 1080, 1280, 720), ("NV12", 1920, 1080, 1280, 720), ("YUV420", 1920, 1080, 1280, 720), ("RGB", 1920, 1080, 1280, 720),
         ("Y", 1280, 720, 1920, 1080), ("NV12", 3840, 2160, 1920, 1080), ("RGB", 1280, 720, 1920, 1080)]
if os.environ.get("SWEEP_CASES"):
    CASES = [(c.split(":")[0],) + tuple(int(v) for v in c.split(":")[1].split("x")) + tuple(int(v) for v in c.split(":")[2].split("x")) for c in os.environ["SWEEP_CASES"].split(",")]
PASSES = int(os.environ.get("SWEEP_PASSES", "3"))
NB = int(os.environ.get("SWEEP_N", "32"))  # frames per dispatch


def main():
    knobs = [int(k, 0) for k in sys.argv[1:]] or [0]
    ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
    print(f"[knobs] interp {INTERP}, {NB} frames per dispatch; cells: min / median of {PASSES} interleaved passes, us per frame")
    print("[knobs] " + " " * 12 + " | ".join(f"{c[0]:>6s} {c[1]}x{c[2]}->{c[3]}x{c[4]}" for c in CASES))
    res = {k: [] for k in knobs}
    for fname, sw, sh, dw, dh in CASES:
        fmt = getattr(capi, fname)
        ring = max(NB, min(256, int(600e6 // ((sw * sh + dw * dh) * 3)) // NB * NB))
        S = [surf(fmt, sw, sh, True) for _ in range(ring)]
        D = [surf(fmt, dw, dh, False) for _ in range(ring)]
        batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + NB]]) for i in range(0, ring, NB)]
        cell = {k: [] for k in knobs}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(PASSES + 1):  # the first pass warms up
            for k in knobs:
                if capi.set_tuning(KEY, k) < 0:
                    cell[k].append(float("nan")); continue
                for b in batches:
                    capi.resize_batch(ex, fmt, INTERP, sw, sh, dw, dh, b)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    for b in batches:
                        capi.resize_batch(ex, fmt, INTERP, sw, sh, dw, dh, b)
                e1.record(); torch.cuda.synchronize()
                cell[k].append(e0.elapsed_time(e1) * 1e3 / 5 / ring)
        capi.set_tuning(KEY, 0)
        for k in knobs:
            v = sorted(cell[k][1:])
            res[k].append((v[0], v[len(v) // 2]))
        del S, D, batches
        torch.cuda.empty_cache()
    for k in knobs:
        print(f"[knobs] {k:#9x}   " + " | ".join(f"{a:10.3f} / {b:6.3f}  " for a, b in res[k]), flush=True)


if __name__ == "__main__":
    main()
