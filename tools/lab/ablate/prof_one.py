"""prof_one.py LIB: per-wave phase timers of the instrumented matrix-core Lanczos kernel (k_lanczos_mfma.hip compiled with -DVPF_LZM_PROF=1 into LIB:
tools/lab/ablate/build.sh prof): one 32-frame dispatch (or one frame with PROF_N=1) of a case, then mean / p10 / p90 over the first 2048
waves of: wave lifetime, setup, and per phase of the march the cycles of the shader clock counter a wave spent there."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
LIBP = os.path.abspath(sys.argv[1]); capi.LIB_PATH = LIBP
N = int(os.environ.get("PROF_N", "32"))
sys.argv = sys.argv[:1]
from resize_batch_bench import surf
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
L = capi.lib()
L.vpf_lab_lzm_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_uint]
for fmt, fname, (sw, sh, dw, dh) in ((capi.RGB, "RGB", (1920, 1080, 1280, 720)), (capi.RGB, "RGB", (3840, 2160, 1920, 1080)), (capi.NV12, "NV12", (1920, 1080, 1280, 720))):
    S = [surf(fmt, sw, sh, True) for _ in range(N)]
    D = [surf(fmt, dw, dh, False) for _ in range(N)]
    b = capi.make_batch([(s[1], d[1]) for s, d in zip(S, D)])
    for _ in range(3):
        capi.resize_batch(ex, fmt, 2, sw, sh, dw, dh, b)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2048 * 16))()
    assert L.vpf_lab_lzm_prof(buf, 2048 * 16) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16).astype(np.float64)
    a = a[a[:, 0] > 0]
    names = ["lifetime", "setup", "p1: A-operand requests", "p1: out read-back", "p1: stage (wait+xor+write)", "p1: out store + fetch", "p1: MFMA loop", "p2: MFMA loop", "p2: tail / sync flush", "group barriers", "final flush"]
    t0 = a[:, 15]
    print(f"[prof] {fname} {sw}x{sh}->{dw}x{dh} n={N}: {len(a)} waves recorded; start spread {np.ptp(t0):.0f} cycles")
    for i, nm in enumerate(names):
        v = a[:, i]
        print(f"[prof]   {nm:16s} mean {v.mean():10.0f}  p10 {np.percentile(v, 10):10.0f}  p90 {np.percentile(v, 90):10.0f}  ({100 * v.mean() / a[:, 0].mean():5.1f} % of lifetime)")
    # who are the stragglers?  prof_id = 4 x linear launch block + wave; picture_order() (k_resize_common.h) maps the launch block to (bx, by, frame)
    if os.environ.get("PROF_MAP"):
        gx, gy, gz = [int(v) for v in os.environ["PROF_MAP"].split(",")]
        raw = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16).astype(np.float64)
        total = gx * gy * gz
        rows = []
        for pid in range(min(2048, total * 4)):
            if raw[pid, 0] <= 0:
                continue
            lin, wv = pid // 4, pid % 4
            xcd, idx, per, rem = lin & 7, lin >> 3, total >> 3, total & 7
            m = xcd * (per + 1) + idx if xcd < rem else rem * (per + 1) + (xcd - rem) * per + idx
            yz = m // gx
            rows.append((m - yz * gx, yz % gy, yz // gy, wv, xcd, raw[pid, 0], raw[pid, 15]))
        r = np.array(rows)
        for name, col in (("bx", 0), ("by", 1), ("wave", 3), ("xcd", 4)):
            print(f"[prof]   lifetime by {name}: " + "  ".join(f"{int(k)}:{r[r[:, col] == k][:, 5].mean():.0f}" for k in np.unique(r[:, col])))
        t0s = r[:, 6] - r[:, 6].min()
        print(f"[prof]   start offsets (cycles after the first wave): p50 {np.percentile(t0s, 50):.0f} p90 {np.percentile(t0s, 90):.0f} max {t0s.max():.0f}; end = start + lifetime: p50 {np.percentile(t0s + r[:, 5], 50):.0f} p90 {np.percentile(t0s + r[:, 5], 90):.0f} max {(t0s + r[:, 5]).max():.0f}")
    del S, D
    torch.cuda.empty_cache()
