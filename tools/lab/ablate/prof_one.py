"""prof_one.py LIB: per-wave phase timers of the instrumented matrix-core Lanczos kernel (tools/lab/ablate/k_lanczos_mfma_prof.hip.txt built into LIB):
one 32-frame dispatch (or one frame with N=1) of a case, then mean / p10 / p90 over the first 2048 waves of: wave lifetime, setup, staging wait
(vmcnt + ds_write + sync), pass 1, pass 2 arithmetic, out transpose + stores, group barriers.  Cycles of the shader clock counter."""
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
    buf = (C.c_ulonglong * (2048 * 8))()
    assert L.vpf_lab_lzm_prof(buf, 2048 * 8) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8).astype(np.float64)
    a = a[a[:, 0] > 0]
    names = ["lifetime", "setup", "stage wait", "pass 1", "pass 2 arith", "transpose+store", "group barriers"]
    t0 = a[:, 7]
    print(f"[prof] {fname} {sw}x{sh}->{dw}x{dh} n={N}: {len(a)} waves recorded; start spread {np.ptp(t0):.0f} cycles")
    for i, nm in enumerate(names):
        v = a[:, i]
        print(f"[prof]   {nm:16s} mean {v.mean():10.0f}  p10 {np.percentile(v, 10):10.0f}  p90 {np.percentile(v, 90):10.0f}  ({100 * v.mean() / a[:, 0].mean():5.1f} % of lifetime)")
    del S, D
    torch.cuda.empty_cache()
