"""wave_times.py KIND FMT SW SH DW DH [--band K] [--mfma K] [--variant V] [--n 32] [--reps 5]
The occupancy of ONE dispatch over time, from per-wave records of the lab build (tools/lab/wt/build.sh, csrc/vpf_wave_times.h): how long
the waves live (histogram), how fast the dispatch fills the chip (ramp), how many rounds of waves it runs, and how long the chip idles
while the last waves finish (tail) — the evidence VERDICT r4 item 2 asks for before / after a change of launch shape.
KIND = bilinear | lanczos | fused (NV12 -> RGB, FMT ignored).  32 frames per dispatch, ring past the Infinity Cache, a train of dispatches
back to back as in the benchmarks; the hook keeps the LAST dispatch of the train (slot = flat wave index, no atomics)."""
import argparse, ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
capi.LIB_PATH = os.environ.get("VPF_WT_LIB", os.path.join(ROOT, "tools", "lab", "wt", "libvpfhip_wt.so"))
sys_argv = sys.argv
sys.argv = sys.argv[:1]
from resize_batch_bench import surf  # noqa: E402
sys.argv = sys_argv


def read_records(which):
    L = capi.lib()
    fn = getattr(L, "vpf_lab_wave_times_" + which)
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_void_p, C.c_uint32]
    cap = 1 << 17
    buf = np.zeros((cap, 8), np.uint32)
    n = fn(buf.ctypes.data, cap)
    return buf[:n]


def analyse(rec, label):
    """rec: the hook's buffer after a train of identical dispatches: slot i holds wave i of the LAST dispatch"""
    if not len(rec):
        print(f"[wave_times] {label}: the hook's buffer could not be read"); return None
    waves = int(rec[0, 6])
    if waves == 0 or waves > len(rec):
        print(f"[wave_times] {label}: {waves} waves per dispatch — no records / more than the buffer holds"); return None
    r = rec[:waves]
    written = (r[:, 0] != 0) & (r[:, 2] != 0)
    if not written.all():
        print(f"[wave_times]   ({(~written).sum()} of {waves} slots never written)")
    r = r[written]
    base = r[:, 0].min()
    t0 = (r[:, 0] - base).astype(np.int64)                 # 10-ns ticks (the 32-bit differences are exact: a dispatch lasts far less than 43 s)
    t1 = (r[:, 2] - base).astype(np.uint32).astype(np.int64)
    ok = (t1 >= t0) & (t1 - t0 < 1_000_000)
    if not ok.all():
        print(f"[wave_times]   ({(~ok).sum()} records dropped: a slot the last dispatch did not write)")
    t0, t1, r = t0[ok], t1[ok], r[ok]
    if not len(t0):
        print(f"[wave_times] {label}: no usable record; waves field {waves}, first slots: {rec[:4].tolist()}"); return None
    ids = r[:, 1]
    xcc, unit = (ids >> 8) & 15, (ids >> 16) & 0xfff
    tick = 0.01
    begin, end = t0.min(), t1.max()
    span = (end - begin) * tick
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.lexsort((-ev[:, 1], ev[:, 0]))]
    live, tt = np.cumsum(ev[:, 1]), ev[:, 0]
    peak = live.max()
    area = float(np.sum(live[:-1] * np.diff(tt))) * tick
    first_full = tt[np.argmax(live >= 0.9 * peak)]
    last_half = tt[len(live) - 1 - np.argmax(live[::-1] >= 0.5 * peak)]
    last_90 = tt[len(live) - 1 - np.argmax(live[::-1] >= 0.9 * peak)]
    life = (t1 - t0) * tick
    q = np.percentile(life, [1, 10, 50, 90, 99])
    slots = len(set(zip(xcc.tolist(), unit.tolist())))
    print(f"[wave_times] {label}: {len(t0)} waves in the dispatch, {slots} SIMDs on {len(set(xcc.tolist()))} XCDs, peak {int(peak)} waves alive = {len(t0) / peak:.2f} rounds")
    print(f"[wave_times]   wave life us: p1 {q[0]:.2f}  p10 {q[1]:.2f}  median {q[2]:.2f}  p90 {q[3]:.2f}  p99 {q[4]:.2f}  (p90 / p10 {q[3] / max(q[1], 1e-9):.2f})")
    hist, edges = np.histogram(life, bins=10)
    print("[wave_times]   histogram: " + "  ".join(f"{edges[i]:.1f}-{edges[i + 1]:.1f}:{hist[i]}" for i in range(len(hist))))
    print(f"[wave_times]   first wave -> last wave {span:.2f} us | ramp to 90 % of the peak {(first_full - begin) * tick:.2f} us | below 90 % of the peak for the last {(end - last_90) * tick:.2f} us, "
          f"below 50 % for the last {(end - last_half) * tick:.2f} us = {100 * (end - last_half) / max(end - begin, 1):.0f} % of the span | mean live waves / peak {area / (peak * span):.2f}")
    # the occupancy curve in ten steps of the span
    steps = np.linspace(begin, end, 11)
    occ = [float(np.mean([(t0 <= x) & (t1 > x) for x in np.linspace(steps[i], steps[i + 1], 8, endpoint=False)], axis=0).sum()) / peak for i in range(10)]
    print("[wave_times]   live waves / peak over the span, in tenths: " + " ".join(f"{o:.2f}" for o in occ))
    marks = []
    for i in range(3):
        m = r[:, 3 + i]
        have = m != 0
        if have.sum() > len(m) // 2:
            d = ((m[have] & ~np.uint32(1)) - base).astype(np.uint32).astype(np.int64) - t0[have]
            marks.append((i, np.percentile(d * tick, [10, 50, 90])))
    if marks:
        print("[wave_times]   marks after the wave's start, us (p10 / median / p90): " + " | ".join(f"mark {i}: {p[0]:.2f} / {p[1]:.2f} / {p[2]:.2f}" for i, p in marks))
    return span


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind"); ap.add_argument("fmt"); ap.add_argument("sw", type=int); ap.add_argument("sh", type=int); ap.add_argument("dw", type=int); ap.add_argument("dh", type=int)
    ap.add_argument("--band", type=lambda x: int(x, 0), default=0); ap.add_argument("--mfma", type=lambda x: int(x, 0), default=0); ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--n", type=int, default=32); ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
    capi.set_tuning(capi.TUNE_RESIZE_BAND, a.band); capi.set_tuning(capi.TUNE_RESIZE_MFMA, a.mfma); capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, a.variant)
    sw, sh, dw, dh, N = a.sw, a.sh, a.dw, a.dh, a.n
    if a.kind == "fused":
        fmt_s, fmt_d, which = capi.NV12, capi.RGB, "fused"
    else:
        fmt_s = fmt_d = getattr(capi, a.fmt); which = "resize" if a.kind == "bilinear" else "lanczos"
    ring = max(N, min(256, int(600e6 // ((sw * sh + dw * dh) * 3)) // N * N))
    S = [surf(fmt_s, sw, sh, True) for _ in range(ring)]
    D = [surf(fmt_d, dw, dh, False) for _ in range(ring)]
    batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + N]]) for i in range(0, ring, N)]
    interp = 1 if a.kind == "bilinear" else 2

    def go(b):
        if a.kind == "fused":
            capi.convert_resize_batch(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, sw, sh, dw, dh, b)
        else:
            capi.resize_batch(ex, fmt_s, interp, sw, sh, dw, dh, b)
    for b in batches:
        go(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nd = 0
    for _ in range(a.reps):
        for b in batches:
            go(b); nd += 1
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / nd
    rec = read_records(which)
    label = f"{a.kind} {a.fmt} {sw}x{sh}->{dw}x{dh} band {a.band:#x} mfma {a.mfma:#x} variant {a.variant}, {N} frames per dispatch"
    span = analyse(rec, label)
    if span is not None:
        print(f"[wave_times]   event time {us:.2f} us per dispatch = {us / N:.3f} us/frame (instrumented build): {us - span:.2f} us per dispatch outside the first-wave -> last-wave span")


if __name__ == "__main__":
    main()
