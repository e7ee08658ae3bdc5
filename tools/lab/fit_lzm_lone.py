"""Fit of the launch planner's ONE-frame-per-dispatch band-height rule (vpf_lzm_plan.h, the block at the end of lzm_plan) to the n = 1 sweeps
profiles/r05_lanczos_shape_sweep_{down,up}_n1.txt (RGB / NV12 / YUV420 / Y x four down-scales + five up-scales, 20-24 launch shapes each).
A lone launch is one round of waves that start together; its time is the time of the fullest CU.  Model, per kernel class (ring of four /
ring of two x 8- / 4-tile strips):   cost = rounds x (a0 + a1 (L - 1) + work x max(tl, ti L)),   L = min(cap, ceil(workgroups / 256)),
with `work` the planner's own tile units.  Least squares on the relative error of every measured shape with bands of up to 16 tiles; then the
REGRET of three planners is printed: the current cost model, the occupancy model choosing strip width AND band height (worse on down-scales:
its cross-class calibration is poor), and the hybrid — strip width by the current model, band height by this one.  The planner uses the hybrid
for the ring-of-two kernels (up-scales) only: on the down-scales the 4-tile class fits badly (15 % mean error), its band heights TIE once the
constants are rounded (r x L is constant where the issue term rules), and a refit with free channel factors gains on 4K -> 1440p what it loses
on 1080p -> 900p — no signal.  The down-scales got a model of their own afterwards (second half of this file): with a ring-fill term, both strip
widths as candidates, fitted to TWO boxes' sweeps and cross-validated between them.  CPU only."""
import collections, math, os, sys
import numpy as np
from scipy.optimize import least_squares
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fit_lzm_up2 as U
import fit_lzm_down as D

W4 = {1: 0.45, 2: 0.9, 3: 0.8}


def feats(planes, n, nt, r):
    """-> (class, resident workgroup slots, workgroups, work of the longest wave) as lzm_plan computes them, or None where the strips do not fit"""
    if not all(U.PB.pb_lzm_span(ch, sw, dw, nt) for ch, sw, sh, dw, dh in planes):
        return None
    up2 = U.is_up2(planes, nt)
    if up2 == 2:
        vol = sum(((p[3] * p[0] + 127) // 128 + 3) // 4 * ((p[4] + 15) // 16) for p in planes) * n / 512.0
        if vol < 150:
            up2 = 0
    slots = (512 if up2 == 2 else 768 if nt == 8 else 1024) if up2 else (512 if nt == 8 else 768)
    wgs, work = 0, 0.0
    for ch, sw, sh, dw, dh in planes:
        tiles = (dh + 15) // 16
        gxp = ((dw * ch + 16 * nt - 1) // (16 * nt) + 3) // 4
        wgs += gxp * ((tiles + r - 1) // r) * n
        scy = sh / dh
        w = {3: 0.45, 2: 0.6, 1: 0.7}[ch] if up2 == 2 else (0.9 if nt == 8 else {3: 0.7, 2: 0.5, 1: 0.75}[ch]) if up2 else (1.0 if nt == 8 else W4[ch])
        vert = (0.5 + 0.5 * scy / 1.5) if nt == 8 else (0.3 + 0.7 * scy / 1.5)
        work = max(work, min(r, tiles) * w * vert)
    return ("u" if up2 else "d") + str(nt), slots, wgs, work


def lone_cost(k, slots, wgs, work):
    L = min(slots // 256, math.ceil(wgs / 256))
    return math.ceil(wgs / slots) * (k[0] + k[1] * (L - 1) + work * max(k[2], k[3] * L))


def fit(cases_):
    by = collections.defaultdict(list)
    for fmt, sw, sh, dw, dh, n, res in cases_:
        for (nt, r), t in res.items():
            f = feats(U.planes_of(fmt, sw, sh, dw, dh), n, nt, r)
            if f and r <= 16:
                by[f[0]].append((t,) + f[1:])
    K = {}
    for cls, rows in sorted(by.items()):
        y = np.array([r[0] for r in rows])
        sol = least_squares(lambda x: np.array([lone_cost(x, *r[1:]) for r in rows]) / y - 1, [5, 1, 1.5, 1.0], bounds=([0, 0, 0.1, 0.1], [20, 10, 10, 10]))
        e = np.abs(np.array([lone_cost(sol.x, *r[1:]) for r in rows]) / y - 1)
        K[cls] = sol.x
        print(f"class {cls}: {len(rows)} shapes, a0 a1 tl ti = {np.round(sol.x, 2)}, relative error mean {e.mean():.3f} max {e.max():.3f}")
    return K


def pick(planes, n, K, how):
    def best_of(nts, cost):
        best = None
        tmax = max((p[4] + 15) // 16 for p in planes)
        for nt in nts:
            for r in range(min(2, tmax), min(tmax, 64) + 1):
                f = feats(planes, n, nt, r)
                if not f:
                    break
                c = cost(f)
                if best is None or c < best[0]:
                    best = (c, nt, r)
        return best[1], best[2]
    old = lambda f: (2.0 + f[3]) * math.ceil(f[2] / f[1])
    new = lambda f: lone_cost(K[f[0]], *f[1:])
    if how == "current":
        return best_of((8, 4), old)
    if how == "occupancy":
        return best_of((8, 4), new)
    nt, _ = best_of((8, 4), old)
    return best_of((nt,), new)


def regret(cases_, K, how, verbose=False):
    out = []
    for fmt, sw, sh, dw, dh, n, res in cases_:
        nt, r = pick(U.planes_of(fmt, sw, sh, dw, dh), n, K, how)
        rs = sorted(rr for (t, rr) in res if t == nt)
        lo = max([x for x in rs if x <= r], default=rs[0]); hi = min([x for x in rs if x >= r], default=rs[-1])
        t = res[(nt, lo)] if lo == hi else float(np.interp(r, [lo, hi], [res[(nt, lo)], res[(nt, hi)]]))
        out.append(t / min(res.values()) - 1.0)
        if verbose:
            b = min(res, key=res.get)
            print(f"  {fmt:6s} {sw}x{sh}->{dw}x{dh}: pick nt{nt} r{r} {t:.2f}  best nt{b[0]} r{b[1]} {res[b]:.2f}  regret {out[-1]:.3f}")
    return round(float(np.mean(out)), 4), round(float(np.max(out)), 4)


# ---- the down-scales (ring of four): a band's ring fill counts, both strip widths are candidates, two boxes' sweeps cross-validate each other
KD = {8: [4.26, 0.81, 0.91, 0.92, 0.49], 4: [3.32, 0.77, 1.05, 1.0, 0.15]}   # a0 ti w1 w2 cs as vpf_lzm_plan.h holds them (w3 = 1, tl = 1)
GATE = 384                                                                  # the rule applies where the current pick has more workgroups than that


def down_wave(planes, nt, r, k):
    wgs, work = 0, 0.0
    for ch, sw, sh, dw, dh in planes:
        tiles = (dh + 15) // 16
        wgs += ((dw * ch + 16 * nt - 1) // (16 * nt) + 3) // 4 * ((tiles + r - 1) // r)
        rr = min(r, tiles)
        work = max(work, {1: k[2], 2: k[3], 3: 1.0}[ch] * (k[4] * (rr * sh / dh + 2.0) + rr))
    return wgs, work


def down_cost(k, planes, nt, r):
    slots = 512 if nt == 8 else 768
    wgs, work = down_wave(planes, nt, r, k)
    return math.ceil(wgs / slots) * (k[0] + work * max(1.0, k[1] * min(slots // 256, math.ceil(wgs / 256))))


def visit_p_cases():
    """profiles/r06_p_lone_downscales.txt -> the sweep files' case tuples (Y 4K -> 1080p left out: one such plane goes to the tile kernel by policy)"""
    import re
    cur, out = None, collections.OrderedDict()
    for line in open(os.path.join(U.ROOT, "profiles", "r06_p_lone_downscales.txt")):
        m = re.match(r"\[lone\] \S+\s+(\w+)\s+(\d+)x(\d+)->(\d+)x(\d+)", line)
        if m:
            cur = (m.group(1),) + tuple(int(v) for v in m.groups()[1:]); out[cur] = {}
            continue
        m = re.match(r"\[lone\]\s+(0x[0-9a-f]+): ([\d.]+) us", line)
        if m and int(m.group(1), 16):
            out[cur][(int(m.group(1), 16) >> 8, int(m.group(1), 16) & 255)] = float(m.group(2))
    return [(f, sw, sh, dw, dh, 1, res) for (f, sw, sh, dw, dh), res in out.items() if not (f == "Y" and dw == 1920)]


def down_fit(cases_):
    K = {}
    for nt in (8, 4):
        rows = [(U.planes_of(f, sw, sh, dw, dh), r, v) for f, sw, sh, dw, dh, n, res in cases_ for (t, r), v in res.items()
                if t == nt and r <= 8 and all(U.PB.pb_lzm_span(ch, s, d, nt) for ch, s, _, d, _ in U.planes_of(f, sw, sh, dw, dh))]
        y = np.array([v for _, _, v in rows])
        sols = [least_squares(lambda x: np.array([down_cost(x, pl, nt, r) for pl, r, _ in rows]) / y - 1, x0, bounds=([0, 0.05, 0.05, 0.05, 0.0], [30, 5, 3, 3, 5]))
                for x0 in ([4, 0.6, 0.5, 0.6, 0.5], [2, 0.8, 0.8, 0.8, 1.0], [6, 0.5, 0.4, 0.5, 0.2])]
        K[nt] = min(sols, key=lambda s: s.cost).x
    return K


def down_pick(planes, K, gate):
    nt0, r0 = pick(planes, 1, {}, "current")
    if down_wave(planes, nt0, r0, K[nt0])[0] <= gate:
        return nt0, r0
    tmax = max((p[4] + 15) // 16 for p in planes)
    return min((down_cost(K[nt], planes, nt, r), -nt, r) for nt in (8, 4) if all(U.PB.pb_lzm_span(ch, s, d, nt) for ch, s, _, d, _ in planes)
               for r in range(min(2, tmax), min(tmax, 64) + 1))[1:]


def down_regret(cases_, K, gate):
    out = []
    for fmt, sw, sh, dw, dh, n, res in cases_:
        nt, r = down_pick(U.planes_of(fmt, sw, sh, dw, dh), K, gate)
        nt = abs(nt)
        rs = sorted(rr for (t, rr) in res if t == nt)
        lo = max([x for x in rs if x <= r], default=rs[0]); hi = min([x for x in rs if x >= r], default=rs[-1])
        t = res[(nt, lo)] if lo == hi else float(np.interp(r, [lo, hi], [res[(nt, lo)], res[(nt, hi)]]))
        out.append(t / min(res.values()) - 1.0)
    return round(float(np.mean(out)), 4), round(float(np.max(out)), 4)


if __name__ == "__main__":
    a, b = list(D.cases("r05_lanczos_shape_sweep_down", (1,))), visit_p_cases()
    print("down-scales, current planner: r05 sweep", regret(a, {}, "current"), " visit p", down_regret(b, KD, 1 << 30))
    for name, train in (("the r05 sweep", a), ("visit p", b), ("both", a + b)):
        K = down_fit(train)
        print(f"  fitted on {name}: 8-tile {np.round(K[8], 2)} 4-tile {np.round(K[4], 2)} -> regret r05 {down_regret(a, K, GATE)}  visit p {down_regret(b, K, GATE)}   (no gate: {down_regret(a, K, 0)})")
    print("  the planner's constants:", KD, "-> regret r05", down_regret(a, KD, GATE), " visit p", down_regret(b, KD, GATE))
    down = list(D.cases("r05_lanczos_shape_sweep_down", (1,)))
    up = list(D.cases("r05_lanczos_shape_sweep_up", (1,)))
    K = fit(down + up)
    for how in ("current", "occupancy", "hybrid"):
        print(f"{how:10s}: down-scales (mean, worst) {regret(down, K, how)}   up-scales {regret(up, K, how)}")
    print("the hybrid, case by case:")
    regret(down, K, "hybrid", True); regret(up, K, "hybrid", True)
