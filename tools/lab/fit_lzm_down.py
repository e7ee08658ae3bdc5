"""Refit of the launch planner's <= 32-frame constants for the ring-of-four kernels (vpf_lzm_plan.h) to the round-5 sweeps of the down-scales
(profiles/r05_lanczos_shape_sweep_down_n*.txt: RGB / NV12 / YUV420 / Y x four size pairs x 32 / 8 / 1 frames per dispatch), with the
ring-of-two constants held: a python restatement of the cost model, grid search for the lowest mean regret.  CPU only."""
import itertools, math, os, re, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fit_lzm_up2 as U
ROOT = U.ROOT


def cases(prefix, ns):
    for n in ns:
        for line in open(os.path.join(ROOT, "profiles", f"{prefix}_n{n}.txt")):
            m = re.match(r"\[lzm-sweep\] (\w+)\s+(\d+)x(\d+)->(\d+)x(\d+) n=(\d+):", line)
            fmt, (sw, sh, dw, dh, nn) = m.group(1), (int(v) for v in m.groups()[1:])
            res = {(int(a), int(b)): float(c) for a, b, c in re.findall(r"nt(\d) r(\d+)=([\d.]+)", line)}
            if nn > 32 and 3 * (sw * sh + dw * dh) > 7_000_000 and fmt == "RGB":
                continue
            yield fmt, sw, sh, dw, dh, nn, res


UP = dict(slots8=768, slots4=1024, w8=0.8, w4={1: 0.65, 2: 0.5, 3: 0.7})


def pick(planes, n, P):
    best = None
    for nt in (8, 4):
        if not all(U.PB.pb_lzm_span(ch, sw, dw, nt) for ch, sw, sh, dw, dh in planes):
            continue
        up2 = U.is_up2(planes, nt)
        slots = (UP["slots8"] if nt == 8 else UP["slots4"]) if up2 else (512 if nt == 8 else 768)
        tmax = max((p[4] + 15) // 16 for p in planes)
        for r in range(min(2, tmax), min(tmax, 64) + 1):
            wgs, work = 0, 0.0
            for ch, sw, sh, dw, dh in planes:
                tiles = (dh + 15) // 16
                gxp = ((dw * ch + 16 * nt - 1) // (16 * nt) + 3) // 4
                wgs += gxp * ((tiles + r - 1) // r) * n
                scy = sh / dh
                w = (UP["w8"] if nt == 8 else UP["w4"][ch]) if up2 else (1.0 if nt == 8 else P["w4"][ch])
                vert = (P["a8"] + (1 - P["a8"]) * scy / 1.5) if nt == 8 else (P["a4"] + (1 - P["a4"]) * scy / 1.5)
                work = max(work, min(r, tiles) * w * vert)
            cost = (P["S"] + work) * math.ceil(wgs / slots)
            if best is None or cost < best[0]:
                best = (cost, nt, r)
    return best[1], best[2]


def regret(P, cs, verbose=False):
    out = []
    for fmt, sw, sh, dw, dh, n, res in cs:
        nt, r = pick(U.planes_of(fmt, sw, sh, dw, dh), n, P)
        rs = sorted(rr for (t, rr) in res if t == nt)
        lo = max([x for x in rs if x <= r], default=rs[0]); hi = min([x for x in rs if x >= r], default=rs[-1])
        t = res[(nt, lo)] if lo == hi else float(np.interp(r, [lo, hi], [res[(nt, lo)], res[(nt, hi)]]))
        out.append(t / min(res.values()) - 1.0)
        if verbose:
            b = min(res, key=res.get)
            print(f"  {fmt:6s} {sw}x{sh}->{dw}x{dh} n={n}: pick nt{nt} r{r} {t:.2f}  best nt{b[0]} r{b[1]} {res[b]:.2f}  regret {out[-1]:.3f}")
    return float(np.mean(out)), float(np.max(out))


if __name__ == "__main__":
    down = list(cases("r05_lanczos_shape_sweep_down", (32, 8, 1)))
    up = list(cases("r05_lanczos_shape_sweep_up", (32, 8, 1)))
    cur = dict(S=2.0, w4={1: 0.45, 2: 0.9, 3: 0.8}, a8=0.5, a4=0.3)
    print("current constants: down", regret(cur, down), " up", regret(cur, up))
    best = None
    for S, a, b, c, a8, a4 in itertools.product((1.0, 1.5, 2.0, 3.0), (0.45, 0.55, 0.65, 0.75, 0.85), (0.7, 0.9, 1.1), (0.8, 0.9, 1.0, 1.1), (0.3, 0.5, 0.7), (0.1, 0.3, 0.5)):
        P = dict(S=S, w4={1: a, 2: b, 3: c}, a8=a8, a4=a4)
        m, w = regret(P, down)
        if best is None or m < best[0]:
            best = (m, w, P)
    print("refit:", best, " up with it", regret(best[2], up))
    regret(best[2], down, True)
    print("one constant at a time (the 1-channel 4-tile strip's work factor):")
    for a in (0.45, 0.5, 0.55, 0.6, 0.65, 0.7):
        P = dict(cur, w4={1: a, 2: 0.9, 3: 0.8})
        print(f"  w4[1] = {a}: down {regret(P, down)}  up {regret(P, up)}")
