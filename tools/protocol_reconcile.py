"""`bench.py --extra` said Lanczos RGB 1080p -> 720p batched = 2.42 us per frame, `tools/resize_batch_bench.py` said 2.07 — same ring
(64), same random data, same entry point, two rounds running.  This runs BOTH tools' set-ups in ONE process on ONE box, each under
  (a) its round-5 timing   bench.py: 2 warm-up steps + 10 timed steps, median of 3 blocks, right after the ~10 idle seconds of the cpu_baseline leg
                           tool:     1 warm call + 5 timed calls, median of 3 passes, in the middle of a table of other shapes
  (b) bench.sustained      >= 300 ms pre-heat of the SAME call, median of five >= 60 ms blocks
with the shader clock (pp_dpm_sclk) printed for every number, for Lanczos-3 and bilinear.  Done = (b) of the two set-ups within 3 %.
python tools/protocol_reconcile.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from videoprocessingframework_amd import capi  # noqa: E402
import resize_batch_bench as rbb  # noqa: E402

dev = torch.device("cuda", 0)
PCI = bench.device_pci(0)
sclk = lambda: bench.sharding.current_sclk_mhz(PCI)  # noqa: E731


def main():
    sw, sh, dw, dh, ring = 1920, 1080, 1280, 720, 64
    nbytes = 3 * (sw * sh + dw * dh)
    for interp, name in ((2, "lanczos"), (1, "bilinear")):
        res = {}
        # ---- bench.py's set-up (its Workload class: one pitched random ring from a seeded generator, torch.empty destinations)
        wl = bench.Workload(f"rgb_resize_1080p_720p_{name}", dev, ring, 0, "batch")
        for idle in (10.0, 0.0):
            torch.cuda.synchronize(); time.sleep(idle)  # the cpu_baseline leg leaves the GPU idle for ~10 s before --extra's short blocks
            c0 = sclk()
            ev = sorted(bench.timed(wl, 10, 2, False)[1] for _ in range(3))[1]
            res[f"bench.py set-up, (a) r5 timing after {idle:.0f} s idle"] = (ev / 10 / ring * 1e6, (c0, sclk()))
        m = bench.sustained(wl.step, pci=PCI)
        res["bench.py set-up, (b) sustained"] = (m["us"] / ring, m["sclk_mhz"])
        del wl
        torch.cuda.empty_cache()
        # ---- the tool's set-up (its surf(): torch.randint sources, torch.zeros destinations, one capi.make_batch over the ring)
        S = [rbb.surf(capi.RGB, sw, sh, True) for _ in range(ring)]
        D = [rbb.surf(capi.RGB, dw, dh, False) for _ in range(ring)]
        batch = capi.make_batch([(s[1], d[1]) for s, d in zip(S, D)])
        fn = lambda: capi.resize_batch(rbb.ex, capi.RGB, interp, sw, sh, dw, dh, batch)  # noqa: E731
        c0 = sclk()
        res["tool set-up, (a) r5 timing (1 warm + 3 x 5 calls)"] = (rbb.timed_burst(fn, 5) / ring, (c0, sclk()))
        m = bench.sustained(fn, pci=PCI)
        res["tool set-up, (b) sustained"] = (m["us"] / ring, m["sclk_mhz"])
        c0 = sclk()
        res["tool set-up, (a) again, right after (b)"] = (rbb.timed_burst(fn, 5) / ring, (c0, sclk()))
        del S, D, batch
        torch.cuda.empty_cache()
        for k, (us, c) in res.items():
            print(f"[reconcile] {name:8s} RGB {sw}x{sh}->{dw}x{dh} ring {ring}: {k:58s} {us:6.3f} us/frame = {nbytes / us / 8e6:.3f} of 8 TB/s   sclk {c[0]}/{c[1]} MHz", flush=True)
        b1, b2 = res["bench.py set-up, (b) sustained"][0], res["tool set-up, (b) sustained"][0]
        print(f"[reconcile] {name:8s} (b) vs (b): {b1:.3f} vs {b2:.3f} us/frame, {abs(b1 - b2) / min(b1, b2) * 100:.1f} % apart", flush=True)


if __name__ == "__main__":
    main()
