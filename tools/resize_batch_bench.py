"""vpf_resize_batch / vpf_remap_batch vs one dispatch per frame (and per plane): us per frame and fraction of the 8 TB/s HBM roofline on
ALGORITHMIC bytes (whole source frame read once + destination written once; a down-scale that skips source rows reads less).
Rings are sized past the 256 MiB Infinity Cache.  Every number follows bench.sustained (round 6: 300 ms pre-heat of the same call, median of five >= 60 ms
blocks, the shader clock printed beside it); VPF_BENCH_PROTOCOL=burst = rounds 2-5's timing (median of VPF_BENCH_PASSES x 5 calls after one warm
call), kept to reconcile old tables.  python tools/resize_batch_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
NAMES = {0: "nearest", 1: "bilinear", 2: "lanczos3"}
if len(sys.argv) > 1:  # e.g. 43: the tiled kernel for bilinear down-scales too (A/B against the row-pair kernel)
    capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, int(sys.argv[1]))
BAND = int(os.environ.get("VPF_BENCH_BAND", "0"), 0)        # rows per wave of the row-pair kernels (0 = policy)
ONLY = os.environ.get("VPF_BENCH_ONLY", "")              # "bilinear": skip the Lanczos lines and the remap section
capi.set_tuning(capi.TUNE_RESIZE_BAND, BAND)
capi.set_tuning(capi.TUNE_RESIZE_MFMA, int(os.environ.get("VPF_BENCH_MFMA", "0"), 0))  # Lanczos matrix-core kernel: 0 policy, 1 never, nt << 8 | tiles per band


def surf(fmt, w, h, rand):
    """-> (keepalive tensors, plane descriptors)"""
    def alloc(rows, rb):
        p = (rb + 255) // 256 * 256
        t = torch.randint(0, 256, (rows, p), dtype=torch.uint8, device=dev) if rand else torch.zeros((rows, p), dtype=torch.uint8, device=dev)
        return t, p
    if fmt == capi.RGB:
        t, p = alloc(h, 3 * w)
        return [t], [(t.data_ptr(), p)], 3 * w * h
    if fmt == capi.Y:
        t, p = alloc(h, w)
        return [t], [(t.data_ptr(), p)], w * h
    if fmt == capi.NV12:
        t, p = alloc(h * 3 // 2, w)
        return [t], [(t.data_ptr(), p), (t.data_ptr() + h * p, p)], w * h * 3 // 2
    if fmt == capi.YUV420:
        a, pa = alloc(h, w); b, pb = alloc(h // 2, w // 2); c, pc = alloc(h // 2, w // 2)
        return [a, b, c], [(a.data_ptr(), pa), (b.data_ptr(), pb), (c.data_ptr(), pc)], w * h * 3 // 2
    raise ValueError(fmt)


PASSES = int(os.environ.get("VPF_BENCH_PASSES", "3"))  # burst protocol only: timed passes per number, the MEDIAN is reported
PROTOCOL = os.environ.get("VPF_BENCH_PROTOCOL", "sustained")  # "sustained" (round 6, every committed table): bench.sustained — 300 ms pre-heat of the SAME call, median of five
                                                              # >= 60 ms blocks, shader clock read beside it; "burst": rounds 2-5's few-call timing, kept for the reconciliation only
import bench  # noqa: E402  (repo root: the one protocol lives next to the headline benchmark)
PCI = bench.device_pci(0)
LAST = {}  # the last measurement's clocks, for the line being printed


def timed_burst(fn, reps, passes=None):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for _ in range(passes or PASSES):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(out)[len(out) // 2]


def timed(fn, reps, passes=None):
    """microseconds per fn() under the protocol in force (`reps` / `passes` only matter to the burst form)"""
    if PROTOCOL == "burst":
        LAST["sclk"] = (bench.sharding.current_sclk_mhz(PCI),) * 2
        return timed_burst(fn, reps, passes)
    m = bench.sustained(fn, pci=PCI)
    LAST["sclk"] = m["sclk_mhz"]
    return m["us"]


def clk():
    a, b = LAST.get("sclk", (None, None))
    return f"sclk {a}/{b} MHz"


def main():
    FORMATS = ((capi.RGB, "RGB"), (capi.NV12, "NV12"), (capi.YUV420, "YUV420")) + (((capi.Y, "Y"),) if os.environ.get("VPF_BENCH_Y") else ())
    for fmt, fname in FORMATS:
        for (sw, sh, dw, dh) in ((1920, 1080, 1280, 720), (1920, 1080, 416, 416), (3840, 2160, 1920, 1080), (1920, 1080, 3840, 2160), (1280, 720, 1920, 1080)):
            ring = max(32, min(256, int(600e6 // (sw * sh * 3 + dw * dh * 3)) // 32 * 32))
            S = [surf(fmt, sw, sh, True) for _ in range(ring)]
            D = [surf(fmt, dw, dh, False) for _ in range(ring)]
            NB = int(os.environ.get("VPF_BENCH_N", "0"))  # frames per batch (0: the whole ring, in dispatches of 32)
            if NB:
                batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + NB]]) for i in range(0, ring, NB)]
            batch = capi.make_batch([(s[1], d[1]) for s, d in zip(S, D)])
            planes = [(capi.planes(s[1]), capi.planes(d[1])) for s, d in zip(S, D)]
            nbytes = S[0][2] + D[0][2]
            for interp in ((1,) if ONLY == "bilinear" else (2,) if ONLY == "lanczos" else (1, 2)):
                if fmt not in (capi.RGB, capi.Y) and (sw, sh, dw, dh) not in ((1920, 1080, 1280, 720), (3840, 2160, 1920, 1080)):
                    continue
                if NB:
                    tb = timed(lambda: [capi.resize_batch(ex, fmt, interp, sw, sh, dw, dh, b) for b in batches], 5) / ring
                else:
                    tb = timed(lambda: capi.resize_batch(ex, fmt, interp, sw, sh, dw, dh, batch), 5) / ring
                ctb = clk()
                ts = timed(lambda: [capi.resize(ex, fmt, interp, sw, sh, s, dw, dh, d) for s, d in planes], 3) / ring
                extra = ""
                if os.environ.get("VPF_BENCH_ONE"):  # batches of ONE frame: the multi-plane / band kernels at single-frame launch sizes
                    ones = [capi.make_batch([(s[1], d[1])]) for s, d in zip(S, D)]
                    t1 = timed(lambda: [capi.resize_batch(ex, fmt, interp, sw, sh, dw, dh, b) for b in ones], 3) / ring
                    extra = f" | batches of one {t1:6.2f} us/frame ({nbytes / t1 / 8e6:.2f})"
                print(f"[resize_batch] {fname:6s} {sw}x{sh}->{dw}x{dh} {NAMES[interp]:8s}: batched {tb:6.2f} us/frame = {nbytes / tb / 1e6:5.2f} TB/s ({nbytes / tb / 8e6:.2f} of 8 TB/s)"
                      f" | one dispatch per frame {ts:6.2f} us/frame ({nbytes / ts / 8e6:.2f}){extra}  ring {ring}  [{PROTOCOL}: batched {ctb}, per frame {clk()}]", flush=True)
            del S, D, batch, planes
            torch.cuda.empty_cache()

    # remap: one pair of maps, many frames
    for (w, h) in (() if ONLY else ((1920, 1080), (3840, 2160))):
        ring = 32
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=dev), torch.arange(w, dtype=torch.float32, device=dev), indexing="ij")
        nx, ny = (xx - (w - 1) / 2) / ((w - 1) / 2), (yy - (h - 1) / 2) / ((h - 1) / 2)
        k = 1 + 0.1 * (nx * nx + ny * ny)
        xm, ym = (nx * k * ((w - 1) / 2) + (w - 1) / 2).contiguous(), (ny * k * ((h - 1) / 2) + (h - 1) / 2).contiguous()
        S = [surf(capi.RGB, w, h, True) for _ in range(ring)]
        D = [surf(capi.RGB, w, h, False) for _ in range(ring)]
        batch = capi.make_batch([(s[1], d[1]) for s, d in zip(S, D)])
        tb = timed(lambda: capi.remap_batch(ex, capi.RGB, w, h, xm.data_ptr(), 4 * w, ym.data_ptr(), 4 * w, w, h, batch), 5) / ring
        ctb = clk()
        ts = timed(lambda: [capi.remap(ex, capi.RGB, w, h, s[1][0], xm.data_ptr(), 4 * w, ym.data_ptr(), 4 * w, w, h, d[1][0]) for s, d in zip(S, D)], 3) / ring
        nb = 14 * w * h  # 8 B of maps + 3 B of source + 3 B written per pixel
        print(f"[remap_batch] RGB {w}x{h} barrel map: batched {tb:6.2f} us/frame = {nb / tb / 1e6:5.2f} TB/s ({nb / tb / 8e6:.2f} of 8 TB/s on 14 B/px)"
              f" | one dispatch per frame {ts:6.2f} us/frame ({nb / ts / 8e6:.2f})  [{PROTOCOL}: batched {ctb}, per frame {clk()}]", flush=True)
        del S, D, batch
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
