"""Times every non-headline kernel family through the C ABI on a ring of device-resident 4K frames (per-frame dispatch,
and batched where the ABI offers it): algorithmic GB/s and fraction of the 8 TB/s HBM peak.  Evidence for DESIGN.md §4."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
W, H, RING, STEPS = int(os.environ.get("VPF_BENCH_W", "3840")), int(os.environ.get("VPF_BENCH_H", "2160")), 16, 5  # frame size: 4K unless VPF_BENCH_W / _H say otherwise
cw, ch = W // 2, H // 2


def planes(fmt, w=W, h=H):
    """allocate pitched planes for fmt; returns (keepalive, desc list, bytes)"""
    shapes = {capi.Y: [(h, w)], capi.NV12: [(h * 3 // 2, w)], capi.YUV420: [(h, w), (h // 2, w // 2), (h // 2, w // 2)],
              capi.YCBCR: [(h, w), (h // 2, w // 2), (h // 2, w // 2)], capi.RGB: [(h, 3 * w)], capi.BGR: [(h, 3 * w)],
              capi.RGB_PLANAR: [(3 * h, w)], capi.YUV444: [(3 * h, w)], capi.RGB_32F: [(h, 12 * w)], capi.RGB_32F_PLANAR: [(3 * h, 4 * w)],
              capi.P10: [(h * 3 // 2, 2 * w)]}[fmt]
    keep, desc, nbytes = [], [], 0
    for rows, rb in shapes:
        pitch = (rb + 255) // 256 * 256
        t = torch.randint(0, 256, (rows, pitch), dtype=torch.uint8, device=dev)
        keep.append(t)
        nbytes += rows * rb
        if fmt in (capi.NV12, capi.P10):
            desc += [(t.data_ptr(), pitch), (t.data_ptr() + h * pitch, pitch)]
        elif fmt in (capi.RGB_PLANAR, capi.YUV444, capi.RGB_32F_PLANAR):
            desc += [(t.data_ptr() + i * h * pitch, pitch) for i in range(3)]
        else:
            desc.append((t.data_ptr(), pitch))
    return keep, desc, nbytes


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(STEPS):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / STEPS


VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # 40 = legacy p4/p16 fast paths, 9 = generic (A/B runs)
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None
capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, VARIANT)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = {v: k for k, v in vars(capi).items() if k.isupper() and isinstance(v, int) and k in ("Y", "RGB", "NV12", "YUV420", "RGB_PLANAR", "BGR", "YCBCR", "YUV444", "RGB_32F", "RGB_32F_PLANAR", "P10")}
pairs = [(capi.NV12, capi.YUV420), (capi.YUV420, capi.NV12), (capi.RGB, capi.RGB_PLANAR), (capi.RGB_PLANAR, capi.RGB), (capi.RGB, capi.BGR),
         (capi.YUV420, capi.RGB), (capi.YUV444, capi.BGR), (capi.NV12, capi.RGB_PLANAR), (capi.RGB, capi.YUV420), (capi.RGB, capi.YUV444),
         (capi.BGR, capi.YCBCR), (capi.NV12, capi.Y), (capi.RGB, capi.Y), (capi.RGB, capi.RGB_32F), (capi.RGB_32F, capi.RGB_32F_PLANAR), (capi.P10, capi.NV12)]
print(f"{'op':28s} {'single GB/s':>12s} {'frac':>6s} {'batch GB/s':>12s} {'frac':>6s}")
for s, d in pairs:
    if ONLY and f"{N[s]}->{N[d]}" not in ONLY:
        continue
    ring = [(planes(s), planes(d))]
    nbytes = ring[0][0][2] + ring[0][1][2]
    RING = max(16, -(-int(1.0e9) // nbytes) // 16 * 16)  # >= 1 GB per pass: well past the 256 MiB Infinity Cache
    ring += [(planes(s), planes(d)) for _ in range(RING - 1)]
    cs = capi.BT_601 if s in (capi.RGB, capi.BGR, capi.RGB_PLANAR) else capi.BT_709
    cr = capi.JPEG if s == capi.YUV444 else capi.MPEG
    if s == capi.YUV444: cs = capi.BT_601
    single = timed(lambda: [capi.convert(ex, s, d, cs, cr, W, H, a[1], b[1]) for a, b in ring])
    batch = capi.make_batch([(a[1], b[1]) for a, b in ring])
    bt = timed(lambda: capi.convert_batch(ex, s, d, cs, cr, W, H, batch))
    print(f"{N[s] + '->' + N[d]:28s} {nbytes * RING / single / 1e9:12.0f} {nbytes * RING / single / 8e12:6.3f} {nbytes * RING / bt / 1e9:12.0f} {nbytes * RING / bt / 8e12:6.3f}", flush=True)
    del ring
    torch.cuda.empty_cache()
# remap: identity + 0.5 px shift and barrel distortion, 4K RGB
for kind in (("shift", "barrel") if (ONLY is None or "remap" in ONLY) else ()):
    xm, ym = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    if kind == "shift":
        xm = xm + 0.5
    else:
        cx, cy = (W - 1) / 2, (H - 1) / 2
        nx, ny = (xm - cx) / cx, (ym - cy) / cy
        k = 1 + 0.1 * (nx * nx + ny * ny)
        xm, ym = (nx * k * cx + cx).astype(np.float32), (ny * k * cy + cy).astype(np.float32)
    dx, dy = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    RING = 16
    ring = [(planes(capi.RGB), planes(capi.RGB)) for _ in range(RING)]
    t = timed(lambda: [capi.remap(ex, capi.RGB, W, H, a[1][0], dx.data_ptr(), 4 * W, dy.data_ptr(), 4 * W, W, H, b[1][0]) for a, b in ring])
    nb = W * H * (8 + 3 + 3)  # maps + unique source bytes + store
    print(f"{'remap RGB 4K ' + kind:28s} {nb * RING / t / 1e9:12.0f} {nb * RING / t / 8e12:6.3f}   ({t / RING * 1e6:.1f} us/frame)", flush=True)
    del ring
