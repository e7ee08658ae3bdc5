"""Fused NV12 -> bilinear -> RGB (vpf_convert_resize_batch) at several scale factors, NEXT TO the pair it fuses measured in the same run
(vpf_convert_batch NV12 -> RGB into an intermediate ring, then vpf_resize_batch bilinear): the fused entry must never lose to the pair
(VERDICT r4, item 3).  32 frames per dispatch (FUSED_N to change), rings of frames past the 256 MiB Infinity Cache, timed by bench.sustained
(round 6: 300 ms pre-heat of the same calls, median of five >= 60 ms blocks, shader clock beside every number) — the method of
tools/resize_batch_bench.py.

Fractions are ALGORITHMIC bytes per time against 8 TB/s, counted like bench.py counts them: the source rows the taps really touch (a 3x
down-scale reads one luma row in three: counting the whole source gave "1.30 of 8 TB/s" until round 4 — a fraction above 1 means the byte
model is wrong) + the destination.  The unfused pair is charged the same algorithmic bytes (what the JOB needs), not its own traffic.

FUSED_VARIANTS="0,47,48": vpf_set_tuning(VPF_TUNE_NV12_RGB_VARIANT) values to run the fused entry under (0 policy, 47 the per-wave strips
of rounds 2-4, 48 workgroup strips beyond 2x); FUSED_PAIR=0 skips the unfused pair."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from videoprocessingframework_amd import capi
from resize_batch_bench import timed, clk, PROTOCOL

dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
N = int(os.environ.get("FUSED_N", "32"))
VARIANTS = [int(v) for v in os.environ.get("FUSED_VARIANTS", "0").split(",")]
PAIR = os.environ.get("FUSED_PAIR", "1") != "0"


def rows_touched(S, D):
    """source rows a D-row bilinear resize of S rows reads, in the kernels' fp32 arithmetic (bench.py: _rows_touched)"""
    d = np.arange(D, dtype=np.float32)
    s_ = np.clip((d + np.float32(0.5)) * np.float32(np.float32(S) / np.float32(D)) - np.float32(0.5), 0, S - 1).astype(np.float32)
    i0 = s_.astype(np.int64)
    i1 = np.minimum(i0 + 1, S - 1)
    return set(i0.tolist()) | set(i1[(s_ - i0.astype(np.float32)) != 0].tolist())


def main():
    for (sw, sh, dw, dh) in ((3840, 2160, 1280, 720), (3840, 2160, 1920, 1080), (3840, 2160, 1600, 900), (1920, 1080, 1280, 720),
                             (1920, 1080, 640, 360), (1920, 1080, 224, 224), (1920, 1080, 3840, 2160), (1280, 720, 1920, 1080)):
        sp, dp, mp = (sw + 255) // 256 * 256, (3 * dw + 255) // 256 * 256, (3 * sw + 255) // 256 * 256
        luma = rows_touched(sh, dh)
        nbytes = len(luma) * sw + len({r >> 1 for r in luma}) * sw + 3 * dw * dh   # touched luma rows + their chroma rows + the destination
        whole = sw * sh * 3 // 2 + 3 * dw * dh
        ring = max(N, min(256, int(600e6 // whole) // N * N))
        src = [torch.randint(0, 256, (sh * 3 // 2, sp), dtype=torch.uint8, device=dev) for _ in range(ring)]
        dst = [torch.zeros((dh, dp), dtype=torch.uint8, device=dev) for _ in range(ring)]
        io = [([(s.data_ptr(), sp), (s.data_ptr() + sh * sp, sp)], [(d.data_ptr(), dp)]) for s, d in zip(src, dst)]
        batches = [capi.make_batch(io[i:i + N]) for i in range(0, ring, N)]
        line = f"[fused] {sw}x{sh} -> {dw}x{dh} (x{sw / dw:.3g}, {nbytes / whole:.2f} of the source+destination bytes touched):"
        best = None
        for v in VARIANTS:
            capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, v)
            us = timed(lambda: [capi.convert_resize_batch(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, sw, sh, dw, dh, b) for b in batches], 5) / ring
            capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0)
            best = us if best is None or (v == 0) else best
            line += f"  fused[v{v}] {us:6.2f} us/frame = {nbytes / us / 8e6:.2f} of 8 TB/s [{clk()}] |"
        if PAIR:
            mid = [torch.empty((sh, mp), dtype=torch.uint8, device=dev) for _ in range(ring)]  # the 3 B/px intermediate the fused entry exists to avoid
            io1 = [([(s.data_ptr(), sp), (s.data_ptr() + sh * sp, sp)], [(m.data_ptr(), mp)]) for s, m in zip(src, mid)]
            io2 = [([(m.data_ptr(), mp)], [(d.data_ptr(), dp)]) for m, d in zip(mid, dst)]
            b1 = [capi.make_batch(io1[i:i + N]) for i in range(0, ring, N)]
            b2 = [capi.make_batch(io2[i:i + N]) for i in range(0, ring, N)]

            def pair():
                for x, y in zip(b1, b2):
                    capi.convert_batch(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, sw, sh, x)
                    capi.resize_batch(ex, capi.RGB, capi.INTERP_LINEAR, sw, sh, dw, dh, y)
            up = timed(pair, 5) / ring
            cpair = clk()
            uc = timed(lambda: [capi.convert_batch(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, sw, sh, x) for x in b1], 5) / ring
            line += f"  convert then resize {up:6.2f} us/frame (convert alone {uc:.2f}) = {nbytes / up / 8e6:.2f} [{cpair}] | fused / pair = {best / up:.2f}" + ("  LOSES" if best > up else "")
            del mid, b1, b2
        print(line + f"  ({N} frames per dispatch, ring {ring}, {PROTOCOL} protocol)", flush=True)
        del src, dst, batches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
