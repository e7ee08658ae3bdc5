"""lone_lanczos.py LIB [--quick | --up | --down]: Lanczos-3, ONE dispatch per frame (vpf_resize in a loop over a ring past the Infinity Cache), us per frame under
bench.sustained for a list of VPF_TUNE_RESIZE_MFMA knob values, with the kernel library LIB (capi.LIB_PATH swapped before the first call) — the
lone matrix-core launch of VERDICT r4 item 6 / r5 item 5 measured shape by shape, and, with tools/lab/ab/libvpfhip_r05_shared_columns.so (the round-5
tree + tools/lab/lanczos_shared_columns.patch: knob | 0x100000), the shared-column form at one frame per dispatch, which round 5 only measured
batched.  A knob the library refuses (set_tuning -> -1) is skipped."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
LIBP = os.path.abspath(sys.argv[1])
capi.LIB_PATH = LIBP
QUICK = "--quick" in sys.argv
DOWN = "--down" in sys.argv   # the 4K down-scales (ring of four): policy against both strip widths at bands of 2 .. 6 tiles — the planner's n = 1 regret
UP = "--up" in sys.argv   # the up-scales (ring of two): policy against forced 4-tile strips with bands of 3 .. 9 tiles — the band heights the n = 1 rule of the planner chooses among
sys.argv = sys.argv[:1]
from resize_batch_bench import surf, timed, clk  # noqa: E402

ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
SC = 0x100000
SHAPES = (
    ("RGB", 3840, 2160, 1920, 1080, [0, (8 << 8) | 2, (8 << 8) | 3, (4 << 8) | 2, (4 << 8) | 3, (4 << 8) | 4,
                                     SC | (4 << 8) | 1, SC | (4 << 8) | 2, SC | (4 << 8) | 3, SC | (4 << 8) | 4, SC | (4 << 8) | 6]),
    ("NV12", 3840, 2160, 1920, 1080, [0, (8 << 8) | 2, (4 << 8) | 2, SC | (8 << 8) | 1, SC | (8 << 8) | 2, SC | (4 << 8) | 2, SC | (4 << 8) | 3, SC | (4 << 8) | 4]),
    ("RGB", 1920, 1080, 1280, 720, [0, 0x40000, 0x40000 | (8 << 8) | 2, 0x40000 | (4 << 8) | 2, 0x40000 | SC | (8 << 8) | 1, 0x40000 | SC | (8 << 8) | 2,
                                    0x40000 | SC | (4 << 8) | 1, 0x40000 | SC | (4 << 8) | 2, 0x40000 | SC | (4 << 8) | 3]),
    ("RGB", 3840, 2160, 2560, 1440, [0, (8 << 8) | 2, SC | (8 << 8) | 1, SC | (8 << 8) | 2, SC | (4 << 8) | 2, SC | (4 << 8) | 3]),
)
DOWNK = [0] + [(8 << 8) | r for r in (2, 3, 4, 6)] + [(4 << 8) | r for r in (2, 3, 4, 5, 6)]
DOWN_SHAPES = tuple((f, 3840, 2160, dw, dh, DOWNK) for f in ("RGB", "NV12", "YUV420", "Y") for (dw, dh) in ((1920, 1080), (2560, 1440)))
UPK = [0] + [(4 << 8) | r for r in (3, 4, 5, 6, 7, 8, 9)]
UP_SHAPES = tuple((f, sw, sh, 3840, 2160, UPK) for f in ("RGB", "NV12", "YUV420") for (sw, sh) in ((1920, 1080), (2560, 1440))) + (("Y", 1920, 1080, 3840, 2160, UPK),)
for fname, sw, sh, dw, dh, knobs in UP_SHAPES if UP else DOWN_SHAPES if DOWN else SHAPES[:2] if QUICK else SHAPES:
    fmt = getattr(capi, fname)
    ring = max(32, min(256, int(600e6 // (sw * sh * 3 + dw * dh * 3)) // 32 * 32))
    S = [surf(fmt, sw, sh, True) for _ in range(ring)]
    D = [surf(fmt, dw, dh, False) for _ in range(ring)]
    planes = [(capi.planes(s[1]), capi.planes(d[1])) for s, d in zip(S, D)]
    nbytes = S[0][2] + D[0][2]
    cells = []
    for k in knobs:
        if capi.set_tuning(capi.TUNE_RESIZE_MFMA, k) == -1:
            cells.append(f"{k:#x}: refused")
            continue
        t = timed(lambda: [capi.resize(ex, fmt, 2, sw, sh, s, dw, dh, d) for s, d in planes], 3) / ring
        cells.append(f"{k:#x}: {t:.2f} us ({nbytes / t / 8e6:.2f}) [{clk()}]")
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
    print(f"[lone] {os.path.basename(LIBP):36s} {fname:5s} {sw}x{sh}->{dw}x{dh} lanczos3, one dispatch per frame, ring {ring}:\n[lone]     " + "\n[lone]     ".join(cells), flush=True)
    del S, D, planes
    torch.cuda.empty_cache()
