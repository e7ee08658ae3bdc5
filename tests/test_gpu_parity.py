"""-m gpu parity tests: HIP kernels (through the C ABI, libvpfhip.so) vs the CPU oracle.

Bar: BIT-EXACT against the oracle's FP32 mode (the restatement of the kernels' operation order) —
which tests/test_oracle_kat.py proves is within 1 LSB of exact round-half-up for every possible input —
and therefore within north_star's +-1 LSB of the specification-level oracle (asserted directly too).
Padding bytes of every pitched destination must come back untouched.
"""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():  # -m gpu on a box without a GPU: fail loudly rather than pass vacuously
    pytest.skip("no GPU visible", allow_module_level=True)

from gpu_util import DevPlanes, assert_planes_equal, stream_handle  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_FUZZ_SEEDS = range(int(os.environ.get("VPF_FUZZ_FIRST", "0")), int(os.environ.get("VPF_FUZZ_FIRST", "0")) + int(os.environ.get("VPF_FUZZ_SEEDS", "64")))
MATS = [(0, 0), (0, 1), (1, 0), (1, 1)]


def _convert(capi, oracle, sf, df, cs, cr, w, h, src, align=256, extra=0, offset=0, variant=0, exact_tol=True):
    s = DevPlanes(src, align, extra, offset)
    d = DevPlanes(oracle.alloc(df, w, h), align, extra, offset)
    prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
    try:
        capi.convert(capi.make_exec(stream_handle()), sf, df, cs, cr, w, h, s.desc(), d.desc())
    finally:
        capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
    torch.cuda.synchronize()
    got, intact = d.download()
    assert intact, "kernel wrote outside the destination rows (padding clobbered)"
    st, want = oracle.convert(sf, df, cs, cr, w, h, src, oracle.FP32)
    assert st == 0
    assert_planes_equal(got, want, f"convert {sf}->{df} cs{cs} cr{cr} {w}x{h} v{variant}")
    if exact_tol:
        st, ex = oracle.convert(sf, df, cs, cr, w, h, src, oracle.EXACT)
        for g, e in zip(got, ex):
            if g.dtype == np.uint8:
                assert np.abs(g.astype(np.int16) - e.astype(np.int16)).max() <= 1  # +-1 LSB (north_star)
    return got


# ---------------------------------------------------------------------------------------------
# headline path: NV12 -> RGB / BGR / RGB_PLANAR
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cs,cr", MATS)
@pytest.mark.parametrize("dst", ["RGB", "BGR", "RGB_PLANAR"])
def test_nv12_to_rgb_matrices(capi, oracle, cs, cr, dst):
    w, h = 256, 64
    for dist in "ABC":
        src = oracle.synth(oracle.NV12, w, h, 1000, dist)
        _convert(capi, oracle, capi.NV12, getattr(capi, dst), cs, cr, w, h, src)


@pytest.mark.parametrize("variant", [4, 8, 9, 12, 30, 37, 40, 44, 45, 46])
@pytest.mark.parametrize("dst", ["RGB", "BGR", "RGB_PLANAR"])
def test_nv12_to_rgb_every_kernel_variant(capi, oracle, variant, dst):
    """every NV12 -> RGB kernel libvpfhip contains (the only values vpf_set_tuning accepts: p4 / p16 with non-temporal or allocating
    stores / p16 capped at 4 workgroups per CU / p16x = blocks numbered straight through the picture / planar r16 / generic) agrees
    bit for bit; the experimental forms live in tools/lab.  Widths around the p16x conditions: 1024 (64 blocks per row: a wave = a
    row pair), 1040 (65: every wave crosses a row boundary at a different lane), 1008 (below 1024: falls back to p16)"""
    for (w, h) in [(1920, 32), (3840, 8), (848, 464), (1280, 18), (1024, 6), (1040, 10), (1008, 4), (4096, 2), (5008, 6)]:
        src = oracle.synth(oracle.NV12, w, h, 1001)
        _convert(capi, oracle, capi.NV12, getattr(capi, dst), 1, 0, w, h, src, variant=variant, exact_tol=False)


@pytest.mark.parametrize("variant", [8, 12, 30, 37, 44, 45, 46])
def test_nv12_to_rgb_variant_falls_back_when_not_applicable(capi, oracle, variant):
    """a 16-B-aligned-only kernel requested on ragged widths / odd bases / the other output class must silently take a
    general kernel with identical pixels (the tuning knob is a hint, never a correctness switch)"""
    for (w, h) in [(1002, 6), (66, 34), (17, 9), (1920, 8)]:
        src = oracle.synth(oracle.NV12, w, h, 1003)
        for dst in (capi.RGB, capi.RGB_PLANAR):
            _convert(capi, oracle, capi.NV12, dst, 1, 0, w, h, src, variant=variant, align=256, exact_tol=False)
            _convert(capi, oracle, capi.NV12, dst, 1, 0, w, h, src, variant=variant, align=64, extra=3, offset=1, exact_tol=False)


def test_nv12_frame_kat_on_gpu(capi, oracle):
    """the committed golden frame (tests/golden/nv12_frame_kat.json) through the HIP path: +-1 LSB, and
    bit-exact vs the oracle's FP32 mode"""
    kat = json.load(open(os.path.join(G, "nv12_frame_kat.json")))
    w, h = kat["w"], kat["h"]
    src = [np.array(kat["y"], np.uint8), np.array(kat["uv"], np.uint8)]
    for key, (cs, cr) in {"601_MPEG": (0, 0), "601_JPEG": (0, 1), "709_MPEG": (1, 0), "709_JPEG": (1, 1)}.items():
        got = _convert(capi, oracle, capi.NV12, capi.RGB, cs, cr, w, h, src)
        want = np.array(kat["rgb"][key], np.uint8)
        assert np.abs(got[0].astype(int) - want.astype(int)).max() <= 1


@pytest.mark.parametrize("w,h", [(2, 2), (1, 1), (3, 5), (4, 2), (6, 4), (10, 6), (17, 9), (64, 2), (66, 34), (255, 3),
                                 (1000, 10), (1022, 6), (1026, 4)])
def test_nv12_to_rgb_ragged_sizes(capi, oracle, w, h):
    """minimal, odd and ragged sizes (generic path) incl. width not multiple of 4/16/64"""
    src = oracle.synth(oracle.NV12, w, h, 1002)
    for dst in (capi.RGB, capi.BGR, capi.RGB_PLANAR):
        _convert(capi, oracle, capi.NV12, dst, 1, 0, w, h, src, align=1)       # pitch == row bytes
        _convert(capi, oracle, capi.NV12, dst, 0, 1, w, h, src, align=64, extra=3, offset=1)  # odd pitch, odd base
        for variant in (0, 4, 40):  # padded (4-B aligned) pitches: the ragged row end / odd last row inside the p4 kernels
            _convert(capi, oracle, capi.NV12, dst, 1, 1, w, h, src, align=256, variant=variant)
    ysrc = oracle.synth(oracle.YUV420, w, h, 1003)
    _convert(capi, oracle, capi.YUV420, capi.RGB, 0, 0, w, h, ysrc, align=256)
    _convert(capi, oracle, capi.YUV420, capi.RGB_PLANAR, 0, 1, w, h, ysrc, align=8)


@pytest.mark.parametrize("align,extra,offset", [(256, 0, 0), (4, 0, 0), (4, 4, 4), (16, 0, 16), (1, 0, 0), (256, 0, 2)])
def test_nv12_to_rgb_pitch_and_alignment(capi, oracle, align, extra, offset):
    w, h = 1280, 24
    src = oracle.synth(oracle.NV12, w, h, 1003)
    for variant in (0, 4, 8):
        _convert(capi, oracle, capi.NV12, capi.RGB, 1, 0, w, h, src, align, extra, offset, variant=variant)


def test_saturation_corners(capi, oracle):
    """every clamp: planes of constant extreme values"""
    w, h = 64, 8
    for yv in (0, 255):
        for uv in (0, 255):
            for vv in (0, 255):
                y = np.full((h, w), yv, np.uint8)
                c = np.empty((h // 2, w), np.uint8)
                c[:, 0::2], c[:, 1::2] = uv, vv
                for cs, cr in MATS:
                    _convert(capi, oracle, capi.NV12, capi.RGB, cs, cr, w, h, [y, c])


def test_exhaustive_yuv_triples_on_gpu(capi, oracle):
    """All 2^24 (Y,U,V) triples through the real kernel, per matrix: a 4096 x 8192 NV12 frame whose 2x2
    quads enumerate (U,V) and whose luma enumerates Y within 16 x 16 blocks of quads. Bit-exact vs FP32 oracle."""
    w, h = 8192, 4096  # chroma grid 4096 x 2048 = 2^23 quads x 4 luma = 2^25 px >= 2 * 2^24 (each triple >= once)
    qy, qx = np.meshgrid(np.arange(h // 2), np.arange(w // 2), indexing="ij")
    # quad index -> (u, v, ybase): u = qx & 255, v = qy & 255, block = (qx >> 8) + 16 * (qy >> 8) in [0, 128)
    u = (qx & 255).astype(np.uint8)
    v = (qy & 255).astype(np.uint8)
    blk = ((qx >> 8) + 16 * (qy >> 8)).astype(np.int32)  # 0..127
    uv = np.empty((h // 2, w), np.uint8)
    uv[:, 0::2], uv[:, 1::2] = u, v
    y = np.empty((h, w), np.uint8)
    # the 4 luma samples of quad (block b) take Y values 2b, 2b+1 (top row) and again 2b, 2b+1 (bottom) -> all 256 Y
    y[0::2, 0::2] = (2 * blk).astype(np.uint8)
    y[0::2, 1::2] = (2 * blk + 1).astype(np.uint8)
    y[1::2, 0::2] = (2 * blk + 1).astype(np.uint8)
    y[1::2, 1::2] = (2 * blk).astype(np.uint8)
    for cs, cr in MATS:
        _convert(capi, oracle, capi.NV12, capi.RGB, cs, cr, w, h, [y, uv], exact_tol=False)


def test_full_size_4k_and_1080p(capi, oracle):
    """BASELINE.json sizes, full frames, bit-exact (configs[1] 1080p NV12->RGB_PLANAR 709 limited; 4K NV12->RGB)"""
    src = oracle.synth(oracle.NV12, 1920, 1080, 1004, "B")
    _convert(capi, oracle, capi.NV12, capi.RGB_PLANAR, 1, 0, 1920, 1080, src)
    src = oracle.synth(oracle.NV12, 3840, 2160, 1005, "A")
    for variant in (0, 4):
        _convert(capi, oracle, capi.NV12, capi.RGB, 1, 0, 3840, 2160, src, variant=variant, exact_tol=False)


def test_batch_matches_single(capi, oracle):
    """vpf_convert_batch over 70 frames (3 dispatches of <=32) == 70 single conversions; outputs independent.  640 px: the chunk-per-row
    p16 kernel; 1040 px (65 blocks per row pair): p16x, every wave crossing a row boundary at a different lane"""
    for (w, h, n, dst) in ((640, 36, 70, "RGB"), (1040, 10, 37, "RGB"), (1040, 10, 37, "RGB_PLANAR"), (2064, 6, 5, "RGB_PLANAR")):
        srcs = [oracle.synth(oracle.NV12, w, h, 2000 + i) for i in range(n)]
        S = [DevPlanes(s) for s in srcs]
        D = [DevPlanes(oracle.alloc(getattr(oracle, dst), w, h)) for _ in range(n)]
        batch = capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)])
        capi.convert_batch(capi.make_exec(stream_handle()), capi.NV12, getattr(capi, dst), 1, 0, w, h, batch)
        torch.cuda.synchronize()
        for i in range(n):
            got, intact = D[i].download()
            assert intact
            _, want = oracle.convert(oracle.NV12, getattr(oracle, dst), 1, 0, w, h, srcs[i])
            assert_planes_equal(got, want, f"batch {dst} {w}x{h} frame {i}")


def test_single_frame_entry_equals_batch_entry(capi, oracle):
    """One frame per launch goes through the scalar-argument kernel entries (k_*_one), two or more through the by-value
    BatchArgs entries: the same bytes must come out of both, for every converter family that has the two entries and
    for the fused kernels (exact 3x, exact 2x, general ratio)."""
    ex = capi.make_exec(stream_handle())
    pairs = [("NV12", "RGB", 640, 36), ("NV12", "BGR", 1056, 8), ("NV12", "RGB_PLANAR", 640, 36), ("NV12", "RGB_PLANAR", 2048, 1536),
             ("NV12", "RGB", 1366, 10), ("NV12", "RGB", 1040, 10), ("YUV420", "BGR", 2064, 6), ("NV12", "RGB_PLANAR", 1040, 10), ("YUV420", "RGB", 640, 36), ("YUV420", "RGB_PLANAR", 640, 36), ("YUV444", "BGR", 640, 12),
             ("RGB", "RGB_PLANAR", 640, 12), ("RGB_PLANAR", "BGR", 640, 12), ("RGB", "BGR", 640, 12), ("RGB", "YUV420", 640, 12),
             ("BGR", "YUV444", 640, 12), ("NV12", "YUV420", 640, 12), ("YUV420", "NV12", 640, 12), ("RGB", "Y", 640, 12),
             ("RGB", "RGB_32F", 640, 12), ("P10", "NV12", 640, 12)]
    for sfmt, dfmt, w, h in pairs:
        sf, df = getattr(oracle, sfmt), getattr(oracle, dfmt)
        cs, cr = next((a, b) for a, b in MATS if capi.convert_supported(getattr(capi, sfmt), getattr(capi, dfmt), a, b))
        srcs = [oracle.synth(sf, w, h, 4100 + i) for i in range(2)]
        S = [DevPlanes(x) for x in srcs]
        D1 = [DevPlanes(oracle.alloc(df, w, h)) for _ in range(2)]
        D2 = [DevPlanes(oracle.alloc(df, w, h)) for _ in range(2)]
        for s, d in zip(S, D1):
            capi.convert(ex, getattr(capi, sfmt), getattr(capi, dfmt), cs, cr, w, h, s.desc(), d.desc())
        capi.convert_batch(ex, getattr(capi, sfmt), getattr(capi, dfmt), cs, cr, w, h, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D2)]))
        torch.cuda.synchronize()
        for i in range(2):
            (a, ia), (b, ib) = D1[i].download(), D2[i].download()
            assert ia and ib
            assert_planes_equal(a, b, f"single vs batch {sfmt}->{dfmt} {w}x{h} frame {i}")
    for (sw, sh, dw, dh) in [(1920, 48, 640, 16), (1920, 64, 960, 32), (640, 360, 213, 120)]:
        srcs = [oracle.synth(oracle.NV12, sw, sh, 4200 + i) for i in range(2)]
        S = [DevPlanes(x) for x in srcs]
        D1 = [DevPlanes(oracle.alloc(oracle.RGB, dw, dh)) for _ in range(2)]
        D2 = [DevPlanes(oracle.alloc(oracle.RGB, dw, dh)) for _ in range(2)]
        for s, d in zip(S, D1):
            capi.convert_resize(ex, capi.NV12, capi.RGB, 1, 0, sw, sh, s.desc(), dw, dh, d.desc())
        capi.convert_resize_batch(ex, capi.NV12, capi.RGB, 1, 0, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D2)]))
        torch.cuda.synchronize()
        for i in range(2):
            (a, ia), (b, ib) = D1[i].download(), D2[i].download()
            assert ia and ib
            assert_planes_equal(a, b, f"single vs batch fused {sw}x{sh}->{dw}x{dh} frame {i}")


def test_linearity_property_full_size(capi, oracle):
    """size-independent property at 4K: full-range luma ramp with neutral chroma maps to R=G=B=Y (JPEG matrices)"""
    w, h = 3840, 2160
    y = (np.arange(w, dtype=np.int64)[None, :] + np.arange(h, dtype=np.int64)[:, None]) % 256
    y = y.astype(np.uint8)
    uv = np.full((h // 2, w), 128, np.uint8)
    s, d = DevPlanes([y, uv]), DevPlanes(oracle.alloc(oracle.RGB_PLANAR, w, h))
    for cs in (0, 1):
        capi.convert(capi.make_exec(stream_handle()), capi.NV12, capi.RGB_PLANAR, cs, capi.JPEG, w, h, s.desc(), d.desc())
        torch.cuda.synchronize()
        got, intact = d.download()
        assert intact and all(np.array_equal(p, y) for p in got)


# ---------------------------------------------------------------------------------------------
# the rest of the converter matrix
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("src_fmt", ["YUV420", "YUV444"])
@pytest.mark.parametrize("dst", ["RGB", "BGR", "RGB_PLANAR"])
def test_planar_yuv_to_rgb(capi, oracle, src_fmt, dst):
    for (w, h) in [(640, 48), (3840, 4), (1040, 6), (30, 14), (7, 5)]:
        src = oracle.synth(getattr(oracle, src_fmt), w, h, 1010)
        for cs, cr in MATS:
            _convert(capi, oracle, getattr(capi, src_fmt), getattr(capi, dst), cs, cr, w, h, src)
        for variant in (40, 9):  # 40: p4 fast path on regular frames, 9: generic
            _convert(capi, oracle, getattr(capi, src_fmt), getattr(capi, dst), 1, 0, w, h, src, variant=variant)
        _convert(capi, oracle, getattr(capi, src_fmt), getattr(capi, dst), 0, 0, w, h, src, align=2, extra=2, offset=2)


RELAYOUT = [("NV12", "YUV420"), ("YUV420", "NV12"), ("RGB", "RGB_PLANAR"), ("RGB_PLANAR", "RGB"), ("RGB", "BGR"),
            ("BGR", "RGB"), ("BGR", "RGB_PLANAR"), ("RGB_PLANAR", "BGR"), ("NV12", "Y"), ("Y", "YUV444"),
            ("RGB", "Y"), ("BGR", "Y"), ("RGB_PLANAR", "Y"), ("P10", "NV12"), ("P12", "NV12")]


@pytest.mark.parametrize("s,d", RELAYOUT)
def test_relayout_converters(capi, oracle, s, d):
    for (w, h) in [(1920, 16), (3840, 6), (848, 464), (1040, 8), (2064, 4), (64, 4), (16, 2), (18, 10), (5, 3), (1, 1)]:
        src = oracle.synth(getattr(oracle, s), w, h, 1020)
        _convert(capi, oracle, getattr(capi, s), getattr(capi, d), 0, 1, w, h, src)
        _convert(capi, oracle, getattr(capi, s), getattr(capi, d), 0, 1, w, h, src, variant=40)  # p4 / p16 fast paths
        _convert(capi, oracle, getattr(capi, s), getattr(capi, d), 0, 1, w, h, src, align=16, extra=16, offset=16)
        _convert(capi, oracle, getattr(capi, s), getattr(capi, d), 0, 1, w, h, src, align=1)
        _convert(capi, oracle, getattr(capi, s), getattr(capi, d), 0, 1, w, h, src, variant=9)  # forced generic


def test_float_converters(capi, oracle):
    for (w, h) in [(320, 20), (3840, 4), (1040, 6), (7, 3)]:
        src = oracle.synth(oracle.RGB, w, h, 1021)
        got = _convert(capi, oracle, capi.RGB, capi.RGB_32F, 0, 0, w, h, src)
        for variant in (40, 9):
            _convert(capi, oracle, capi.RGB, capi.RGB_32F, 0, 0, w, h, src, variant=variant)
        assert got[0].dtype == np.float32 and got[0].max() <= 1.0
        for variant in (0, 9):
            _convert(capi, oracle, capi.RGB_32F, capi.RGB_32F_PLANAR, 0, 0, w, h, got, variant=variant)


@pytest.mark.parametrize("s", ["RGB", "BGR", "RGB_PLANAR"])
@pytest.mark.parametrize("d", ["YUV444", "YUV420", "YCBCR"])
def test_rgb_to_yuv(capi, oracle, s, d):
    for (w, h) in [(640, 48), (3840, 4), (1040, 6), (2064, 2), (30, 14), (7, 5), (1, 1)]:
        src = oracle.synth(getattr(oracle, s), w, h, 1030)
        for cr in (0, 1):
            _convert(capi, oracle, getattr(capi, s), getattr(capi, d), 0, cr, w, h, src)
        for variant in (40, 9):  # 40: p4 fast path on regular frames, 9: quad kernel
            _convert(capi, oracle, getattr(capi, s), getattr(capi, d), 0, 0, w, h, src, variant=variant)


def test_roundtrip_rgb_yuv420_nv12_rgb(capi, oracle):
    """the return path of samples/SamplePyTorch.py:150-158 (BT.601 MPEG): RGB -> YUV420 -> NV12 -> YUV420 -> RGB
    reproduces an in-gamut picture that is constant over 2x2 quads to within 2 LSB (two roundings; measured bound
    on the oracle).  (The JPEG / NPP "YUV" model is NOT round-trippable: V = .877(R-Y)+128 clips for saturated reds.)"""
    w, h = 256, 64
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (h // 2, w // 2, 3), dtype=np.uint8)
    rgb = np.repeat(np.repeat(q, 2, 0), 2, 1).reshape(h, 3 * w)
    a = _convert(capi, oracle, capi.RGB, capi.YUV420, 0, 0, w, h, [rgb])
    b = _convert(capi, oracle, capi.YUV420, capi.NV12, 0, 0, w, h, a)
    c = _convert(capi, oracle, capi.NV12, capi.YUV420, 0, 0, w, h, b)
    d = _convert(capi, oracle, capi.YUV420, capi.RGB, 0, 0, w, h, c)
    assert all(np.array_equal(x, y) for x, y in zip(a, c))
    assert np.abs(d[0].astype(int) - rgb.astype(int)).max() <= 2


# ---------------------------------------------------------------------------------------------
# resize / remap / fused
# ---------------------------------------------------------------------------------------------
def _resize(capi, oracle, fmt, interp, sw, sh, dw, dh, seed=1040, align=256):
    src = oracle.synth(fmt, sw, sh, seed)
    _, want = oracle.resize(fmt, interp, sw, sh, src, dw, dh, oracle.FP32)
    for variant in (0, 40, 9):  # 0: tiled (up-scale) or row-pair LDS kernel where they apply; 40: row-pair LDS; 9: gather
        s, d = DevPlanes(src, align), DevPlanes(oracle.alloc(fmt, dw, dh), align)
        prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
        try:
            capi.resize(capi.make_exec(stream_handle()), fmt, interp, sw, sh, s.desc(), dw, dh, d.desc())
        finally:
            capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
        torch.cuda.synchronize()
        got, intact = d.download()
        assert intact
        assert_planes_equal(got, want, f"resize fmt{fmt} {sw}x{sh}->{dw}x{dh} v{variant}")
    _, ex = oracle.resize(fmt, interp, sw, sh, src, dw, dh, oracle.EXACT)
    for g, e in zip(got, ex):
        assert np.abs(g.astype(int) - e.astype(int)).max() <= 1
    return got


@pytest.mark.parametrize("fmt", ["RGB", "BGR", "Y", "YUV420", "YUV444", "RGB_PLANAR", "NV12"])
def test_resize_bilinear(capi, oracle, fmt):
    f = getattr(capi, fmt)
    for (sw, sh, dw, dh) in [(3840, 64, 1280, 22), (640, 360, 224, 224), (100, 60, 333, 201), (64, 64, 64, 64), (9, 7, 2, 2),
                             (1000, 40, 300, 13), (2000, 16, 260, 5), (4096, 8, 258, 2), (300, 20, 1000, 70),
                             (320, 180, 1280, 720), (640, 40, 700, 333), (50, 30, 1921, 47),
                             (1920, 64, 960, 32), (72, 20, 36, 10), (1000, 8, 500, 4), (2056, 6, 1028, 3),  # exact 2x
                             (3840, 16, 1920, 8), (1056, 4, 528, 2), (32, 2, 16, 1), (2080, 6, 1040, 3)]:  # exact 2x, r16 form (w % 32 == 0): whole / ragged last chunk
        _resize(capi, oracle, f, capi.INTERP_LINEAR, sw, sh, dw, dh)
    _resize(capi, oracle, f, capi.INTERP_LINEAR, 128, 72, 50, 30, align=1)
    _resize(capi, oracle, f, capi.INTERP_NEAREST, 128, 72, 50, 30)


@pytest.mark.parametrize("fmt", ["RGB_32F", "RGB_32F_PLANAR"])
def test_resize_float_surfaces(capi, oracle, fmt):
    """reference R4 / R5 (Tasks.cpp:1334-1445): float surfaces through vpf_resize, every filter: bit-exact vs the oracle's
    fp32 restatement, within float rounding of the double-precision evaluation; misaligned float rows are refused"""
    f = getattr(capi, fmt)
    for (sw, sh, dw, dh) in [(640, 360, 224, 224), (100, 60, 333, 201), (64, 64, 64, 64), (9, 7, 20, 15), (1280, 30, 320, 9),
                             (960, 45, 320, 15), (300, 35, 100, 7)]:  # the last two: odd integer factors (centre-sample shortcut)
        src = oracle.synth(f, sw, sh, 1045)
        for interp in (capi.INTERP_NEAREST, capi.INTERP_LINEAR, capi.INTERP_LANCZOS3):
            _, want = oracle.resize(f, interp, sw, sh, src, dw, dh, oracle.FP32)
            for align, variant in ((256, 0), (4, 0), (256, 43), (256, 9)):  # tiled Lanczos / gather (unaligned) / tiled bilinear too / forced gather
                s, d = DevPlanes(src, align), DevPlanes(oracle.alloc(f, dw, dh), align)
                prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
                try:
                    capi.resize(capi.make_exec(stream_handle()), f, interp, sw, sh, s.desc(), dw, dh, d.desc())
                finally:
                    capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
                torch.cuda.synchronize()
                got, intact = d.download()
                assert intact
                assert_planes_equal(got, want, f"float resize {fmt} interp {interp} {sw}x{sh}->{dw}x{dh} a{align} v{variant}")
            _, ex = oracle.resize(f, interp, sw, sh, src, dw, dh, oracle.EXACT)
            for g, e in zip(got, ex):  # fp32 source coordinates carry ~1e-7 * x of error; samples are in [0, 1)
                if interp == capi.INTERP_NEAREST:  # a coordinate that rounds across a sample boundary picks a neighbour
                    assert (g != e).mean() < 0.01
                else:
                    assert np.abs(g - e).max() < 3e-4
    src = oracle.synth(f, 64, 16, 1)
    s, d = DevPlanes(src, 256, 0, 2), DevPlanes(oracle.alloc(f, 32, 8))
    assert capi.resize(capi.make_exec(stream_handle()), f, capi.INTERP_LINEAR, 64, 16, s.desc(), 32, 8, d.desc(), check=False) == capi.ERR_BAD_ARG


def test_resize_4k_to_720p_full(capi, oracle):
    """BASELINE.json configs[2]: 3840x2160 RGB -> 1280x720 bilinear, full frame"""
    _resize(capi, oracle, capi.RGB, capi.INTERP_LINEAR, 3840, 2160, 1280, 720)


def test_fused_convert_resize(capi, oracle):
    for (sw, sh, dw, dh) in [(3840, 2160, 1280, 720), (640, 360, 224, 224), (100, 60, 333, 201), (18, 10, 7, 5),
                             (1920, 64, 260, 9), (4096, 16, 258, 3), (322, 38, 1000, 111),
                             (1920, 64, 960, 32), (3840, 32, 1920, 16), (64, 36, 32, 18), (2000, 16, 1000, 8), (1936, 8, 968, 4),  # exact 2x
                             (1920, 48, 640, 16), (960, 30, 320, 10), (1280, 50, 256, 10)]:                                      # exact 3x / 5x
        for sfmt in ("NV12", "YUV420"):
            src = oracle.synth(getattr(oracle, sfmt), sw, sh, 1050)
            for dfmt in ("RGB", "BGR", "RGB_PLANAR"):
                _, want = oracle.convert_resize(getattr(oracle, sfmt), getattr(oracle, dfmt), 1, 0, sw, sh, src, dw, dh)
                for variant, align in ((0, 256), (40, 256), (9, 256), (0, 2), (47, 256), (48, 256)):  # fast paths, general LDS kernel, forced gather, unaligned (-> gather), per-wave strips (rounds 2-4), workgroup strips beyond 2x
                    s, d = DevPlanes(src, align), DevPlanes(oracle.alloc(getattr(oracle, dfmt), dw, dh), align)
                    prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
                    try:
                        capi.convert_resize(capi.make_exec(stream_handle()), getattr(capi, sfmt), getattr(capi, dfmt), 1, 0,
                                            sw, sh, s.desc(), dw, dh, d.desc())
                    finally:
                        capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
                    torch.cuda.synchronize()
                    got, intact = d.download()
                    assert intact
                    assert_planes_equal(got, want, f"fused {sfmt}->{dfmt} {sw}x{sh}->{dw}x{dh} v{variant} a{align}")


def test_fused_convert_resize_batch(capi, oracle):
    """vpf_convert_resize_batch over 35 frames (2 dispatches) == the unfused two-step result per frame"""
    sw, sh, dw, dh, n = 640, 360, 213, 120, 35
    srcs = [oracle.synth(oracle.NV12, sw, sh, 3000 + i) for i in range(n)]
    S = [DevPlanes(s) for s in srcs]
    D = [DevPlanes(oracle.alloc(oracle.BGR, dw, dh)) for _ in range(n)]
    batch = capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)])
    capi.convert_resize_batch(capi.make_exec(stream_handle()), capi.NV12, capi.BGR, 1, 0, sw, sh, dw, dh, batch)
    torch.cuda.synchronize()
    for i in (0, 13, 31, 32, 34):
        got, intact = D[i].download()
        _, mid = oracle.convert(oracle.NV12, oracle.BGR, 1, 0, sw, sh, srcs[i])
        _, want = oracle.resize(oracle.BGR, oracle.LINEAR, sw, sh, mid, dw, dh)
        assert intact
        assert_planes_equal(got, want, f"fused batch frame {i}")


@pytest.mark.parametrize("sf,df,sw,sh,dw,dh,n", [("NV12", "RGB", 1920, 1080, 800, 450, 32), ("YUV420", "RGB_PLANAR", 1280, 720, 500, 300, 64), ("NV12", "BGR", 1920, 1080, 803, 401, 33)])
def test_fused_row_band_by_policy(capi, oracle, sf, df, sw, sh, dw, dh, n):
    """factors beyond ~2x on launches that cover the chip: the POLICY takes the per-tap kernel's row-band form (k_convert_resize_band, round 6: four
    rows per wave, column taps once, the next row's strips in flight, pairs across the two source rows).  Every frame == convert-then-resize
    (the oracle), through the 32- and the 128-frame table, ragged right edge and bottom band included."""
    osf, odf = getattr(oracle, sf), getattr(oracle, df)
    srcs = [oracle.synth(osf, sw, sh, 8800 + i) for i in range(3)]
    wants = [oracle.convert_resize(osf, odf, 1, 0, sw, sh, s, dw, dh)[1] for s in srcs]
    S = [DevPlanes(srcs[i % 3]) for i in range(n)]
    D = [DevPlanes(oracle.alloc(odf, dw, dh, fill=0x5A)) for _ in range(n)]
    capi.convert_resize_batch(capi.make_exec(stream_handle()), getattr(capi, sf), getattr(capi, df), 1, 0, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
    torch.cuda.synchronize()
    for i in range(n):
        got, intact = D[i].download()
        assert intact
        assert_planes_equal(got, wants[i % 3], f"fused row band {sf}->{df} {sw}x{sh}->{dw}x{dh} frame {i} of {n}")


@pytest.mark.parametrize("variant", [0, 47, 48, 49])  # 49: the per-tap kernel's row-band form (round 6) on the same shapes
def test_fused_strip_kernels_at_every_band_height(capi, capi_forms, oracle, variant):
    """k_convert_strip_wg (round 5: one RGB strip per workgroup, conversions dealt out over all 256 lanes) with R = 16 / 8 / 4 / 2 rows per
    wave — the launcher picks R from the strip's LDS bytes and the number of workgroups, so batches of mid-sized frames reach every
    instantiation; ragged right / bottom edges, odd source row parity at the workgroup's first row, sources narrower than one 8-px
    group.  47 = the per-wave strips it replaced, 48 = workgroup strips beyond 2x.  Every frame == convert-then-resize (the oracle)."""
    if variant == 47:  # the per-wave strips live in the lab build of the library (tools/lab/libvpfhip_forms.so); the product refuses the value
        assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 47) == -1
        capi = capi_forms
    cases = [("NV12", "RGB", 640, 360, 1280, 720, 12), ("YUV420", "RGB_PLANAR", 480, 270, 1000, 610, 9), ("NV12", "BGR", 1280, 720, 854, 480, 10),
             ("YUV420", "RGB", 1920, 360, 1288, 239, 8), ("NV12", "RGB_PLANAR", 1280, 720, 1920, 1080, 6), ("NV12", "RGB", 1920, 1080, 800, 450, 3),
             ("NV12", "RGB", 1920, 540, 1600, 450, 6), ("YUV420", "BGR", 16, 8, 300, 170, 5), ("NV12", "RGB", 8, 64, 9, 70, 4), ("NV12", "RGB", 648, 366, 431, 243, 33)]
    prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
    try:
        for sf, df, sw, sh, dw, dh, n in cases:
            srcs = [oracle.synth(getattr(oracle, sf), sw, sh, 5100 + i) for i in range(min(n, 3))]
            wants = [oracle.convert_resize(getattr(oracle, sf), getattr(oracle, df), 1, 0, sw, sh, s_, dw, dh)[1] for s_ in srcs]
            S = [DevPlanes(srcs[i % len(srcs)]) for i in range(n)]
            D = [DevPlanes(oracle.alloc(getattr(oracle, df), dw, dh, fill=9)) for _ in range(n)]
            capi.convert_resize_batch(capi.make_exec(stream_handle()), getattr(capi, sf), getattr(capi, df), 1, 0, sw, sh, dw, dh,
                                      capi.make_batch([(s_.desc(), d_.desc()) for s_, d_ in zip(S, D)]))
            torch.cuda.synchronize()
            for i in sorted({0, 1, n // 2, n - 1}):
                got, intact = D[i].download()
                assert intact
                assert_planes_equal(got, wants[i % len(srcs)], f"fused strips v{variant} {sf}->{df} {sw}x{sh}->{dw}x{dh} frame {i} of {n}")
    finally:
        capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)


def _maps(kind, w, h):
    xm, ym = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    if kind == "identity":
        return xm, ym
    if kind == "shift":
        return xm + 0.5, ym + 0.25
    # barrel distortion r' = r (1 + 0.1 r^2), SURVEY.md §8(d)
    cx, cy = (w - 1) / 2, (h - 1) / 2
    nx, ny = (xm - cx) / cx, (ym - cy) / cy
    k = 1 + 0.1 * (nx * nx + ny * ny)
    return (nx * k * cx + cx).astype(np.float32), (ny * k * cy + cy).astype(np.float32)


@pytest.mark.parametrize("kind", ["identity", "shift", "barrel"])
def test_remap(capi, oracle, kind):
    for (w, h, variant, align) in [(1920, 1080, 0, 256), (1920, 1080, 40, 256), (1920, 1080, 9, 256), (333, 77, 0, 256), (640, 48, 0, 1), (644, 40, 0, 4)]:
        src = oracle.synth(oracle.RGB, w, h, 1060)
        xm, ym = _maps(kind, w, h)
        if kind == "barrel":
            xm[3, 5:9] = np.nan  # NaN coordinates are out of range too
        s = DevPlanes(src, align)
        d = DevPlanes(oracle.alloc(oracle.RGB, w, h, fill=9), align)
        dx, dy = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
        prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
        try:
            capi.remap(capi.make_exec(stream_handle()), capi.RGB, w, h, s.desc()[0], dx.data_ptr(), 4 * w, dy.data_ptr(), 4 * w,
                       w, h, d.desc()[0])
        finally:
            capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
        torch.cuda.synchronize()
        got, intact = d.download()
        assert intact
        _, want = oracle.remap(oracle.RGB, w, h, src, xm, ym, dst=oracle.alloc(oracle.RGB, w, h, fill=9))
        assert_planes_equal(got, want, f"remap {kind} {w}x{h} v{variant} a{align}")
        # the independent link: within 1 LSB of the specification-level oracle (float64 blend of the same float32 coordinates)
        _, exact = oracle.remap(oracle.RGB, w, h, src, xm, ym, mode=oracle.EXACT, dst=oracle.alloc(oracle.RGB, w, h, fill=9))
        assert np.abs(got[0].astype(int) - exact[0].astype(int)).max() <= 1, f"remap {kind} {w}x{h} v{variant}: HIP vs EXACT > 1 LSB"
        if kind == "identity":
            assert np.array_equal(got[0], src[0])


def test_async_on_user_stream(capi, oracle):
    """launches are asynchronous on the caller's stream and ordered with other work on it"""
    w, h = 1920, 1080
    src = oracle.synth(oracle.NV12, w, h, 1070)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s, d = DevPlanes(src), DevPlanes(oracle.alloc(oracle.RGB, w, h))
        ex = capi.make_exec(st.cuda_stream)
        for _ in range(4):
            capi.convert(ex, capi.NV12, capi.RGB, 1, 0, w, h, s.desc(), d.desc())
        st.synchronize()
        got, _ = d.download()
    _, want = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
    assert_planes_equal(got, want, "stream")


@pytest.mark.parametrize("dst", ["RGB", "BGR", "RGB_PLANAR"])
def test_dst_reused_hint_changes_the_store_policy_not_the_pixels(capi, oracle, dst):
    """VPF_EXEC_DST_REUSED (allocating instead of non-temporal stores on single-frame launches): identical bytes, padding
    intact, also where the hinted kernel does not apply (ragged width -> general kernel)"""
    for (w, h) in [(1920, 32), (848, 464), (1002, 6)]:
        src = oracle.synth(oracle.NV12, w, h, 1077)
        s, d = DevPlanes(src), DevPlanes(oracle.alloc(getattr(oracle, dst), w, h))
        ex = capi.make_exec(stream_handle(), flags=capi.EXEC_DST_REUSED)
        capi.convert(ex, capi.NV12, getattr(capi, dst), 1, 0, w, h, s.desc(), d.desc())
        torch.cuda.synchronize()
        got, intact = d.download()
        assert intact
        _, want = oracle.convert(oracle.NV12, getattr(oracle, dst), 1, 0, w, h, src, oracle.FP32)
        assert_planes_equal(got, want, f"dst_reused {dst} {w}x{h}")


def test_plain_c_client_of_the_abi(oracle, tmp_path):
    """include/vpf_hip.h from a plain C99 program (gcc, no C++ / Python / torch in the process): the drop-in boundary as
    any FFI would use it.  tests/c/abi_smoke.c converts a constant frame and reports the pixel; compare with the oracle."""
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("gcc"):
        pytest.skip("no gcc on this box")
    exe = str(tmp_path / "abi_smoke")
    pkg = os.path.join(root, "videoprocessingframework_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"),
                           "-I/opt/rocm/include", os.path.join(root, "tests", "c", "abi_smoke.c"), "-o", exe, "-L" + pkg, "-lvpfhip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    for (y, u, v) in [(16, 128, 128), (235, 128, 128), (81, 90, 240), (145, 54, 34), (0, 0, 0), (255, 255, 255), (120, 200, 30)]:
        r = subprocess.run([exe, str(y), str(u), str(v)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
        lines = r.stdout.strip().split("\n")
        assert lines[-1] == "ok"
        assert tuple(int(t) for t in lines[0].split()) == oracle.yuv2rgb_px(1, 0, y, u, v, oracle.FP32), (y, u, v, lines[0])


def test_abi_is_hip_graph_capturable(capi, oracle):
    """the C ABI never synchronises, allocates or queries the stream, so a chain of per-frame calls can be captured into a
    hipGraph (torch.cuda.CUDAGraph = hipStreamBeginCapture on the ABI's stream) and replayed"""
    w, h = 640, 360
    src = oracle.synth(oracle.NV12, w, h, 1075)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = DevPlanes(src)
        mid, out = DevPlanes(oracle.alloc(oracle.RGB, w, h)), DevPlanes(oracle.alloc(oracle.RGB_PLANAR, 224, 126))
        pl = DevPlanes(oracle.alloc(oracle.RGB_PLANAR, w, h))
        ex = capi.make_exec(st.cuda_stream)

        def chain():
            capi.convert(ex, capi.NV12, capi.RGB, 1, 0, w, h, s.desc(), mid.desc())
            capi.convert(ex, capi.RGB, capi.RGB_PLANAR, 1, 0, w, h, mid.desc(), pl.desc())
            capi.resize(ex, capi.RGB_PLANAR, capi.INTERP_LINEAR, w, h, pl.desc(), 224, 126, out.desc())

        chain(); st.synchronize()   # warm-up outside capture (module load)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            chain()
        for t in mid.bufs + pl.bufs + out.bufs:
            t.fill_(0xCD)
        g.replay(); g.replay(); st.synchronize()
        got, intact = out.download()
    assert intact
    _, a = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, src)
    _, b = oracle.convert(oracle.RGB, oracle.RGB_PLANAR, 1, 0, w, h, a)
    _, c = oracle.resize(oracle.RGB_PLANAR, oracle.LINEAR, w, h, b, 224, 126, oracle.FP32)
    assert_planes_equal(got, c, "graph replay")


def test_batch_entries_are_hip_graph_capturable(capi, oracle):
    """vpf_convert_batch -> vpf_resize_batch (row-band / matrix-core kernels with forced shapes, as a large batch would pick them) -> vpf_remap_batch captured
    into one hipGraph and replayed: the batch entries, too, neither synchronise nor allocate (their frame tables travel in the kernarg)"""
    w, h, dw, dh, n = 640, 360, 427, 240, 5
    srcs = [oracle.synth(oracle.NV12, w, h, 1080 + i) for i in range(n)]
    yy, xx = np.meshgrid(np.arange(dh, dtype=np.float32), np.arange(dw, dtype=np.float32), indexing="ij")
    xm, ym = (xx * 0.9 + 3.25).astype(np.float32), (yy * 0.9 + 1.5).astype(np.float32)
    st = torch.cuda.Stream()
    for interp, band, mfma in ((1, 4, 0), (2, 0, (8 << 8) | 2)):
        with torch.cuda.stream(st):
            S = [DevPlanes(p) for p in srcs]
            M = [DevPlanes(oracle.alloc(oracle.RGB, w, h)) for _ in range(n)]
            R = [DevPlanes(oracle.alloc(oracle.RGB, dw, dh)) for _ in range(n)]
            O = [DevPlanes(oracle.alloc(oracle.RGB, dw, dh, fill=7)) for _ in range(n)]
            dx, dy = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
            ex = capi.make_exec(st.cuda_stream)
            b1 = capi.make_batch([(s.desc(), m.desc()) for s, m in zip(S, M)])
            b2 = capi.make_batch([(m.desc(), r.desc()) for m, r in zip(M, R)])
            b3 = capi.make_batch([(r.desc(), o.desc()) for r, o in zip(R, O)])

            def chain():
                capi.convert_batch(ex, capi.NV12, capi.RGB, 1, 0, w, h, b1)
                capi.resize_batch(ex, capi.RGB, interp, w, h, dw, dh, b2)
                capi.remap_batch(ex, capi.RGB, dw, dh, dx.data_ptr(), 4 * dw, dy.data_ptr(), 4 * dw, dw, dh, b3)

            capi.set_tuning(capi.TUNE_RESIZE_BAND, band); capi.set_tuning(capi.TUNE_RESIZE_MFMA, mfma)
            try:
                chain(); st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    chain()
            finally:
                capi.set_tuning(capi.TUNE_RESIZE_BAND, 0); capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
            for d in M + R:
                for t in d.bufs:
                    t.fill_(0xCD)
            g.replay(); st.synchronize()
            for i in range(n):
                _, a = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, srcs[i])
                _, b = oracle.resize(oracle.RGB, interp, w, h, a, dw, dh, oracle.FP32)
                got, intact = R[i].download()
                assert intact
                assert_planes_equal(got, b, f"graph replay, resize_batch interp {interp} frame {i}")
                _, want = oracle.remap(oracle.RGB, dw, dh, b, xm, ym, dst=oracle.alloc(oracle.RGB, dw, dh, fill=7))  # out-of-range: untouched
                got, intact = O[i].download()
                assert intact
                assert_planes_equal(got, want, f"graph replay, remap_batch frame {i}")


@pytest.mark.parametrize("fmt", ["RGB", "NV12"])
def test_lanczos_upscale_is_hip_graph_capturable_with_cold_tables(capi, oracle, fmt):
    """A Lanczos up-scale (the ring-of-two kernels and their own row-table layout) of a shape nobody has resized before, captured WITHOUT a warm-up:
    the table builds are nodes of the graph (a capturing stream always queues its own), the replay — twice, sources changed in between — writes
    the oracle's pixels."""
    w, h, dw, dh, n = (214, 118, 428, 236, 3) if fmt == "RGB" else (222, 126, 444, 252, 3)
    f, of = getattr(capi, fmt), getattr(oracle, fmt)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        srcs = [oracle.synth(of, w, h, 4200 + i) for i in range(n)]
        S = [DevPlanes(p) for p in srcs]
        D = [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
        ex = capi.make_exec(st.cuda_stream)
        b = capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)])
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            capi.resize_batch(ex, f, 2, w, h, dw, dh, b)
        for rep in range(2):
            if rep:   # new pixels in the same buffers
                srcs = [oracle.synth(of, w, h, 4300 + i) for i in range(n)]
                for s_, p in zip(S, srcs):
                    s_.upload(p)
            for d in D:
                for t in d.bufs:
                    t.fill_(0xCD)
            g.replay(); st.synchronize()
            for i in range(n):
                got, intact = D[i].download()
                assert intact
                assert_planes_equal(got, oracle.resize(of, 2, w, h, srcs[i], dw, dh, oracle.FP32)[1], f"graph replay {rep}, lanczos up-scale {fmt} frame {i}")


# ---------------------------------------------------------------------------------------------
# randomised shape / pitch / alignment fuzz (deterministic seeds): every converter family, bit-exact
# ---------------------------------------------------------------------------------------------
FUZZ_PAIRS = [("NV12", "RGB"), ("NV12", "BGR"), ("NV12", "RGB_PLANAR"), ("YUV420", "RGB"), ("YUV420", "BGR"), ("YUV444", "RGB"),
              ("NV12", "YUV420"), ("YUV420", "NV12"), ("RGB", "RGB_PLANAR"), ("RGB_PLANAR", "RGB"), ("RGB", "BGR"),
              ("RGB", "YUV420"), ("BGR", "YUV444"), ("RGB_PLANAR", "YUV444"), ("BGR", "YCBCR"), ("RGB", "Y"),
              ("YUV444", "BGR"), ("YUV444", "RGB_PLANAR"), ("RGB_PLANAR", "BGR"), ("BGR", "RGB_PLANAR"), ("RGB_PLANAR", "Y"),
              ("NV12", "Y"), ("Y", "YUV444"), ("P10", "NV12"), ("RGB", "RGB_32F"), ("RGB_PLANAR", "YUV420"), ("YUV420", "RGB_PLANAR")]


@pytest.mark.parametrize("seed", _FUZZ_SEEDS)  # soak: VPF_FUZZ_SEEDS=500 (VPF_FUZZ_FIRST=n: seeds n .. n + VPF_FUZZ_SEEDS - 1, fresh cases)
def test_fuzz_shapes_pitches_alignments(capi, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    for _ in range(12):
        s, d = FUZZ_PAIRS[int(rng.integers(len(FUZZ_PAIRS)))]
        kind = int(rng.integers(5))
        if kind == 0:    # "video-like": multiples of 16, aligned
            w, h = 16 * int(rng.integers(1, 130)), 2 * int(rng.integers(1, 40))
            align, extra, offset = 256, 0, 0
        elif kind == 1:  # multiples of 4, dword alignment only
            w, h = 4 * int(rng.integers(1, 300)), 2 * int(rng.integers(1, 30))
            align, extra, offset = 4, 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 4))
        elif kind == 4:  # 16-px regular but only 16-B aligned, padded pitches, shifted bases (r16 kernels off the 256-B grid)
            w, h = 16 * int(rng.integers(1, 200)), 2 * int(rng.integers(1, 24))
            align, extra, offset = 16, 16 * int(rng.integers(0, 4)), 16 * int(rng.integers(0, 5))
        elif kind == 2:  # anything goes
            w, h = int(rng.integers(1, 700)), int(rng.integers(1, 50))
            align, extra, offset = 1, int(rng.integers(0, 5)), int(rng.integers(0, 7))
        else:            # wide and flat / narrow and tall
            w, h = (int(rng.integers(2000, 5000)), int(rng.integers(1, 5))) if rng.integers(2) else (int(rng.integers(1, 9)), int(rng.integers(200, 600)))
            align, extra, offset = 16, 0, 0
        cs = 0 if s in ("RGB", "BGR", "RGB_PLANAR") else int(rng.integers(2))
        cr = int(rng.integers(2))
        if s == "NV12" and d in ("RGB", "BGR", "RGB_PLANAR"):
            variant = int(rng.choice([0, 0, 4, 8, 9, 12, 30, 37, 40, 44, 45, 46]))
        elif s == "YUV420" and d in ("RGB", "BGR", "RGB_PLANAR"):
            variant = int(rng.choice([0, 0, 8, 12, 30, 37, 44, 45, 46, 4, 40, 9]))
        else:
            variant = int(rng.choice([0, 0, 0, 40, 9]))
        src = oracle.synth(getattr(oracle, s), w, h, int(rng.integers(1 << 30)), "ABC"[int(rng.integers(3))])
        _convert(capi, oracle, getattr(capi, s), getattr(capi, d), cs, cr, w, h, src, align, extra, offset, variant=variant)


@pytest.mark.parametrize("seed", _FUZZ_SEEDS)
def test_fuzz_resize_and_fused(capi, capi_forms, oracle, seed):
    """random (format, filter, source size, destination size, alignment, kernel family): the tiled / row-pair / gather
    resize kernels and the LDS / gather fused kernels against the oracle, bit for bit"""
    rng = np.random.default_rng(9000 + seed)
    for _ in range(6):
        sw, sh = int(rng.integers(1, 900)), int(rng.integers(1, 120))
        if rng.integers(3) == 0:
            sw = 16 * int(rng.integers(1, 160))  # wide regular rows: several 64-column tiles / 256-px wave spans
        dw, dh = max(1, int(sw * rng.uniform(0.15, 3.0))), max(1, int(sh * rng.uniform(0.15, 3.0)))
        dw, dh = min(dw, 2600), min(dh, 300)
        if rng.integers(4) == 0:  # exact integer ratios (2x: quad kernel; odd: exact-alignment shortcuts), per axis
            kx, ky = int(rng.integers(1, 6)), int(rng.integers(1, 6))
            dw, dh = max(1, min(dw, 600)), max(1, min(dh, 40))
            sw, sh = dw * kx, dh * ky
        align = int(rng.choice([256, 256, 16, 4, 1]))
        variant = int(rng.choice([0, 0, 40, 43, 9]))
        if rng.integers(4) == 0:  # fused NV12 / YUV420 -> resize -> RGB family
            variant = int(rng.choice([variant, variant, 47, 48, 49]))  # (+ the per-wave strips of rounds 2-4 / workgroup strips beyond 2x / the per-tap kernel's row-band form)
            sw, sh = sw + (sw & 1), sh + (sh & 1)
            sf, df = str(rng.choice(["NV12", "YUV420"])), str(rng.choice(["RGB", "BGR", "RGB_PLANAR"]))
            src = oracle.synth(getattr(oracle, sf), sw, sh, int(rng.integers(1 << 30)))
            _, want = oracle.convert_resize(getattr(oracle, sf), getattr(oracle, df), 1, 0, sw, sh, src, dw, dh)
            _, exact = oracle.convert_resize(getattr(oracle, sf), getattr(oracle, df), 1, 0, sw, sh, src, dw, dh, oracle.EXACT)
            s, d = DevPlanes(src, align), DevPlanes(oracle.alloc(getattr(oracle, df), dw, dh), align)
            lib = capi_forms if variant == 47 else capi  # (the per-wave strips: the lab build of the library)
            prev = lib.set_tuning(lib.TUNE_NV12_RGB_VARIANT, variant)
            try:
                lib.convert_resize(lib.make_exec(stream_handle()), getattr(lib, sf), getattr(lib, df), 1, 0, sw, sh, s.desc(), dw, dh, d.desc())
            finally:
                lib.set_tuning(lib.TUNE_NV12_RGB_VARIANT, prev)
            what = f"fused {sf}->{df}"
        else:
            fmt = str(rng.choice(["RGB", "BGR", "Y", "NV12", "YUV420", "RGB_PLANAR"]))
            interp = int(rng.choice([capi.INTERP_NEAREST, capi.INTERP_LINEAR, capi.INTERP_LINEAR, capi.INTERP_LANCZOS3]))
            if fmt in ("NV12", "YUV420"):
                sw, sh, dw, dh = sw + (sw & 1), sh + (sh & 1), dw + (dw & 1), dh + (dh & 1)
            src = oracle.synth(getattr(oracle, fmt), sw, sh, int(rng.integers(1 << 30)))
            _, want = oracle.resize(getattr(oracle, fmt), interp, sw, sh, src, dw, dh, oracle.FP32)
            # (nearest picks ONE source sample: where the exact coordinate is a tie the fp32 coordinate may pick its neighbour — a whole
            # pixel apart, not an LSB; the specification-level check applies to the two interpolating filters)
            exact = oracle.resize(getattr(oracle, fmt), interp, sw, sh, src, dw, dh, oracle.EXACT)[1] if interp != capi.INTERP_NEAREST else None
            s, d = DevPlanes(src, align), DevPlanes(oracle.alloc(getattr(oracle, fmt), dw, dh), align)
            prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
            try:
                capi.resize(capi.make_exec(stream_handle()), getattr(capi, fmt), interp, sw, sh, s.desc(), dw, dh, d.desc())
            finally:
                capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
            what = f"resize {fmt} interp {interp}"
        torch.cuda.synchronize()
        got, intact = d.download()
        assert intact, what
        assert_planes_equal(got, want, f"{what} {sw}x{sh}->{dw}x{dh} v{variant} a{align}")
        if exact is not None:  # the independent link: within north_star's +-1 LSB of the exact-rational specification, every fuzz case
            assert max(int(np.abs(g.astype(int) - e.astype(int)).max()) for g, e in zip(got, exact)) <= 1, f"{what} {sw}x{sh}->{dw}x{dh}: HIP vs EXACT > 1 LSB"


@pytest.mark.parametrize("seed", _FUZZ_SEEDS)
def test_fuzz_remap(capi, oracle, seed):
    """random source size, map size (!= source size), map family (affine / noisy / mostly out of range, with NaN and Inf
    entries), pixel format, alignment, kernel: out-of-range destinations keep their previous content"""
    rng = np.random.default_rng(11000 + seed)
    for _ in range(4):
        sw, sh = int(rng.integers(1, 700)), int(rng.integers(1, 90))
        dw, dh = int(rng.integers(1, 700)), int(rng.integers(1, 90))
        if rng.integers(3) == 0:
            dw = 4 * int(rng.integers(1, 200))
        yy, xx = np.meshgrid(np.arange(dh, dtype=np.float32), np.arange(dw, dtype=np.float32), indexing="ij")
        fam = int(rng.integers(4))
        if fam == 0:    # affine: rotation + scale about the centre
            a, sc = rng.uniform(-0.6, 0.6), rng.uniform(0.5, 1.8)
            cx, cy = (dw - 1) / 2, (dh - 1) / 2
            xm = (np.cos(a) * (xx - cx) - np.sin(a) * (yy - cy)) * sc * sw / dw + (sw - 1) / 2
            ym = (np.sin(a) * (xx - cx) + np.cos(a) * (yy - cy)) * sc * sh / dh + (sh - 1) / 2
        elif fam == 1:  # independent random coordinates, partly outside
            xm, ym = rng.uniform(-3, sw + 2, (dh, dw)), rng.uniform(-3, sh + 2, (dh, dw))
        elif fam == 2:  # stretched identity with sub-pixel noise
            xm = xx * (sw / dw) + rng.uniform(-0.75, 0.75, (dh, dw))
            ym = yy * (sh / dh) + rng.uniform(-0.75, 0.75, (dh, dw))
        else:           # exact integer and half-integer coordinates, edges included
            xm = np.round(rng.uniform(-1, sw, (dh, dw)) * 2) / 2
            ym = np.round(rng.uniform(-1, sh, (dh, dw)) * 2) / 2
        xm, ym = xm.astype(np.float32), ym.astype(np.float32)
        for _ in range(int(rng.integers(0, 4))):
            xm[int(rng.integers(dh)), int(rng.integers(dw))] = rng.choice([np.nan, np.inf, -np.inf, 1e30])
            ym[int(rng.integers(dh)), int(rng.integers(dw))] = rng.choice([np.nan, np.inf, -np.inf, -1e30])
        fmt = str(rng.choice(["RGB", "BGR"]))
        align, variant = int(rng.choice([256, 16, 4, 1])), int(rng.choice([0, 0, 40, 9]))  # LDS tile / p4 gather / per-pixel
        src = oracle.synth(getattr(oracle, fmt), sw, sh, int(rng.integers(1 << 30)))
        s, d = DevPlanes(src, align), DevPlanes(oracle.alloc(getattr(oracle, fmt), dw, dh, fill=77), align)
        dx, dy = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
        prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
        try:
            capi.remap(capi.make_exec(stream_handle()), getattr(capi, fmt), sw, sh, s.desc()[0], dx.data_ptr(), 4 * dw, dy.data_ptr(), 4 * dw,
                       dw, dh, d.desc()[0])
        finally:
            capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
        torch.cuda.synchronize()
        got, intact = d.download()
        assert intact
        _, want = oracle.remap(getattr(oracle, fmt), sw, sh, src, xm, ym, dst=oracle.alloc(getattr(oracle, fmt), dw, dh, fill=77))
        assert_planes_equal(got, want, f"remap fam{fam} {fmt} {sw}x{sh}->{dw}x{dh} v{variant} a{align}")
        _, exact = oracle.remap(getattr(oracle, fmt), sw, sh, src, xm, ym, mode=oracle.EXACT, dst=oracle.alloc(getattr(oracle, fmt), dw, dh, fill=77))
        assert np.abs(got[0].astype(int) - exact[0].astype(int)).max() <= 1, f"remap fam{fam} {fmt} {sw}x{sh}->{dw}x{dh}: HIP vs EXACT > 1 LSB"


def test_large_frame_8k(capi, oracle):
    """largest practical picture (8192 x 8192, 67 Mpx, 302 MB of traffic in one frame): 32-bit index math, grid limits"""
    w, h = 8192, 8192
    src = oracle.synth(oracle.NV12, w, h, 1090)
    for variant in (0, 4):
        _convert(capi, oracle, capi.NV12, capi.RGB, 1, 0, w, h, src, variant=variant, exact_tol=False)
    _convert(capi, oracle, capi.NV12, capi.YUV420, 1, 0, w, h, src, exact_tol=False)


def test_full_size_batch_checksum_of_checksums(capi, oracle):
    """BASELINE headline shape: 32 x 3840x2160 in ONE dispatch.  Size-independent property: the batch output equals the
    per-frame dispatch output frame by frame (checksum of checksums), and two frames are checked bit-exactly vs the oracle."""
    w, h, n = 3840, 2160, 32
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    src = [torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(n)]
    out_b = [torch.zeros((h, 3 * w), dtype=torch.uint8, device="cuda") for _ in range(n)]
    out_s = [torch.zeros((h, 3 * w), dtype=torch.uint8, device="cuda") for _ in range(n)]
    sd = [[(s.data_ptr(), w), (s.data_ptr() + h * w, w)] for s in src]
    ex = capi.make_exec(stream_handle())
    capi.convert_batch(ex, capi.NV12, capi.RGB, 1, 0, w, h, capi.make_batch([(sd[i], [(out_b[i].data_ptr(), 3 * w)]) for i in range(n)]))
    for i in range(n):
        capi.convert(ex, capi.NV12, capi.RGB, 1, 0, w, h, sd[i], [(out_s[i].data_ptr(), 3 * w)])
    torch.cuda.synchronize()
    sums_b = [int(o.to(torch.int64).sum().item()) ^ int(o[::7, ::13].to(torch.int64).sum().item() << 20) for o in out_b]
    sums_s = [int(o.to(torch.int64).sum().item()) ^ int(o[::7, ::13].to(torch.int64).sum().item() << 20) for o in out_s]
    assert sums_b == sums_s and len(set(sums_b)) == n  # identical per frame, and frames are not aliases of each other
    assert all(torch.equal(a, b) for a, b in zip(out_b, out_s))
    for i in (0, 31):
        a = src[i].cpu().numpy()
        _, want = oracle.convert(oracle.NV12, oracle.RGB, 1, 0, w, h, [np.ascontiguousarray(a[:h]), np.ascontiguousarray(a[h:])])
        assert np.array_equal(out_b[i].cpu().numpy(), want[0]), i


@pytest.mark.parametrize("fmt", ["RGB", "Y", "YUV420", "NV12", "RGB_PLANAR"])
def test_resize_lanczos3(capi, oracle, fmt):
    """Lanczos-3 (the filter the reference resizer requests, Tasks.cpp:1190): bit-exact vs the oracle's FP32 restatement
    (polynomial sin/cos, identical weights), within 1 LSB of the double-precision evaluation"""
    f = getattr(capi, fmt)
    for (sw, sh, dw, dh) in [(640, 360, 224, 224), (100, 60, 333, 201), (64, 64, 64, 64), (1920, 32, 640, 11), (9, 7, 20, 15),
                             (1280, 720, 427, 240), (320, 180, 1280, 720), (2000, 40, 130, 37), (1921, 70, 97, 3), (300, 1000, 150, 20)]:
        src = oracle.synth(f, sw, sh, 1100)
        _, want = oracle.resize(f, oracle.LANCZOS3, sw, sh, src, dw, dh, oracle.FP32)
        for variant, align in ((0, 256), (9, 256), (0, 1)):  # tiled separable kernel, forced gather, unaligned (-> gather)
            s, d = DevPlanes(src, align), DevPlanes(oracle.alloc(f, dw, dh), align)
            prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
            try:
                capi.resize(capi.make_exec(stream_handle()), f, capi.INTERP_LANCZOS3, sw, sh, s.desc(), dw, dh, d.desc())
            finally:
                capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
            torch.cuda.synchronize()
            got, intact = d.download()
            assert intact
            assert_planes_equal(got, want, f"lanczos fmt{fmt} {sw}x{sh}->{dw}x{dh} v{variant} a{align}")
        _, ex = oracle.resize(f, oracle.LANCZOS3, sw, sh, src, dw, dh, oracle.EXACT)
        for g, e in zip(got, ex):
            assert np.abs(g.astype(int) - e.astype(int)).max() <= 1
        if (sw, sh) == (dw, dh):
            assert_planes_equal(got, src, "lanczos identity")


@pytest.mark.parametrize("cs,cr", MATS)
def test_full_size_vs_independent_torch_float64(capi, cs, cr):
    """A third, independent evaluation at BASELINE's full size: the published matrix in torch float64 on the GPU (chroma
    replicated with repeat_interleave, round half up, clamp).  Tolerance: 1 LSB per channel (north_star), and the two may
    differ on at most 1 % of samples (ties: the kernels round half to even, this reference half up)."""
    coef = {(0, 0): (1.164, 16, 1.596, -0.392, -0.813, 2.017), (0, 1): (1.0, 0, 1.140, -0.394, -0.581, 2.032),
            (1, 0): (1.164384, 16, 1.792741, -0.213249, -0.532909, 2.112402), (1, 1): (1.0, 0, 1.5748, -0.187324, -0.468124, 1.8556)}[(cs, cr)]
    cy, off, rv, gu, gv, bu = coef
    w, h = 3840, 2160
    gen = torch.Generator(device="cuda")
    gen.manual_seed(123 + 2 * cs + cr)
    nv12 = torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, device="cuda", generator=gen)
    out = torch.zeros((h, 3 * w), dtype=torch.uint8, device="cuda")
    capi.convert(capi.make_exec(stream_handle()), capi.NV12, capi.RGB, cs, cr, w, h,
                 [(nv12.data_ptr(), w), (nv12.data_ptr() + h * w, w)], [(out.data_ptr(), 3 * w)])
    y = nv12[:h].double() - off
    uv = nv12[h:].view(h // 2, w // 2, 2).double() - 128.0
    u = uv[..., 0].repeat_interleave(2, 0).repeat_interleave(2, 1)
    v = uv[..., 1].repeat_interleave(2, 0).repeat_interleave(2, 1)
    ref = torch.stack([cy * y + rv * v, cy * y + gu * u + gv * v, cy * y + bu * u], dim=-1)
    ref = torch.floor(ref + 0.5).clamp_(0, 255).to(torch.int16).view(h, 3 * w)
    torch.cuda.synchronize()
    d = (out.to(torch.int16) - ref).abs()
    assert int(d.max()) <= 1
    assert float((d > 0).double().mean()) < 0.01


def test_tuning_hook_rejects_values_outside_the_product(capi):
    """vpf_set_tuning accepts only kernels that libvpfhip contains and that write correct pixels; round 1's experimental forms and
    bandwidth probes (which wrote garbage on purpose) are no longer reachable through the public ABI"""
    assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0) >= 0
    for v in (1, 7, 15, 22, 23, 26, 27, 38, 41, 100, -1):
        assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, v) == -1
        assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0) == 0      # unchanged
    assert capi.set_tuning(7, 0) == -1                                    # unknown key
    for v in (4, 8, 9, 12, 30, 37, 40, 43, 44, 45, 46, 48, 49):
        capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, v)
        assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0) == v
    # forms no policy selects are not in the product (lab build: tools/lab/libvpfhip_forms.so): their knob values change nothing here
    assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 47) == -1 and capi.set_tuning(capi.TUNE_RESIZE_BAND, 0x10000) == -1 and capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0x20000) == -1
    assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0) == 0 and capi.set_tuning(capi.TUNE_RESIZE_BAND, 0) == 0 and capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0) == 0


# ---------------------------------------------------------------------------------------------
# vpf_resize_batch / vpf_remap_batch: every plane of every frame in as few dispatches as possible
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", ["RGB", "NV12", "YUV420", "RGB_PLANAR", "Y", "RGB_32F", "RGB_32F_PLANAR"])
@pytest.mark.parametrize("interp", [0, 1, 2])
def test_resize_batch_equals_the_oracle_frame_by_frame(capi, oracle, fmt, interp):
    """n frames through ONE vpf_resize_batch call (multi-plane kernels where every plane lands in the same family, per-plane batched
    launches otherwise) must equal n independent oracle resizes: tiled (Lanczos, up-scale), row-pair (bilinear down-scale), exact 2x,
    odd-integer 3x, gather (unaligned), float surfaces; n = 1, 3 and 33 (two dispatches)"""
    f = getattr(capi, fmt)
    cases = [(640, 360, 224, 224, 256), (320, 180, 1280, 720, 256), (1920, 64, 960, 32, 256), (1152, 48, 384, 16, 256), (128, 72, 50, 30, 1), (100, 60, 333, 201, 256)]
    if fmt.startswith("RGB_32F"):
        cases = [(320, 180, 200, 101, 256), (96, 54, 32, 18, 256)]
    for ci, (sw, sh, dw, dh, align) in enumerate(cases):
        for n in ((1, 3, 33) if ci == 0 else (3,)):
            srcs = [oracle.synth(f, sw, sh, 5000 + i) for i in range(min(n, 4))]
            S = [DevPlanes(srcs[i % len(srcs)], align) for i in range(n)]
            D = [DevPlanes(oracle.alloc(f, dw, dh), align) for _ in range(n)]
            capi.resize_batch(capi.make_exec(stream_handle()), f, interp, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
            torch.cuda.synchronize()
            wants = [oracle.resize(f, interp, sw, sh, s, dw, dh, oracle.FP32)[1] for s in srcs]
            for i in range(n):
                got, intact = D[i].download()
                assert intact
                assert_planes_equal(got, wants[i % len(srcs)], f"resize_batch {fmt} interp {interp} {sw}x{sh}->{dw}x{dh} frame {i} of {n}")


def test_resize_batch_validation(capi, oracle):
    ex = capi.make_exec(stream_handle())
    s, d = DevPlanes(oracle.synth(capi.RGB, 64, 32, 1)), DevPlanes(oracle.alloc(capi.RGB, 32, 16))
    b = capi.make_batch([(s.desc(), d.desc())])
    assert capi.resize_batch(ex, capi.P10, 1, 64, 32, 32, 16, b, check=False) == capi.ERR_UNSUPPORTED
    assert capi.resize_batch(ex, capi.RGB, 7, 64, 32, 32, 16, b, check=False) == capi.ERR_UNSUPPORTED
    assert capi.resize_batch(ex, capi.RGB, 1, 64, 32, 32, 16, b, n=0, check=False) == capi.ERR_BAD_ARG
    assert capi.resize_batch(ex, capi.RGB, 1, 6400, 32, 32, 16, b, check=False) == capi.ERR_BAD_ARG      # pitch < row bytes
    assert capi.remap_batch(ex, capi.NV12, 64, 32, 16, 256, 16, 256, 32, 16, b, check=False) == capi.ERR_UNSUPPORTED
    assert capi.remap_batch(ex, capi.RGB, 64, 32, 0, 256, 16, 256, 32, 16, b, check=False) == capi.ERR_BAD_ARG


@pytest.mark.parametrize("n", [1, 5, 34])
def test_remap_batch_equals_the_oracle_frame_by_frame(capi, oracle, n):
    for (sw, sh, dw, dh, variant, align) in [(640, 360, 640, 360, 0, 256), (333, 77, 200, 61, 0, 256), (640, 48, 320, 24, 9, 256), (644, 40, 644, 40, 0, 4)]:
        rng = np.random.default_rng(n + dw)
        xm = (np.tile(np.arange(dw, dtype=np.float32), (dh, 1)) * (sw / dw) + rng.uniform(-0.7, 0.7, (dh, dw))).astype(np.float32)
        ym = (np.tile(np.arange(dh, dtype=np.float32)[:, None], (1, dw)) * (sh / dh) + rng.uniform(-0.7, 0.7, (dh, dw))).astype(np.float32)
        xm[1, 2:6] = np.nan
        srcs = [oracle.synth(oracle.RGB, sw, sh, 6000 + i) for i in range(min(n, 3))]
        S = [DevPlanes(srcs[i % len(srcs)], align) for i in range(n)]
        D = [DevPlanes(oracle.alloc(oracle.RGB, dw, dh, fill=9), align) for _ in range(n)]
        dx, dy = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
        prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
        try:
            capi.remap_batch(capi.make_exec(stream_handle()), capi.RGB, sw, sh, dx.data_ptr(), 4 * dw, dy.data_ptr(), 4 * dw, dw, dh,
                             capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
        finally:
            capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
        torch.cuda.synchronize()
        wants = [oracle.remap(oracle.RGB, sw, sh, s, xm, ym, dst=oracle.alloc(oracle.RGB, dw, dh, fill=9))[1] for s in srcs]
        for i in range(n):
            got, intact = D[i].download()
            assert intact
            assert_planes_equal(got, wants[i % len(srcs)], f"remap_batch {sw}x{sh}->{dw}x{dh} v{variant} frame {i} of {n}")


@pytest.mark.parametrize("shape", [(16, 4), (16, 8), (32, 8), (64, 8), (8, 4), (24, 4)])
def test_tiled_resize_shapes_write_identical_pixels(capi, oracle, shape):
    """VPF_TUNE_RESIZE_TILE (rows per tile | waves per workgroup << 8) is a measurement knob: every shape the planner could pick writes
    the oracle's pixels; a shape that does not fit falls back (gather form) with the same pixels"""
    ty, wpb = shape
    assert capi.set_tuning(capi.TUNE_RESIZE_TILE, ty | (wpb << 8)) >= 0
    try:
        for fmt, interp, (sw, sh, dw, dh) in [(capi.RGB, 2, (640, 360, 427, 240)), (capi.NV12, 2, (320, 180, 640, 360)), (capi.RGB, 1, (320, 180, 1280, 720)),
                                              (capi.YUV420, 2, (1280, 96, 200, 15))]:
            src = oracle.synth(fmt, sw, sh, 7000)
            s, d = DevPlanes(src), DevPlanes(oracle.alloc(fmt, dw, dh))
            capi.resize(capi.make_exec(stream_handle()), fmt, interp, sw, sh, s.desc(), dw, dh, d.desc())
            torch.cuda.synchronize()
            got, intact = d.download()
            assert intact
            assert_planes_equal(got, oracle.resize(fmt, interp, sw, sh, src, dw, dh, oracle.FP32)[1], f"tile shape {shape} fmt {fmt} interp {interp}")
    finally:
        capi.set_tuning(capi.TUNE_RESIZE_TILE, 0)
    assert capi.set_tuning(capi.TUNE_RESIZE_TILE, 7) == -1 and capi.set_tuning(capi.TUNE_RESIZE_TILE, 16 | (5 << 8)) == -1


@pytest.mark.parametrize("shape", [None, (8, 4), (8, 8), (16, 8), (32, 8), (12, 4)])
def test_lanczos_tile_kernel_writes_the_oracle_pixels(capi, oracle, shape):
    """The tiled separable Lanczos kernel (LanczosTileTask) takes what the matrix-core kernel does not — the strongest down-scales, one small
    frame per dispatch — and, with that kernel switched off (VPF_TUNE_RESIZE_MFMA = 1), everything else as well.  By policy, with the matrix
    cores forced onto single frames too (| 0x40000), and
    with forced tile shapes (VPF_TUNE_RESIZE_TILE): strong and mild down-scales, up-scales, tiles on the left / right image edge (margins
    replicated), clamped rows merged at the top / bottom, pictures smaller than the filter, multi-plane formats in one launch, planes of a
    format that go different ways (NV12 chroma past the window limit), a 33-frame batch, single frames through vpf_resize."""
    cases = [("RGB", 1920, 270, 416, 104, 2), ("RGB", 1280, 720, 224, 224, 3), ("YUV420", 1920, 136, 224, 28, 2), ("NV12", 1280, 72, 300, 20, 2),
             ("RGB", 640, 360, 427, 240, 2), ("NV12", 320, 180, 640, 360, 2), ("Y", 997, 61, 333, 47, 2), ("RGB", 1000, 61, 211, 47, 33),
             ("RGB", 20, 12, 45, 31, 2), ("RGB", 2, 3, 300, 5, 2), ("Y", 700, 2, 64, 1, 2), ("YUV444", 600, 90, 130, 31, 2), ("RGB", 3000, 40, 700, 9, 2)]
    if shape is not None:
        assert capi.set_tuning(capi.TUNE_RESIZE_TILE, shape[0] | (shape[1] << 8)) >= 0
    try:
        for mfma in (0, 1, 0x40000):  # policy (small single frames: the tile kernel) | never the matrix cores | the matrix cores wherever they fit
            capi.set_tuning(capi.TUNE_RESIZE_MFMA, mfma)
            for fmt, sw, sh, dw, dh, n in cases:
                f, of = getattr(capi, fmt), getattr(oracle, fmt)
                srcs = [oracle.synth(of, sw, sh, 7700 + i) for i in range(min(n, 3))]
                wants = [oracle.resize(of, 2, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
                S = [DevPlanes(srcs[i % len(srcs)]) for i in range(n)]
                D = [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
                capi.resize_batch(capi.make_exec(stream_handle()), f, 2, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
                one = DevPlanes(oracle.alloc(of, dw, dh))
                capi.resize(capi.make_exec(stream_handle()), f, 2, sw, sh, S[0].desc(), dw, dh, one.desc())
                torch.cuda.synchronize()
                for i in range(n):
                    got, intact = D[i].download()
                    assert intact
                    assert_planes_equal(got, wants[i % len(srcs)], f"lanczos tile shape {shape} mfma {mfma} {fmt} {sw}x{sh}->{dw}x{dh} frame {i} of {n}")
                got, intact = one.download()
                assert intact
                assert_planes_equal(got, wants[0], f"lanczos tile shape {shape} mfma {mfma} {fmt} {sw}x{sh}->{dw}x{dh} single frame")
    finally:
        capi.set_tuning(capi.TUNE_RESIZE_TILE, 0)
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)


@pytest.mark.parametrize("band", [1, 2, 4, 8, 16, 0x104, 0x204, 0x304, 0x804, 0x10000, 0x10002, 0x10008, 0x10104, 0x10304, 0x20000, 0x20008, 0x20304])
def test_row_band_kernels_write_the_row_pair_pixels(capi, capi_forms, oracle, band):
    """VPF_TUNE_RESIZE_BAND = destination rows per wave of the bilinear row-pair kernels (policy: 16 / 8 / 4 / 2 for launches with >= 2048
    workgroups, 1 otherwise).  Every value writes the oracle's pixels: general and > 2x down-scales, shared and disjoint source rows,
    heights that are not a multiple of the band, one-row pictures, ragged widths, fx == 0 columns (even integer factor on x only), an
    up-scale forced onto the row-pair family (variant 40: repeated source rows), multi-plane formats, and a 33-frame batch.
    4 | nb << 8: the march form (nb 4-row bands per wave, the next band's rows in flight while this one is blended, the walk's lerps carried
    from band to band) on the down-scales of wide 1-channel planes, the plain 4-row form everywhere else.
    | 0x10000 (round 5): the PERSISTENT launch of the same band kernels — the resident workgroups pull (frame, plane, wave row, column chunk)
    items from the stream's work counters (k_planes_mp_persist): more items than waves, fewer items than waves, one item."""
    cases = [("RGB", 640, 360, 427, 240, 0, 3), ("RGB", 1920, 96, 416, 37, 0, 2), ("NV12", 1280, 72, 854, 48, 0, 3), ("YUV420", 642, 90, 300, 31, 0, 2),
             ("RGB", 300, 5, 200, 1, 0, 2), ("Y", 997, 61, 333, 47, 0, 2), ("RGB", 512, 90, 128, 61, 0, 2), ("RGB", 200, 50, 333, 77, 40, 2),
             ("NV12", 200, 50, 320, 96, 40, 2), ("RGB", 640, 360, 224, 224, 0, 33), ("RGB", 1919, 64, 1280, 43, 0, 2),
             ("RGB", 320, 180, 1280, 720, 0, 2), ("YUV420", 96, 54, 160, 90, 0, 3),                                # up-scales (the band family by policy when forced)
             ("Y", 1500, 40, 1000, 27, 0, 2), ("YUV444", 600, 40, 500, 31, 0, 2), ("NV12", 1200, 40, 2040, 68, 0, 2),  # 1-channel planes that 512-column chunks fill well: 8 px per lane
             ("Y", 1920, 300, 1280, 200, 0, 3), ("NV12", 1920, 270, 1280, 180, 0, 2), ("Y", 1030, 131, 1025, 67, 0, 2), ("NV12", 1600, 98, 1100, 66, 0, 2),  # ... down-scales: the march form
             ("Y", 1100, 77, 1100, 77, 0, 2), ("YUV444", 1536, 50, 1024, 34, 0, 2), ("Y", 2000, 9, 1999, 5, 0, 2),
             ("RGB", 1, 1, 9, 7, 0, 2), ("RGB", 2, 3, 300, 5, 0, 2), ("Y", 3, 2, 5, 70, 0, 2), ("NV12", 4, 4, 18, 10, 0, 2), ("RGB", 5, 2, 3, 1, 0, 2)]  # tiny pictures
    if band & 0x10000:  # the persistent launch lives in the lab build of the library
        capi = capi_forms
    assert capi.set_tuning(capi.TUNE_RESIZE_BAND, band) >= 0
    try:
        for fmt, sw, sh, dw, dh, variant, n in cases:
            f, of = getattr(capi, fmt), getattr(oracle, fmt)
            srcs = [oracle.synth(of, sw, sh, 7100 + i) for i in range(min(n, 3))]
            S = [DevPlanes(srcs[i % len(srcs)]) for i in range(n)]
            D = [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
            prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, variant)
            try:
                capi.resize_batch(capi.make_exec(stream_handle()), f, 1, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
            finally:
                capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
            torch.cuda.synchronize()
            wants = [oracle.resize(of, 1, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
            for i in range(n):
                got, intact = D[i].download()
                assert intact
                assert_planes_equal(got, wants[i % len(srcs)], f"band {band} {fmt} {sw}x{sh}->{dw}x{dh} v{variant} frame {i} of {n}")
    finally:
        capi.set_tuning(capi.TUNE_RESIZE_BAND, 0)
    assert capi.set_tuning(capi.TUNE_RESIZE_BAND, 3) == -1 and capi.set_tuning(capi.TUNE_RESIZE_BAND, 32) == -1
    assert capi.set_tuning(capi.TUNE_RESIZE_BAND, 0x208) == -1 and capi.set_tuning(capi.TUNE_RESIZE_BAND, 0x904) == -1  # bands per wave: 4-row bands only, at most 8
    assert capi.set_tuning(capi.TUNE_RESIZE_BAND, 0x100004) == -1 and capi.set_tuning(capi.TUNE_RESIZE_BAND, 0x40004) == -1                                                     # forms: | 0x10000 the persistent launch, | 0x20000 eight pixels per lane on every 1-channel plane


@pytest.mark.parametrize("fmt", ["Y", "NV12"])
def test_bilinear_march_form_by_policy(capi, oracle, fmt):
    """a launch large enough that the POLICY picks the march form for the 1-channel plane (32 frames of 1080p -> 720p: 8-row bands would
    run; instead 4-row bands, two per wave): every frame equals the oracle"""
    sw, sh, dw, dh, n = 1920, 1080, 1280, 720, 32
    f, of = getattr(capi, fmt), getattr(oracle, fmt)
    srcs = [oracle.synth(of, sw, sh, 7300 + i) for i in range(3)]
    wants = [oracle.resize(of, 1, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
    S = [DevPlanes(srcs[i % 3]) for i in range(n)]
    D = [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
    capi.resize_batch(capi.make_exec(stream_handle()), f, 1, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
    torch.cuda.synchronize()
    for i in range(n):
        got, intact = D[i].download()
        assert intact
        assert_planes_equal(got, wants[i % 3], f"march by policy {fmt} frame {i}")


@pytest.mark.parametrize("fmt,knob,sizes", [("Y", 0x10104, (1920, 1080, 1280, 720)), ("NV12", 0x10204, (1920, 1080, 1280, 720)), ("YUV420", 0x10004, (1280, 720, 854, 480)),
                                            ("RGB", 0x10010, (640, 360, 1280, 720)), ("RGB", 0x10008, (1280, 720, 854, 480))])
def test_persistent_band_launch_with_more_items_than_waves_on_two_streams(capi_forms, oracle, fmt, knob, sizes):
    """k_planes_mp_persist (round 5) where it matters: 32-frame dispatches whose wave items outnumber the resident waves several times (every
    wave pulls many items; the XCDs finish their own eighth and help the others; the last ticket of each counter puts it back to zero for
    the next launch), three dispatches in a row per stream, two streams at once (a slot of counters each: vpf_persist.h), then the same
    streams again after a synchronisation.  Every frame equals the oracle; a hipGraph capture takes the plain grid (same pixels).
    Round 6: the second form (chunks, a static first chunk, prefetched tickets, two counter sets per stream taking turns) — in the lab build."""
    capi = capi_forms
    sw, sh, dw, dh = sizes
    f, of = getattr(capi, fmt), getattr(oracle, fmt)
    n = 32
    srcs = [oracle.synth(of, sw, sh, 7900 + i) for i in range(2)]
    wants = [oracle.resize(of, 1, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    assert capi.set_tuning(capi.TUNE_RESIZE_BAND, knob) >= 0
    try:
        jobs = []
        for rnd in range(2):
            for si, st in enumerate(streams):
                with torch.cuda.stream(st):
                    S = [DevPlanes(srcs[(i + si) % 2]) for i in range(n)]
                    ex = capi.make_exec(st.cuda_stream)
                    for rep in range(3):
                        D = [DevPlanes(oracle.alloc(of, dw, dh, fill=3)) for _ in range(n)]
                        capi.resize_batch(ex, f, 1, sw, sh, dw, dh, capi.make_batch([(s_.desc(), d_.desc()) for s_, d_ in zip(S, D)]))
                        jobs.append((si, S, D))
            torch.cuda.synchronize()
        st = streams[0]
        with torch.cuda.stream(st):
            S = [DevPlanes(srcs[i % 2]) for i in range(n)]
            D = [DevPlanes(oracle.alloc(of, dw, dh, fill=3)) for _ in range(n)]
            b = capi.make_batch([(s_.desc(), d_.desc()) for s_, d_ in zip(S, D)])
            ex = capi.make_exec(st.cuda_stream)
            capi.resize_batch(ex, f, 1, sw, sh, dw, dh, b); st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                capi.resize_batch(ex, f, 1, sw, sh, dw, dh, b)
            for d_ in D:
                for t in d_.bufs:
                    t.fill_(0xCD)
            g.replay(); st.synchronize()
            jobs.append((0, S, D))
        torch.cuda.synchronize()
    finally:
        capi.set_tuning(capi.TUNE_RESIZE_BAND, 0)
    for ji, (si, S, D) in enumerate(jobs):
        for i in (0, 1, 13, 30, 31):
            got, intact = D[i].download()
            assert intact
            assert_planes_equal(got, wants[(i + si) % 2] if ji < len(jobs) - 1 else wants[i % 2], f"persistent {fmt} knob {knob:#x} job {ji} frame {i}")


@pytest.mark.parametrize("fmt,interp,sizes", [("RGB", 1, (1920, 1080, 1280, 720)), ("RGB", 2, (1920, 1080, 1280, 720)), ("NV12", 1, (1920, 1080, 1280, 720)),
                                              ("NV12", 2, (1920, 1080, 1280, 720)), ("RGB", 1, (1280, 720, 1920, 1080)), ("RGB", 2, (1280, 720, 1920, 1080)),
                                              ("RGB", 2, (3840, 2160, 1920, 1080)), ("RGB", 1, (7680, 4320, 5120, 2880)), ("RGB", 2, (7680, 4320, 5120, 2880)),
                                              ("NV12", 2, (7680, 4320, 3840, 2160))])
def test_batched_resize_at_full_size_with_the_kernels_the_policy_picks(capi, oracle, fmt, interp, sizes):
    """the row-band / march kernels are chosen by POLICY only for large launches (the tests above force them on small pictures): 32
    full-size frames per dispatch, no tuning — every frame must equal the oracle (two distinct pictures alternate through the batch)"""
    sw, sh, dw, dh = sizes
    f, of = getattr(capi, fmt), getattr(oracle, fmt)
    n = 32 if sw < 7000 else 8   # 8K: eight frames still leave thousands of workgroups
    srcs = [oracle.synth(of, sw, sh, 7400 + i) for i in range(2)]
    S = [DevPlanes(srcs[i % 2]) for i in range(n)]
    D = [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
    capi.resize_batch(capi.make_exec(stream_handle()), f, interp, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
    torch.cuda.synchronize()
    wants = [oracle.resize(of, interp, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
    for i in ((0, 1, 2, 15, 30, 31) if n == 32 else (0, 1, 6, 7)):
        got, intact = D[i].download()
        assert intact
        assert_planes_equal(got, wants[i % 2], f"policy batch {fmt} interp {interp} {sw}x{sh}->{dw}x{dh} frame {i}")


@pytest.mark.parametrize("shape", [1, (4 << 8) | 1, (4 << 8) | 3, (8 << 8) | 1, (8 << 8) | 2, (8 << 8) | 64, 5,
                                   0x10000, 0x10000 | (4 << 8) | 1, 0x10000 | (8 << 8) | 2, 0x10000 | 5])
def test_lanczos_mfma_kernel_shapes_write_the_oracle_pixels(capi, oracle, shape):
    """VPF_TUNE_RESIZE_MFMA = N-tiles per wave << 8 | 16-row destination tiles per band of the matrix-core Lanczos kernel (1 = never: the
    gather form), | 0x10000 = filter weights evaluated inside the kernel instead of loaded from the per-shape tables (the path a full table
    arena takes).  Every value writes the oracle's pixels: down- and up-scales, strips on the left / right image edge (clamped taps merged
    on the edge sample) and pictures narrower than one strip or one N-tile, bands cut by the bottom edge, factors the kernel's 64-B
    window / four-tile ring cannot hold (-> gather form), multi-plane formats (chroma planes with their own factors and channel counts),
    heights below one tile, a 33-frame batch"""
    cases = [("RGB", 640, 360, 427, 240, 3), ("RGB", 320, 180, 1280, 720, 2), ("NV12", 1280, 72, 854, 48, 3), ("YUV420", 642, 90, 300, 31, 2),
             ("RGB", 300, 50, 200, 7, 2), ("Y", 997, 61, 333, 47, 2), ("RGB", 96, 54, 700, 33, 2), ("RGB", 640, 360, 224, 224, 33), ("RGB", 1919, 64, 1280, 43, 2),
             ("YUV444", 100, 60, 333, 201, 2), ("RGB", 20, 12, 45, 31, 2), ("RGB", 1280, 200, 640, 100, 2), ("NV12", 640, 400, 320, 200, 2), ("RGB", 500, 300, 233, 140, 2),
             ("Y", 2000, 100, 701, 43, 2), ("NV12", 854, 480, 1280, 720, 2), ("RGB", 1000, 37, 1000, 37, 2),
             # pictures smaller than the filter: every tap of some columns / rows is a clamped edge sample
             ("RGB", 1, 1, 9, 7, 2), ("RGB", 2, 3, 300, 5, 2), ("Y", 3, 2, 5, 70, 2), ("NV12", 4, 4, 18, 10, 2), ("RGB", 5, 1, 3, 1, 2), ("Y", 700, 2, 64, 1, 2)]
    assert capi.set_tuning(capi.TUNE_RESIZE_MFMA, shape) >= 0
    try:
        for fmt, sw, sh, dw, dh, n in cases:
            f, of = getattr(capi, fmt), getattr(oracle, fmt)
            srcs = [oracle.synth(of, sw, sh, 7300 + i) for i in range(min(n, 3))]
            S = [DevPlanes(srcs[i % len(srcs)]) for i in range(n)]
            D = [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
            capi.resize_batch(capi.make_exec(stream_handle()), f, 2, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
            torch.cuda.synchronize()
            wants = [oracle.resize(of, 2, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
            for i in range(n):
                got, intact = D[i].download()
                assert intact
                assert_planes_equal(got, wants[i % len(srcs)], f"mfma shape {shape:#x} {fmt} {sw}x{sh}->{dw}x{dh} frame {i} of {n}")
    finally:
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
    assert capi.set_tuning(capi.TUNE_RESIZE_MFMA, (3 << 8) | 2) == -1 and capi.set_tuning(capi.TUNE_RESIZE_MFMA, -1) == -1 and capi.set_tuning(capi.TUNE_RESIZE_MFMA, (8 << 8) | 65) == -1


@pytest.mark.parametrize("knob", [0, (8 << 8) | 1, (8 << 8) | 5, (8 << 8) | 64, (4 << 8) | 2, (4 << 8) | 9, 0x10000, 0x10000 | (8 << 8) | 3, 0x10000 | (4 << 8) | 7,
                                  0x80000, 0x80000 | (8 << 8) | 5, 0x80000 | 0x10000 | (4 << 8) | 2])
def test_lanczos_upscales_with_the_ring_of_two(capi, oracle, knob):
    """Up-scales whose 16-row destination tiles find their source rows in two consecutive 16-row source tiles march with a ring of TWO
    (LanczosMfmaTask<.., UP2>: overlapping two-tile chunks in registers, one K chunk in pass 2, its own row-table layout), | 0x80000 with the
    ring of four like everything else: the oracle's pixels either way, with tables and without (| 0x10000), 8- and 4-tile strips, bands of
    one tile .. the whole column (several weight groups per band), heights that end inside a tile and inside a group, factors from 1.05 to
    9 (the mildest ones fail the two-tile bound on some tiles and keep the ring of four: both kernels run in this test), formats whose
    chroma planes have their own sizes, 33 frames per dispatch, a mixed launch (up-scale in y only)."""
    cases = [("RGB", 320, 180, 640, 360, 3), ("RGB", 480, 270, 720, 405, 2), ("Y", 333, 217, 1000, 651, 2), ("NV12", 426, 240, 1280, 720, 2), ("YUV420", 320, 180, 854, 480, 2),
             ("YUV444", 100, 60, 333, 201, 2), ("RGB", 200, 300, 420, 333, 2), ("RGB", 160, 90, 1440, 810, 2), ("Y", 640, 100, 1280, 131, 2), ("RGB", 320, 200, 336, 211, 2),
             ("RGB", 64, 36, 96, 54, 33), ("Y", 50, 1000, 75, 1500, 2), ("RGB", 300, 40, 200, 97, 2), ("NV12", 640, 360, 1920, 1080, 2), ("RGB", 7, 5, 40, 33, 2),
             # 1.5 x on a launch large enough for the WIDE 8-tile ring-of-two strips (LzMfma8uw: the planner's volume gate; forced 8-tile shapes below it keep the ring of four)
             ("RGB", 1280, 720, 1920, 1080, 32), ("NV12", 1280, 720, 1920, 1080, 48)]
    assert capi.set_tuning(capi.TUNE_RESIZE_MFMA, knob) >= 0
    try:
        for fmt, sw, sh, dw, dh, n in cases:
            f, of = getattr(capi, fmt), getattr(oracle, fmt)
            srcs = [oracle.synth(of, sw, sh, 9100 + i) for i in range(min(n, 3))]
            S = [DevPlanes(srcs[i % len(srcs)]) for i in range(n)]
            D = [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
            capi.resize_batch(capi.make_exec(stream_handle()), f, 2, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
            torch.cuda.synchronize()
            wants = [oracle.resize(of, 2, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
            for i in (range(n) if n <= 33 else (0, 1, 2, 31, 32, 33, n - 1)):
                got, intact = D[i].download()
                assert intact
                assert_planes_equal(got, wants[i % len(srcs)], f"ring of two, knob {knob:#x} {fmt} {sw}x{sh}->{dw}x{dh} frame {i} of {n}")
            del S, D
            torch.cuda.empty_cache()
    finally:
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
    assert capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0x100000) == -1


@pytest.mark.parametrize("knob", [0, 0x10000, (4 << 8) | 1, 0x10000 | (4 << 8) | 3])
def test_lanczos_two_chunk_windows_write_the_oracle_pixels(capi, oracle, knob):
    """Horizontal factors of ~2.2 .. 6 (1080p -> 416 x 416 in front of a network): the taps of 16 destination bytes spread over more than 64
    source bytes, and the matrix-core kernel takes them with 128-B windows — pass 1 chains two MFMAs per product (LzMfma4k4 / k6 / k8 by
    the length of the staged rows).  Vertical factors of ~2.9 .. 6 (thumbnails): half tiles.  Tables and in-kernel weights, policy and forced band heights: 1-, 2- and 3-channel planes, every
    staging width, ragged widths (partial last strip / last tile), picture edges, multi-plane formats, a 33-frame batch."""
    cases = [("Y", 1408, 90, 640, 41, 2), ("RGB", 1920, 270, 416, 104, 3), ("RGB", 2200, 100, 400, 45, 2), ("NV12", 1920, 270, 640, 120, 2),
             ("YUV420", 1280, 180, 400, 70, 2), ("RGB", 1999, 131, 417, 51, 2), ("Y", 1920, 1080, 416, 416, 2), ("RGB", 640, 360, 224, 224, 33),
             ("YUV444", 1000, 64, 217, 29, 2), ("Y", 3000, 40, 520, 17, 2), ("RGB", 700, 33, 130, 13, 2), ("NV12", 3840, 128, 1000, 50, 2),
             # vertical factors of ~2.9 .. 6: HALF tiles (8 destination rows per 16-row MFMA tile), with one- and two-chunk windows
             ("RGB", 1920, 1080, 480, 270, 2), ("Y", 1280, 720, 224, 224, 2), ("NV12", 1920, 1080, 384, 216, 2), ("YUV420", 1280, 720, 224, 224, 2),
             ("RGB", 640, 1000, 427, 201, 3), ("Y", 500, 900, 700, 190, 2), ("RGB", 1000, 333, 250, 71, 2), ("Y", 300, 599, 120, 101, 33),
             # horizontal factors of ~6 .. 10: three-chunk (192-B) windows on 2-tile strips (LzMfma2k6 / 2k8) — the reference's sample resize
             ("YUV420", 1920, 1080, 224, 224, 2), ("RGB", 1920, 270, 224, 56, 3), ("NV12", 3840, 540, 416, 104, 2), ("Y", 2000, 100, 201, 37, 2),
             ("RGB", 3840, 200, 416, 77, 2), ("YUV444", 1500, 64, 170, 13, 2), ("RGB", 1283, 90, 131, 19, 33)]
    assert capi.set_tuning(capi.TUNE_RESIZE_MFMA, knob) >= 0
    try:
        for fmt, sw, sh, dw, dh, n in cases:
            f, of = getattr(capi, fmt), getattr(oracle, fmt)
            srcs = [oracle.synth(of, sw, sh, 7500 + i) for i in range(min(n, 3))]
            S = [DevPlanes(srcs[i % len(srcs)]) for i in range(n)]
            D = [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
            capi.resize_batch(capi.make_exec(stream_handle()), f, 2, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
            torch.cuda.synchronize()
            wants = [oracle.resize(of, 2, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
            exact = oracle.resize(of, 2, sw, sh, srcs[0], dw, dh, oracle.EXACT)[1]
            for i in range(n):
                got, intact = D[i].download()
                assert intact
                assert_planes_equal(got, wants[i % len(srcs)], f"two-chunk windows knob {knob:#x} {fmt} {sw}x{sh}->{dw}x{dh} frame {i} of {n}")
            got0 = D[0].download()[0]
            assert max(int(np.abs(a.astype(np.int16) - b.astype(np.int16)).max()) for a, b in zip(got0, exact)) <= 1
    finally:
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)


def test_kernel_selection_of_the_round_4_forms():
    """which kernel a launch takes is policy, and policy regressions are silent (same pixels): the launch log (VPF_HIP_LOG=2) names the
    two-chunk matrix-core Lanczos kernel for a network-input down-scale, the tile kernel for one small Lanczos plane (or small multi-plane
    frame) per dispatch, the three-chunk form for the batched sample resize, and the march form of the row-band bilinear kernel for a
    large batch of Y planes"""
    import subprocess
    code = f"""
import sys
sys.path.insert(0, {ROOT!r})
import torch
from videoprocessingframework_amd import capi
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
def planes(w, h, c):
    p = (w * c + 255) // 256 * 256
    t = torch.zeros((h, p), dtype=torch.uint8, device="cuda")
    return t, [(t.data_ptr(), p)]
def batch(fmt, c, interp, sw, sh, dw, dh, n):
    S = [planes(sw, sh, c) for _ in range(n)]; D = [planes(dw, dh, c) for _ in range(n)]
    capi.resize_batch(ex, fmt, interp, sw, sh, dw, dh, capi.make_batch([(s[1], d[1]) for s, d in zip(S, D)]))
    torch.cuda.synchronize()
    return S, D
print("A", file=sys.stderr); batch(capi.RGB, 3, 2, 1920, 1080, 416, 416, 4)
print("B", file=sys.stderr); s, d = planes(1920, 1080, 3), planes(1280, 720, 3)
capi.resize(ex, capi.RGB, 2, 1920, 1080, capi.planes(s[1]), 1280, 720, capi.planes(d[1])); torch.cuda.synchronize()
print("C", file=sys.stderr); batch(capi.Y, 1, 1, 1920, 1080, 1280, 720, 32)
def yuv420(w, h):
    t = [planes(w, h, 1), planes(w // 2, h // 2, 1), planes(w // 2, h // 2, 1)]
    return t, [q[1][0] for q in t]
print("D", file=sys.stderr)
S = [yuv420(1920, 1080) for _ in range(8)]; D = [yuv420(224, 224) for _ in range(8)]
capi.resize(ex, capi.YUV420, 2, 1920, 1080, capi.planes(S[0][1]), 224, 224, capi.planes(D[0][1])); torch.cuda.synchronize()
print("E", file=sys.stderr)
capi.resize_batch(ex, capi.YUV420, 2, 1920, 1080, 224, 224, capi.make_batch([(s[1], d[1]) for s, d in zip(S, D)])); torch.cuda.synchronize()
print("done")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, VPF_HIP_LOG="2"), timeout=300)
    assert r.returncode == 0 and "done" in r.stdout, r.stdout + r.stderr
    a, rest = r.stderr.split("\nB\n", 1)[0], r.stderr.split("\nB\n", 1)[1]
    b, rest = rest.split("\nC\n", 1)
    c, rest = rest.split("\nD\n", 1)
    d, e = rest.split("\nE\n", 1)
    assert "LzMfma4k" in a, a
    assert "k_resize_lztile" in b and "k_lanczos_mfma" not in b, b
    assert "RowBand4wm" in c, c
    assert "TileLz" in d and "k_lanczos_mfma" not in d, d   # the reference sample's resize (YUV420 1080p -> 224 x 224), one frame per dispatch: the tile kernel
    assert "LzMfma2k" in e, e                                # ... batched: three-chunk windows on the matrix cores


def test_lanczos_weight_tables_across_streams(capi, oracle):
    """The matrix-core Lanczos kernel loads its filter weights from per-shape tables that the first launch of a shape builds on ITS stream.
    A second stream using the same shape right away — before the first stream's build need have run — queues its own build instead of
    reading a table that may not exist yet; a shape first met under one band height and then another gets a second row table.  Several
    shapes on several streams, issued back to back with no synchronisation in between: every frame equals the oracle."""
    streams = [torch.cuda.Stream() for _ in range(3)]
    shapes = [("RGB", 1283, 211, 857, 140), ("NV12", 1920, 240, 1280, 160), ("RGB", 1283, 211, 857, 140), ("Y", 811, 97, 1622, 194), ("RGB", 1283, 211, 640, 97)]
    jobs = []
    for rnd, knob in enumerate((0, (8 << 8) | 2, (4 << 8) | 3)):
        capi.set_tuning(capi.TUNE_RESIZE_MFMA, knob)
        try:
            for si, st in enumerate(streams):
                for ci, (fmt, sw, sh, dw, dh) in enumerate(shapes):
                    f, of = getattr(capi, fmt), getattr(oracle, fmt)
                    src = oracle.synth(of, sw, sh, 9100 + 10 * rnd + ci)
                    with torch.cuda.stream(st):
                        S, D = DevPlanes(src), DevPlanes(oracle.alloc(of, dw, dh))
                        capi.resize_batch(capi.make_exec(st.cuda_stream), f, 2, sw, sh, dw, dh, capi.make_batch([(S.desc(), D.desc())]))
                    jobs.append((fmt, of, sw, sh, dw, dh, src, S, D, si, knob))
        finally:
            capi.set_tuning(capi.TUNE_RESIZE_MFMA, 0)
    torch.cuda.synchronize()
    for fmt, of, sw, sh, dw, dh, src, S, D, si, knob in jobs:
        got, intact = D.download()
        assert intact
        assert_planes_equal(got, oracle.resize(of, 2, sw, sh, src, dw, dh, oracle.FP32)[1], f"tables across streams: {fmt} {sw}x{sh}->{dw}x{dh} stream {si} knob {knob:#x}")


def test_resize_with_a_caller_owned_workspace(capi, oracle):
    """vpf_resize_workspace_bytes / vpf_resize_ws / vpf_resize_batch_ws (include/vpf_hip.h): the Lanczos tables of a shape live in memory the
    CALLER owns (NPP's scratch-buffer pattern; ResizeSurface allocates one next to its destination surface).  One workspace serves shape
    after shape (a new shape is built behind the old one's readers, stream-ordered), batches and single frames, two streams in turn; a
    workspace that is too small, misaligned or absent falls back to the library's arena; filters without tables ask for none.  Every
    result equals the oracle."""
    assert capi.resize_workspace_bytes(capi.RGB, capi.INTERP_LINEAR, 1920, 1080, 1280, 720) == 0
    assert capi.resize_workspace_bytes(capi.RGB_32F, capi.INTERP_LANCZOS3, 640, 360, 320, 180) == 0
    need = capi.resize_workspace_bytes(capi.RGB, capi.INTERP_LANCZOS3, 1920, 1080, 1280, 720)
    assert 256 * 1024 < need < 4 * 1024 * 1024
    assert capi.resize_workspace_bytes(capi.NV12, capi.INTERP_LANCZOS3, 1920, 1080, 1280, 720) > need // 3
    shapes = [("RGB", 1283, 211, 857, 140), ("NV12", 1920, 240, 1280, 160), ("Y", 811, 97, 1622, 194), ("RGB", 1283, 211, 640, 97), ("YUV420", 642, 130, 1000, 200),
              ("RGB", 1920, 270, 416, 104), ("YUV420", 1920, 540, 224, 112), ("NV12", 1920, 540, 480, 136)]   # two- / three-chunk windows, half tiles: larger column tables
    big = max(capi.resize_workspace_bytes(getattr(capi, f), 2, sw, sh, dw, dh) for f, sw, sh, dw, dh in shapes)
    mem = torch.zeros(big + 512, dtype=torch.uint8, device="cuda")
    base = (mem.data_ptr() + 255) // 256 * 256
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    variants = {"fits": capi.make_workspace(base, big), "small": capi.make_workspace(base, 4096), "misaligned": capi.make_workspace(base + 16, big - 16), "none": None}
    jobs = []
    for rnd in range(3):
        for ci, (fmt, sw, sh, dw, dh) in enumerate(shapes):
            f, of = getattr(capi, fmt), getattr(oracle, fmt)
            for name, ws in variants.items():
                st = streams[(rnd + ci) & 1] if name == "fits" else streams[0]   # the good workspace changes streams as it goes: that rebuilds, never corrupts
                n = 1 + (rnd + ci) % 3
                srcs = [oracle.synth(of, sw, sh, 9500 + 100 * rnd + 10 * ci + i) for i in range(n)]
                with torch.cuda.stream(st):
                    S, D = [DevPlanes(p) for p in srcs], [DevPlanes(oracle.alloc(of, dw, dh)) for _ in range(n)]
                    ex = capi.make_exec(st.cuda_stream)
                    if n == 1:
                        capi.resize_ws(ex, f, 2, sw, sh, S[0].desc(), dw, dh, D[0].desc(), ws)
                    else:
                        capi.resize_batch_ws(ex, f, 2, sw, sh, dw, dh, capi.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]), ws)
                if name == "fits":
                    torch.cuda.synchronize()  # (the workspace is used on one stream AT A TIME: the next round may run on the other stream)
                jobs.append((name, of, sw, sh, dw, dh, srcs, S, D))
    torch.cuda.synchronize()
    for name, of, sw, sh, dw, dh, srcs, S, D in jobs:
        for src, d in zip(srcs, D):
            got, intact = d.download()
            assert intact
            assert_planes_equal(got, oracle.resize(of, 2, sw, sh, src, dw, dh, oracle.FP32)[1], f"workspace {name}: {sw}x{sh}->{dw}x{dh}")
    del mem


def test_lanczos_weight_tables_from_concurrent_threads(capi, oracle):
    """Eight host threads, each with a stream of its own, resize the same few shapes (and some of their own) at the same time through
    vpf_resize_batch: the table cache is shared by every caller of the library in the process, and every thread's first use of a shape
    must queue that shape's build on ITS stream whoever allocated the entry.  Every frame equals the oracle (the ctypes calls release the
    GIL, so the launches really interleave)."""
    import threading

    shapes = [("RGB", 1283, 211, 857, 140), ("NV12", 1280, 240, 854, 160), ("Y", 811, 97, 1622, 194), ("RGB", 640, 360, 427, 240)]
    srcs = {i: oracle.synth(getattr(oracle, sh[0]), sh[1], sh[2], 9500 + i) for i, sh in enumerate(shapes)}
    wants = {i: oracle.resize(getattr(oracle, sh[0]), 2, sh[1], sh[2], srcs[i], sh[3], sh[4], oracle.FP32)[1] for i, sh in enumerate(shapes)}
    errors, results = [], []
    start = threading.Barrier(8)

    def worker(tid):
        try:
            st = torch.cuda.Stream()
            ex = capi.make_exec(st.cuda_stream)
            mine = []
            with torch.cuda.stream(st):
                own = ("RGB", 500 + 16 * tid, 120, 333 + 8 * tid, 80)
                osrc = oracle.synth(oracle.RGB, own[1], own[2], 9600 + tid)
                staged = [(i, DevPlanes(srcs[i]), DevPlanes(oracle.alloc(getattr(oracle, sh[0]), sh[3], sh[4]))) for i, sh in enumerate(shapes)]
                oS, oD = DevPlanes(osrc), DevPlanes(oracle.alloc(oracle.RGB, own[3], own[4]))
                st.synchronize()
                start.wait(timeout=120)  # (a thread that failed earlier aborts the barrier: nobody waits forever)
                for rep in range(3):
                    for i, S, D in (staged if tid % 2 == 0 else staged[::-1]):
                        sh = shapes[i]
                        capi.resize_batch(ex, getattr(capi, sh[0]), 2, sh[1], sh[2], sh[3], sh[4], capi.make_batch([(S.desc(), D.desc())]))
                    capi.resize_batch(ex, capi.RGB, 2, own[1], own[2], own[3], own[4], capi.make_batch([(oS.desc(), oD.desc())]))
                st.synchronize()
            for i, S, D in staged:
                mine.append((f"thread {tid} shape {shapes[i]}", D, wants[i]))
            mine.append((f"thread {tid} own shape {own}", oD, oracle.resize(oracle.RGB, 2, own[1], own[2], osrc, own[3], own[4], oracle.FP32)[1]))
            results.extend(mine)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))
            start.abort()

    ts = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(8)]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    assert not any(t.is_alive() for t in ts), "a worker thread is stuck"
    assert not errors, errors
    torch.cuda.synchronize()
    assert len(results) == 8 * 5
    for what, D, want in results:
        got, intact = D.download()
        assert intact
        assert_planes_equal(got, want, what)


def test_lanczos_weight_table_arena_full(tmp_path):
    """Without a caller-owned workspace the per-shape weight tables live in a small arena of static device memory with least-recently-used
    eviction; a table larger than the arena is not kept at all and that plane runs with its weights evaluated inside the kernel.
    VPF_HIP_LANCZOS_TABLE_KB shrinks the arena so that both happen after a few shapes: in a child process (the knob is read once) a dozen
    shapes twice over — evictions behind events, some planes of one launch with a table and others without, three planes of equal shape
    sharing one entry —, all equal the oracle; with 0 KB nothing gets a table."""
    import subprocess
    import textwrap

    code = textwrap.dedent("""
        import os, sys
        import numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import oracle
        from videoprocessingframework_amd import capi
        from gpu_util import DevPlanes
        ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
        n = 0
        for rep in range(2):
            for i, (fmt, sw, sh, dw, dh) in enumerate([("RGB", 640, 360, 427, 240), ("NV12", 1280, 720, 854, 480), ("YUV420", 642, 362, 300, 170), ("Y", 997, 61, 333, 47),
                                                        ("RGB", 320, 180, 640, 360), ("RGB", 1919, 64, 1280, 43), ("NV12", 854, 480, 1280, 720), ("RGB", 500, 300, 333, 200),
                                                        ("YUV444", 100, 60, 150, 90), ("RGB", 1280, 200, 640, 100), ("RGB", 700, 400, 467, 267), ("Y", 2000, 100, 1000, 50)]):
                f, of = getattr(capi, fmt), getattr(oracle, fmt)
                src = oracle.synth(of, sw, sh, 8800 + i)
                S, D = DevPlanes(src), DevPlanes(oracle.alloc(of, dw, dh))
                capi.resize_batch(ex, f, 2, sw, sh, dw, dh, capi.make_batch([(S.desc(), D.desc())]))
                torch.cuda.synchronize()
                got, intact = D.download()
                want = oracle.resize(of, 2, sw, sh, src, dw, dh, oracle.FP32)[1]
                assert intact and all(np.array_equal(g, w) for g, w in zip(got, want)), (fmt, sw, sh, dw, dh, rep)
                n += 1
        print("ARENA-OK", n)
    """) % (ROOT, ROOT)
    for kb in ("300", "0", "40"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VPF_HIP_LANCZOS_TABLE_KB=kb), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ARENA-OK 24" in r.stdout, (kb, r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("seed", _FUZZ_SEEDS)
def test_fuzz_resize_batch(capi, capi_forms, oracle, seed):
    """random format (multi-plane formats exercise the one-launch-for-all-planes kernels, odd sizes give the chroma planes their own
    scale factors), filter, size pair (incl. exact 2x, odd integer factors, up-scales), frame count, alignment and kernel family: every
    frame of the batch equals the oracle"""
    rng = np.random.default_rng(13000 + seed)
    for _ in range(3):
        fmt = str(rng.choice(["RGB", "BGR", "Y", "NV12", "YUV420", "YUV444", "RGB_PLANAR", "YCBCR"]))
        interp = int(rng.integers(0, 3))
        kind = int(rng.integers(5))
        if kind == 0:      # exact 2x
            dw, dh = int(rng.integers(2, 200)) * 2, int(rng.integers(1, 30)) * 2
            sw, sh = 2 * dw, 2 * dh
        elif kind == 1:    # odd integer factor
            k = int(rng.choice([3, 5]))
            dw, dh = int(rng.integers(2, 120)), int(rng.integers(2, 25))
            sw, sh = k * dw, k * dh
        elif kind == 2:    # up-scale
            sw, sh = int(rng.integers(8, 200)), int(rng.integers(6, 40))
            dw, dh = int(sw * rng.uniform(1.0, 3.0)), int(sh * rng.uniform(1.0, 3.0))
        else:              # general down-scale
            sw, sh = int(rng.integers(32, 900)), int(rng.integers(10, 200))
            dw, dh = max(2, int(sw / rng.uniform(1.0, 6.5))), max(2, int(sh / rng.uniform(1.0, 6.5)))   # (2.2 .. 6: two-chunk windows / half tiles of the matrix-core Lanczos kernel)
        n = int(rng.choice([1, 2, 3, 5, 40, 130], p=[0.3, 0.25, 0.2, 0.15, 0.05, 0.05]))  # 40: one dispatch through the 128-frame table; 130: 128 + 2
        if n >= 40 and sw * sh > 24000:  # (many frames: small ones)
            k = (sw * sh / 24000.0) ** 0.5
            sw, sh, dw, dh = max(2, int(sw / k)) & ~1, max(2, int(sh / k)) & ~1, max(2, int(dw / k)) & ~1, max(2, int(dh / k)) & ~1
        align, variant = int(rng.choice([256, 256, 16, 4, 1])), int(rng.choice([0, 0, 0, 40, 43, 9]))
        of = getattr(oracle, fmt)
        nsrc = min(n, 3)
        srcs = [oracle.synth(of, sw, sh, int(rng.integers(1 << 30))) for _ in range(nsrc)]
        S = [DevPlanes(srcs[i % nsrc], align) for i in range(n)]
        D = [DevPlanes(oracle.alloc(of, dw, dh), align) for _ in range(n)]
        band = int(rng.choice([0, 1, 2, 4, 8, 16, 0x104, 0x204, 0x304]))  # rows per wave of the row-pair kernels (small batches would never leave 1 by policy); 4 | nb << 8: the march form
        march = int(rng.choice([0, 1, (4 << 8) | 1, (8 << 8) | 1, (8 << 8) | 2, 3, 64, 0x20000, 0x20000 | 2, 0x20000 | 5, 0x40000, 0x80000, 0x80000 | (8 << 8) | 2]))  # 0x20000: the two-role form (pass 1 / pass 2 on different waves; lab build of the library); 0x40000: small single frames too; 0x80000: no ring of two
        if march != 1 and not (march & 0x20000) and rng.integers(3) == 0:
            march |= 0x10000                                                                # ... with its weights evaluated in the kernel, not loaded from the shape's tables
        lib = capi_forms if march & 0x20000 else capi
        f = getattr(lib, fmt)
        prev = lib.set_tuning(lib.TUNE_NV12_RGB_VARIANT, variant)
        lib.set_tuning(lib.TUNE_RESIZE_BAND, band)
        assert lib.set_tuning(lib.TUNE_RESIZE_MFMA, march) >= 0
        try:
            lib.resize_batch(lib.make_exec(stream_handle()), f, interp, sw, sh, dw, dh, lib.make_batch([(s.desc(), d.desc()) for s, d in zip(S, D)]))
        finally:
            lib.set_tuning(lib.TUNE_NV12_RGB_VARIANT, prev)
            lib.set_tuning(lib.TUNE_RESIZE_BAND, 0)
            lib.set_tuning(lib.TUNE_RESIZE_MFMA, 0)
        torch.cuda.synchronize()
        wants = [oracle.resize(of, interp, sw, sh, p, dw, dh, oracle.FP32)[1] for p in srcs]
        exacts = [oracle.resize(of, interp, sw, sh, p, dw, dh, oracle.EXACT)[1] for p in srcs] if interp != capi.INTERP_NEAREST else None  # (nearest: a coordinate tie moves a whole pixel, see test_fuzz_resize_and_fused)
        for i in range(n):
            got, intact = D[i].download()
            assert intact
            assert_planes_equal(got, wants[i % nsrc], f"fuzz resize_batch {fmt} interp {interp} {sw}x{sh}->{dw}x{dh} n{n} a{align} v{variant} band{band} mfma{march:#x} frame {i}")
            if exacts is not None:
                assert max(int(np.abs(g.astype(int) - e.astype(int)).max()) for g, e in zip(got, exacts[i % nsrc])) <= 1, \
                    f"fuzz resize_batch {fmt} interp {interp} {sw}x{sh}->{dw}x{dh} frame {i}: HIP vs EXACT > 1 LSB"
