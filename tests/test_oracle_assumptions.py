"""The oracle's assumption switches (A2 chroma up-sampling, A6 chroma decimation, A8 resize coordinates; DESIGN.md §2: A10
Lanczos support when minifying).
Each alternative is checked against a separate numpy/float64 restatement written here from its definition, so that when real
NPP output arrives (tests/test_reference_fixtures.py) flipping a switch is known to do what its name says."""
import numpy as np
import pytest


def _rhu(v):
    return np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8)


M709 = dict(cy=1.164384, off=16, rv=1.792741, gu=-0.213249, gv=-0.532909, bu=2.112402)


def _up(plane, siting, w, h):
    """bilinear 4:2:0 -> full resolution, float64; siting 1 = centred, 2 = left (co-sited horizontally, centred vertically)"""
    ch, cw = plane.shape
    p = plane.astype(np.float64)
    y = np.arange(h)
    cy = (y - 0.5) / 2.0
    x = np.arange(w)
    cx = (x - 0.5) / 2.0 if siting == 1 else x / 2.0
    y0, x0 = np.floor(cy).astype(int), np.floor(cx).astype(int)
    fy, fx = cy - y0, cx - x0
    c = lambda i, n: np.clip(i, 0, n - 1)
    top = p[c(y0, ch)][:, c(x0, cw)] * (1 - fx) + p[c(y0, ch)][:, c(x0 + 1, cw)] * fx
    bot = p[c(y0 + 1, ch)][:, c(x0, cw)] * (1 - fx) + p[c(y0 + 1, ch)][:, c(x0 + 1, cw)] * fx
    return top * (1 - fy)[:, None] + bot * fy[:, None]


@pytest.mark.parametrize("siting", [1, 2])
@pytest.mark.parametrize("src_fmt", ["NV12", "YUV420"])
def test_a2_interpolated_chroma(oracle, siting, src_fmt):
    o = oracle
    w, h = 38, 26
    fmt = getattr(o, src_fmt)
    src = o.synth(fmt, w, h, 31)
    if src_fmt == "NV12":
        U, V = src[1][:, 0::2], src[1][:, 1::2]
    else:
        U, V = src[1], src[2]
    yy = M709["cy"] * (src[0].astype(np.float64) - M709["off"])
    uu, vv = _up(U, siting, w, h) - 128, _up(V, siting, w, h) - 128
    want = np.stack([_rhu(yy + M709["rv"] * vv), _rhu(yy + M709["gu"] * uu + M709["gv"] * vv), _rhu(yy + M709["bu"] * uu)], -1).reshape(h, 3 * w)
    with o.assume(o.A2_CHROMA_UPSAMPLE, siting):
        st, got = o.convert(fmt, o.RGB, o.BT_709, o.MPEG, w, h, src, o.EXACT)
        assert st == 0
        # float64 vs exact rationals: identical except ties at exactly .5
        d = np.abs(got[0].astype(int) - want.astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 0.002
        assert o.convert(fmt, o.RGB, o.BT_709, o.MPEG, w, h, src, o.FP32)[0] == 1   # FP32 = the kernels = default convention only
    st, rep = o.convert(fmt, o.RGB, o.BT_709, o.MPEG, w, h, src, o.EXACT)
    assert st == 0 and (rep[0] != got[0]).mean() > 0.3       # the switch really changes the picture on random chroma
    assert o.lib().vpfo_get_assumption(o.A2_CHROMA_UPSAMPLE) == 0  # restored


def test_a2_is_invisible_on_flat_chroma_and_444(oracle):
    o = oracle
    w, h = 32, 16
    src = o.synth(o.NV12, w, h, 5)
    src[1][...] = np.array([90, 200], np.uint8)[None, :].repeat(w // 2, 0).reshape(1, w).repeat(h // 2, 0)
    base = o.convert(o.NV12, o.RGB, o.BT_601, o.JPEG, w, h, src, o.EXACT)[1][0]
    s444 = o.synth(o.YUV444, w, h, 6)
    b444 = o.convert(o.YUV444, o.RGB, o.BT_601, o.JPEG, w, h, s444, o.EXACT)[1][0]
    for siting in (1, 2):
        with o.assume(o.A2_CHROMA_UPSAMPLE, siting):
            assert np.array_equal(o.convert(o.NV12, o.RGB, o.BT_601, o.JPEG, w, h, src, o.EXACT)[1][0], base)
            assert np.array_equal(o.convert(o.YUV444, o.RGB, o.BT_601, o.JPEG, w, h, s444, o.EXACT)[1][0], b444)


def test_a6_top_left_decimation(oracle):
    o = oracle
    w, h = 30, 18
    src = o.synth(o.RGB, w, h, 77)
    st, full = o.convert(o.RGB, o.YUV444, o.BT_601, o.MPEG, w, h, src, o.EXACT)
    assert st == 0
    with o.assume(o.A6_CHROMA_DECIMATE, 1):
        st, got = o.convert(o.RGB, o.YUV420, o.BT_601, o.MPEG, w, h, src, o.EXACT)
        assert st == 0
        assert np.array_equal(got[0], full[0])
        assert np.array_equal(got[1], full[1][0::2, 0::2]) and np.array_equal(got[2], full[2][0::2, 0::2])
        assert o.convert(o.RGB, o.YUV420, o.BT_601, o.MPEG, w, h, src, o.FP32)[0] == 1
    st, mean = o.convert(o.RGB, o.YUV420, o.BT_601, o.MPEG, w, h, src, o.EXACT)
    assert st == 0 and (mean[1] != got[1]).mean() > 0.5


def _bilinear(src, dw, dh, coord):
    sh, sw = src.shape
    p = src.astype(np.float64)
    sy = np.clip(coord(np.arange(dh), sh, dh), 0, sh - 1)
    sx = np.clip(coord(np.arange(dw), sw, dw), 0, sw - 1)
    y0, x0 = np.floor(sy).astype(int), np.floor(sx).astype(int)
    y1, x1 = np.minimum(y0 + 1, sh - 1), np.minimum(x0 + 1, sw - 1)
    fy, fx = (sy - y0)[:, None], (sx - x0)[None, :]
    top = p[y0][:, x0] + fx * (p[y0][:, x1] - p[y0][:, x0])
    bot = p[y1][:, x0] + fx * (p[y1][:, x1] - p[y1][:, x0])
    return _rhu(top + fy * (bot - top))


COORDS = {
    0: lambda d, S, D: (d + 0.5) * (S / D) - 0.5,
    1: lambda d, S, D: d * (S / D),
    2: lambda d, S, D: d * ((S - 1) / (D - 1)) if D > 1 else d * 0.0,
}


@pytest.mark.parametrize("conv", [0, 1, 2])
@pytest.mark.parametrize("sizes", [((40, 28), (17, 11)), ((21, 13), (50, 31)), ((64, 32), (32, 16))])
def test_a8_resize_coordinate_conventions(oracle, conv, sizes):
    o = oracle
    (sw, sh), (dw, dh) = sizes
    src = o.synth(o.Y, sw, sh, 3)
    want = _bilinear(src[0], dw, dh, COORDS[conv])
    with o.assume(o.A8_RESIZE_COORDS, conv):
        st, got = o.resize(o.Y, o.LINEAR, sw, sh, src, dw, dh, o.EXACT)
        assert st == 0
        d = np.abs(got[0].astype(int) - want.astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 0.01
        assert o.resize(o.Y, o.LINEAR, sw, sh, src, dw, dh, o.FP32)[0] == (1 if conv else 0)
        st, lz = o.resize(o.Y, o.LANCZOS3, sw, sh, src, dw, dh, o.EXACT)       # the switch reaches the Lanczos taps too
        assert st == 0
    if conv == 2:  # corners aligned: the four corner pixels are copied
        assert got[0][0, 0] == src[0][0, 0] and got[0][-1, -1] == src[0][-1, -1] and lz[0][0, 0] == src[0][0, 0]


def _lanczos_wide(src, dw, dh):
    """A10 = 1 written from its definition in float64 numpy: per axis fs = max(1, S/D), every source sample i with |i - s| <= 3 fs weighted
    L((i - s) / fs), normalised, indices clamped to the picture; horizontal pass first"""
    def axis(p, S, D):  # p: (S, n) -> (D, n)
        fs = max(1.0, S / D)
        out = np.zeros((D, p.shape[1]))
        for d in range(D):
            s = (d + 0.5) * (S / D) - 0.5
            i = np.arange(int(np.ceil(s - 3 * fs)), int(np.floor(s + 3 * fs)) + 1)
            x = (i - s) / fs
            w = np.where(np.abs(x) >= 3, 0.0, np.sinc(x) * np.sinc(x / 3))
            out[d] = (w / w.sum()) @ p[np.clip(i, 0, S - 1)]
        return out
    sh, sw = src.shape
    return _rhu(axis(axis(src.astype(np.float64).T, sw, dw).T, sh, dh))


@pytest.mark.parametrize("sizes", [((200, 120), (133, 80)), ((90, 70), (30, 35)), ((64, 48), (11, 7)), ((40, 30), (40, 30)), ((31, 23), (80, 61))])
def test_a10_lanczos_support_scaled_when_minifying(oracle, sizes):
    """A10: the reference asks NPP for NPPI_INTER_LANCZOS (Tasks.cpp:1190,1248) and NPP does not say whether the kernel widens when
    minifying.  0 = six taps whatever the scale (what the HIP kernels implement); 1 = support scaled by max(1, S/D) (PIL / swscale)."""
    o = oracle
    (sw, sh), (dw, dh) = sizes
    src = o.synth(o.Y, sw, sh, 12)
    st, six = o.resize(o.Y, o.LANCZOS3, sw, sh, src, dw, dh, o.EXACT)
    assert st == 0
    with o.assume(o.A10_LANCZOS_MINIFY, 1):
        st, wide = o.resize(o.Y, o.LANCZOS3, sw, sh, src, dw, dh, o.EXACT)
        assert st == 0
        assert o.resize(o.Y, o.LANCZOS3, sw, sh, src, dw, dh, o.FP32)[0] == 1   # FP32 = the kernels = six taps only
        assert o.resize(o.Y, o.LINEAR, sw, sh, src, dw, dh, o.FP32)[0] == 0     # ... the other filters are not touched by the switch
        d = np.abs(wide[0].astype(int) - _lanczos_wide(src[0], dw, dh).astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 0.01                          # the switch does what its definition says (ties only)
        flat = o.alloc(o.Y, sw, sh, fill=173)
        assert (o.resize(o.Y, o.LANCZOS3, sw, sh, flat, dw, dh, o.EXACT)[1][0] == 173).all()   # a flat picture stays flat
        rgb = o.synth(o.RGB, sw, sh, 13)
        st, w3 = o.resize(o.RGB, o.LANCZOS3, sw, sh, rgb, dw, dh, o.EXACT)      # packed: each channel is the 1-channel filter
        assert st == 0 and np.array_equal(w3[0][:, 1::3], o.resize(o.Y, o.LANCZOS3, sw, sh, [np.ascontiguousarray(rgb[0][:, 1::3])], dw, dh, o.EXACT)[1][0])
    assert o.lib().vpfo_get_assumption(o.A10_LANCZOS_MINIFY) == 0
    if sw <= dw and sh <= dh:   # identity and up-scales: the two readings are the same filter
        assert np.array_equal(wide[0], six[0])
        if (sw, sh) == (dw, dh):
            assert np.array_equal(wide[0], src[0])
    else:                       # minifying noise: they are NOT — this is what the pin kit's classifier decides on NPP's output
        d = np.abs(wide[0].astype(int) - six[0].astype(int))
        assert d.mean() > 3 and d.max() > 15


def test_a10_equals_pil_lanczos_on_a_mid_range_down_scale(oracle):
    """... and reading 1 IS what PIL calls LANCZOS: interior pixels of a 1.5 x down-scale agree to <= 2 LSB (PIL rounds the horizontal pass
    to 8 bits and uses Q22 weights; mid-range input keeps its clamped intermediate out of play), while reading 0 is ~8 LSB away on average"""
    Image = pytest.importorskip("PIL.Image")
    o = oracle
    sw, sh, dw, dh = 200, 120, 133, 80
    rng = np.random.default_rng(5)
    src = [np.ascontiguousarray(rng.integers(64, 192, (sh, sw), dtype=np.uint8))]
    pil = np.asarray(Image.fromarray(src[0], "L").resize((dw, dh), Image.LANCZOS)).astype(int)
    with o.assume(o.A10_LANCZOS_MINIFY, 1):
        wide = o.resize(o.Y, o.LANCZOS3, sw, sh, src, dw, dh, o.EXACT)[1][0].astype(int)
    six = o.resize(o.Y, o.LANCZOS3, sw, sh, src, dw, dh, o.EXACT)[1][0].astype(int)
    inner = (slice(6, -6), slice(6, -6))
    assert np.abs(wide - pil)[inner].max() <= 2 and np.abs(wide - pil)[inner].mean() < 0.5
    assert np.abs(six - pil)[inner].mean() > 3


def test_switch_rejects_unknown_keys_and_values(oracle):
    L = oracle.lib()
    assert L.vpfo_set_assumption(3, 0) == -1 and L.vpfo_set_assumption(oracle.A6_CHROMA_DECIMATE, 2) == -1
    assert L.vpfo_set_assumption(oracle.A8_RESIZE_COORDS, 3) == -1 and L.vpfo_get_assumption(oracle.A8_RESIZE_COORDS) == 0
    assert L.vpfo_set_assumption(oracle.A10_LANCZOS_MINIFY, 2) == -1 and L.vpfo_get_assumption(oracle.A10_LANCZOS_MINIFY) == 0
