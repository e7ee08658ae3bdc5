"""The oracle against implementations its authors did not write: the chain HIP == FP32 mode, FP32 within 1 LSB of EXACT,
EXACT == a `fractions` restatement ends in two files by the same hand.  Where PyTorch and PIL define the SAME operation, EXACT must equal
them: bilinear resize = torch.nn.functional.interpolate(bilinear, align_corners=False) in float64; remap = F.grid_sample(bilinear,
padding_mode="border"); Lanczos-3 on up-scales = PIL's LANCZOS (interior, mid-range input: PIL rounds and clamps its horizontal pass to
8 bits).  Differences of one LSB are exact ties (x.5) rounded differently, a fraction of a percent of the bytes.  CPU only.

What this does NOT pin: NPP itself (closed; parity stays "unpinned") and Lanczos when minifying, where the textbook six-tap
filter and PIL's widened kernel are different filters (assumption A10, tests/test_oracle_assumptions.py)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
F = torch.nn.functional


def _rhu(t):
    return np.clip(np.floor(t + 0.5), 0, 255).astype(np.uint8)


SHAPES = [((64, 48), (40, 30)), ((37, 29), (91, 55)), ((320, 180), (200, 120)), ((200, 120), (133, 80)), ((1920, 1080), (1280, 720))]


@pytest.mark.parametrize("sizes", SHAPES)
@pytest.mark.parametrize("fmt", ["Y", "RGB"])
def test_bilinear_resize_equals_torch_interpolate(oracle, sizes, fmt):
    """reference: nppiResize_8u_C3R / _C1R with NPPI_INTER_LINEAR — the oracle's EXACT bilinear vs torch (pixel-centre coordinates, edge clamp)"""
    o = oracle
    (sw, sh), (dw, dh) = sizes
    if sw * sh > 10 ** 6 and fmt == "RGB":
        pytest.skip("the 1080p case runs on one channel")
    ch = 3 if fmt == "RGB" else 1
    src = o.synth(getattr(o, fmt), sw, sh, 21)
    st, got = o.resize(getattr(o, fmt), o.LINEAR, sw, sh, src, dw, dh, o.EXACT)
    assert st == 0
    t = torch.from_numpy(src[0].reshape(sh, sw, ch).astype(np.float64)).permute(2, 0, 1)[None]
    want = _rhu(F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()).reshape(dh, dw * ch)
    d = np.abs(got[0].astype(int) - want.astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.002, (d.max(), (d != 0).mean())
    # ... and the kernels' arithmetic (FP32 mode) is within one LSB of the third party too
    d = np.abs(o.resize(getattr(o, fmt), o.LINEAR, sw, sh, src, dw, dh, o.FP32)[1][0].astype(int) - want.astype(int))
    assert d.max() <= 1


@pytest.mark.parametrize("sizes", [((64, 48), (40, 30)), ((123, 77), (200, 90)), ((320, 180), (320, 180))])
def test_remap_equals_torch_grid_sample(oracle, sizes):
    """reference: nppiRemap_8u_C3R with NPPI_INTER_LINEAR (Tasks.cpp:1590) — the oracle's EXACT remap vs grid_sample on in-range coordinates
    (out-of-range ones leave the destination untouched in the oracle [A9]; grid_sample has no such notion)"""
    o = oracle
    (sw, sh), (dw, dh) = sizes
    rng = np.random.default_rng(8)
    src = o.synth(o.RGB, sw, sh, 22)
    xm = rng.uniform(0, sw - 1, (dh, dw)).astype(np.float32)
    ym = rng.uniform(0, sh - 1, (dh, dw)).astype(np.float32)
    xm[0, :4] = [0.0, sw - 1, 0.5, sw - 1.5]
    ym[0, :4] = [0.0, sh - 1, sh - 1, 0.25]
    st, got = o.remap(o.RGB, sw, sh, src, xm, ym, o.EXACT)
    assert st == 0
    t = torch.from_numpy(src[0].reshape(sh, sw, 3).astype(np.float64)).permute(2, 0, 1)[None]
    grid = torch.stack([torch.from_numpy(xm.astype(np.float64)) * (2.0 / (sw - 1)) - 1.0, torch.from_numpy(ym.astype(np.float64)) * (2.0 / (sh - 1)) - 1.0], -1)[None]
    want = _rhu(F.grid_sample(t, grid, mode="bilinear", padding_mode="border", align_corners=True)[0].permute(1, 2, 0).numpy()).reshape(dh, dw * 3)
    d = np.abs(got[0].astype(int) - want.astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.002, (d.max(), (d != 0).mean())
    d = np.abs(o.remap(o.RGB, sw, sh, src, xm, ym, o.FP32)[1][0].astype(int) - want.astype(int))
    assert d.max() <= 1


@pytest.mark.parametrize("sizes", [((64, 48), (160, 120)), ((100, 60), (150, 90)), ((37, 29), (91, 55)), ((80, 80), (80, 80))])
def test_lanczos3_up_scale_equals_pil(oracle, sizes):
    """reference: NPPI_INTER_LANCZOS (Tasks.cpp:1190) — on up-scales every reading of "Lanczos-3" is the same six taps; PIL's LANCZOS is one.
    Interior pixels only (PIL drops taps beyond the edge and renormalises, the oracle clamps them onto the edge sample), mid-range input
    (PIL stores its horizontal pass as clamped 8-bit values)."""
    Image = pytest.importorskip("PIL.Image")
    o = oracle
    (sw, sh), (dw, dh) = sizes
    rng = np.random.default_rng(9)
    for ch, fmt, mode in ((1, o.Y, "L"), (3, o.RGB, "RGB")):
        src = [np.ascontiguousarray(rng.integers(64, 192, (sh, sw * ch), dtype=np.uint8))]
        pil = np.asarray(Image.fromarray(src[0].reshape(sh, sw, ch).squeeze(), mode).resize((dw, dh), Image.LANCZOS)).reshape(dh, dw * ch).astype(int)
        m = int(np.ceil(3 * max(dw / sw, dh / sh))) + 1
        inner = (slice(m, dh - m), slice(m * ch, (dw - m) * ch))
        for md in (o.EXACT, o.FP32):   # the specification, and the integer definition the kernels implement
            got = o.resize(fmt, o.LANCZOS3, sw, sh, src, dw, dh, md)[1][0].astype(int)
            d = np.abs(got - pil)[inner]
            assert d.max() <= 1, (ch, md, d.max())
            assert (d != 0).mean() < 0.3   # PIL rounds its horizontal pass to 8 bits: +-0.5 LSB of noise into the vertical pass moves a fifth of the bytes by one
        # the same structure as PIL — horizontal pass rounded to 8 bits, then the vertical pass (a resize that keeps one axis is the identity
        # on that axis) — removes that noise: what is left are ties and PIL's Q22 weights
        st, hpass = o.resize(fmt, o.LANCZOS3, sw, sh, src, dw, sh, o.EXACT)
        two = o.resize(fmt, o.LANCZOS3, dw, sh, hpass, dw, dh, o.EXACT)[1][0].astype(int)
        d = np.abs(two - pil)[inner]
        assert st == 0 and d.max() <= 1 and (d != 0).mean() < 0.01, (ch, d.max(), (d != 0).mean())
