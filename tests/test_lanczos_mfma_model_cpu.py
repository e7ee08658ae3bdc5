"""The bookkeeping of the MFMA Lanczos kernel (tests/lanczos_mfma_model.py: windows, K-slot layout, signed-byte splits and their
constants, ring slots, tile triggers) against the oracle's definition of 8-bit Lanczos-3, on the CPU."""
import numpy as np
import pytest

from lanczos_mfma_model import Model


def _run(oracle, ch, sw, sh, dw, dh, nt, band, seed=3, kc=1, rt=16, up2=False):
    fmt = {1: oracle.Y, 3: oracle.RGB}.get(ch)
    rng = np.random.default_rng(seed)
    if ch == 2:   # a 2-channel plane = the chroma plane of an NV12 picture twice as large
        src = [rng.integers(0, 256, (2 * sh, 2 * sw), dtype=np.uint8), rng.integers(0, 256, (sh, 2 * sw), dtype=np.uint8)]
        _, want = oracle.resize(oracle.NV12, oracle.LANCZOS3, 2 * sw, 2 * sh, src, 2 * dw, 2 * dh, oracle.FP32)
        plane, want = src[1], want[1]
    else:
        src = [rng.integers(0, 256, (sh, sw * ch), dtype=np.uint8)]
        _, want = oracle.resize(fmt, oracle.LANCZOS3, sw, sh, src, dw, dh, oracle.FP32)
        plane, want = src[0], want[0]
    m = Model(ch, sw, sh, dw, dh, oracle.lanczos_taps(sw, dw), oracle.lanczos_taps(sh, dh), nt=nt, band_rows=band, kc=kc, rt=rt, up2=up2)
    got = m.run(plane)
    assert np.array_equal(got, want), f"ch{ch} {sw}x{sh}->{dw}x{dh} nt{nt} band{band}: {np.argwhere(got != want)[:5]}"
    return m


@pytest.mark.parametrize("ch", [1, 2, 3])
def test_model_equals_the_oracle(oracle, ch):
    for (sw, sh, dw, dh, nt, band) in ((96, 54, 64, 36, 4, 32), (64, 36, 96, 54, 8, 16), (128, 72, 64, 36, 2, 48), (50, 41, 50, 41, 4, 32),
                                       (37, 29, 53, 71, 4, 64), (7, 5, 40, 33, 8, 16), (120, 90, 57, 43, 8, 32), (40, 200, 40, 97, 4, 96),
                                       (3, 3, 9, 9, 4, 16), (1, 1, 5, 4, 4, 16), (200, 17, 95, 40, 8, 16)):
        _run(oracle, ch, sw, sh, dw, dh, nt, band)


@pytest.mark.parametrize("ch", [1, 2, 3])
def test_model_with_the_ring_of_two(oracle, ch):
    """up-scales whose destination tiles find their source rows in two consecutive source tiles (vpf_bound_lzm_rows_two): overlapping
    two-tile chunks in the register file, one K chunk in pass 2 — same bytes"""
    for (sw, sh, dw, dh, nt, band) in ((64, 36, 96, 54, 8, 16), (64, 54, 128, 108, 4, 48), (37, 29, 53, 71, 4, 64), (7, 5, 40, 33, 8, 16),
                                       (3, 3, 9, 9, 4, 16), (1, 1, 5, 4, 4, 16), (48, 90, 72, 135, 8, 144), (20, 100, 30, 300, 4, 112)):
        m = _run(oracle, ch, sw, sh, dw, dh, nt, band, up2=True)
        assert m.max_tile_span <= 1


@pytest.mark.parametrize("ch", [1, 2, 3])
def test_model_with_two_chunk_windows(oracle, ch):
    """horizontal factors whose taps do not fit a 64-B window (the model asserts k < 64 kc for every tap): the 128-B windows hold them, the
    second K chunk accumulating onto the first; 4-tile strips"""
    for (sw, sh, dw, dh, band) in ((300, 40, 65, 17, 16), (480, 54, 104, 26, 32), (333, 29, 111, 23, 16), (200, 20, 37, 11, 16)):
        m = _run(oracle, ch, sw, sh, dw, dh, 4, band, kc=2)
        assert 64 <= m.max_k < 128   # (these shapes need the second chunk)


@pytest.mark.parametrize("ch", [1, 2, 3])
def test_model_with_half_tiles(oracle, ch):
    """vertical factors at which 16 destination rows need more than the ring's four source tiles (the model asserts the span of every tile):
    8 rows per tile do not; with one- and two-chunk windows"""
    for (sw, sh, dw, dh, band, kc) in ((64, 200, 40, 47, 16, 1), (300, 160, 65, 33, 8, 2), (480, 300, 104, 55, 24, 2), (90, 131, 60, 23, 32, 1)):
        m = _run(oracle, ch, sw, sh, dw, dh, 4, band, kc=kc, rt=8)
        assert m.max_tile_span <= 3
    with pytest.raises(AssertionError):   # (the same shape with full tiles does not fit the ring)
        _run(oracle, ch, 64, 200, 40, 47, 4, 16)


@pytest.mark.parametrize("ch", [1, 2, 3])
def test_model_with_three_chunk_windows(oracle, ch):
    """horizontal factors up to ~10 (1080p -> 224 x 224): 192-B windows, 2-tile strips, half tiles"""
    for (sw, sh, dw, dh, band) in ((480, 135, 56, 28, 8), (700, 90, 77, 23, 16), (960, 64, 104, 17, 8)):
        m = _run(oracle, ch, sw, sh, dw, dh, 2, band, kc=3, rt=8)
        assert 128 <= m.max_k < 192


def test_model_flat_and_extremes(oracle):
    """0 / 255 pictures exercise the signed-byte offsets: every constant that is off shows up as a uniform error"""
    for val in (0, 255, 128, 127):
        src = np.full((40, 60 * 3), val, np.uint8)
        m = Model(3, 60, 40, 41, 27, oracle.lanczos_taps(60, 41), oracle.lanczos_taps(40, 27), nt=4, band_rows=16)
        assert (m.run(src) == val).all()
