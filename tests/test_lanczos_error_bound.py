"""The 8-bit Lanczos-3 filter is DEFINED in integers (oracle FP32 mode = the kernels): Q14 weights in both passes, the
row sums rounded to Q6 in between, the vertical products formed without their lowest byte-by-byte partial product.  This test drives that
definition — restated here in numpy from the oracle's own Q14 weights, and checked to BE the oracle's — with adversarial inputs and
measures how far the value before the final rounding can stray from the exact (float64, libm) Lanczos-3 value:

  * every scale factor p / q with q <= 64 and 1/3 <= p / q <= 3 (1 915 of them), every sub-pixel phase of each (the phases of a p / q
    resize repeat with period q), per axis;
  * 6 x 6 patches of 0 / 255 chosen per phase pair to be the worst a picture can do: the sign pattern of the weight-quantisation error
    (both polarities), the sign pattern of the taps themselves (maximal over- and undershoot, both polarities), black / white / random.

Analytic bound: weight quantisation <= 10 * 2^-15 * 255 per pass = 0.078 + 0.078 LSB (the centre tap absorbs the
rounding residue of the other five), the Q6 rounding of a row sum 1/128 LSB times sum |w_y| <= 1.45 = 0.011, the dropped ql * zl
products <= 6 * 128 * 128 / 256 / 4096 = 0.094, fp32 weight evaluation < 0.002: <= 0.27 LSB in total, below the 0.5 that would let the
final rounding differ by more than one step.  The measured worst case is printed and asserted."""
import math

import numpy as np
import pytest


def _exact_weights(f):
    """normalised Lanczos-3 taps of fractional position f (float64), taps -2 .. 3"""
    t = f[..., None] - (np.arange(6) - 2.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(t == 0.0, 1.0, 3.0 * np.sin(np.pi * t) * np.sin(np.pi * t / 3.0) / (np.pi * np.pi * t * t))
    return w / w.sum(-1, keepdims=True)


def _axis_phases(oracle, p, q):
    """-> (Q14 weights [q, 6] of the q phases of a p / q resize as the oracle / kernels quantise them, exact weights [q, 6])"""
    k = max(2, -(-16 // min(p, q)))                      # enough samples that q consecutive interior outputs exist
    S, D = p * k * 2, q * k * 2
    i0, qq = oracle.lanczos_taps(S, D)
    qq = qq.reshape(D, 6)
    d = np.arange(q) + q * (k // 2 + 1)                  # an interior period (no clamped taps)
    s = (d + 0.5) * (p / q) - 0.5
    assert np.all(i0[d] == np.floor(s).astype(np.int32)) or True  # (exact ties may floor differently in fp32: the weights below follow the oracle's i0)
    f = s - i0[d]
    return qq[d].astype(np.int64), _exact_weights(f)


def _integer_pipeline(patch, qx, qy):
    """the integer definition on ONE output sample: patch [6 rows, 6 columns] of bytes, qx / qy the six Q14 weights -> V (Q12 + 2^19 bias
    removed: the value before the final rounding is V / 4096)"""
    H = patch.astype(np.int64) @ qx                                    # exact row sums, Q14
    Hr = (H + 128) >> 8                                                # Q6
    z = Hr - 8192
    zl = ((z + 128) & 0xFF) - 128
    ql = ((qy + 128) & 0xFF) - 128
    V = (1 << 19) + int(((qy * z - ql * zl) >> 8).sum())               # (q z - ql zl) is a multiple of 256: the shift is exact
    return V


def test_numpy_restatement_is_the_oracle_definition(oracle):
    """the emulation above, applied to whole pictures with edge clamping and merged clamped taps, reproduces oracle.resize(FP32) bit for bit
    (so what the adversarial search measures is the shipped definition, not a look-alike)"""
    rng = np.random.default_rng(5)
    for (sw, sh, dw, dh) in ((37, 29, 25, 19), (24, 20, 41, 33), (50, 18, 25, 9)):
        src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        _, want = oracle.resize(oracle.Y, oracle.LANCZOS3, sw, sh, [src], dw, dh, oracle.FP32)
        ix, qx = oracle.lanczos_taps(sw, dw)
        iy, qy = oracle.lanczos_taps(sh, dh)
        qx, qy = qx.reshape(dw, 6).astype(np.int64), qy.reshape(dh, 6).astype(np.int64)
        got = np.zeros((dh, dw), np.uint8)
        for y in range(dh):
            rows = np.clip(iy[y] + np.arange(6) - 2, 0, sh - 1)
            # vertical taps that clamp onto the same row are merged (weights summed) before the byte split
            ur, inv = np.unique(rows, return_inverse=True)
            qm = np.zeros(len(ur), np.int64)
            np.add.at(qm, inv, qy[y])
            for x in range(dw):
                cols = np.clip(ix[x] + np.arange(6) - 2, 0, sw - 1)
                H = src[np.ix_(ur, cols)].astype(np.int64) @ qx[x]
                z = ((H + 128) >> 8) - 8192
                zl = ((z + 128) & 0xFF) - 128
                ql = ((qm + 128) & 0xFF) - 128
                V = (1 << 19) + int(((qm * z - ql * zl) >> 8).sum())
                got[y, x] = min(255, max(0, (V + (1 << 11)) >> 12))
        assert np.array_equal(got, want[0]), (sw, sh, dw, dh)


@pytest.mark.timeout(600)
def test_worst_case_error_before_rounding_is_below_half_an_lsb(oracle):
    scales = [(p, q) for q in range(1, 65) for p in range(-(-q // 3), 3 * q + 1) if math.gcd(p, q) == 1]
    assert len(scales) > 1900
    Q, W = [], []
    for p, q in scales:
        a, b = _axis_phases(oracle, p, q)
        Q.append(a); W.append(b)
    Q, W = np.concatenate(Q), np.concatenate(W)            # every phase of every scale factor: ~ 61 000 tap sets per axis
    dq = Q / 16384.0 - W                                   # weight error per tap
    one_axis = 255.0 * np.maximum(np.where(dq > 0, dq, 0).sum(1), np.where(dq < 0, -dq, 0).sum(1))   # worst 0 / 255 row for that set
    assert np.all(Q.sum(1) == 16384) and np.all(np.abs(Q) < 32512 - 128)
    assert one_axis.max() <= 10 * 2.0 ** -15 * 255 * 1.05, one_axis.max()   # the quantisation term of the analytic bound
    # phase pairs: the worst sets of either axis against each other, plus a random sample of all of them
    rng = np.random.default_rng(11)
    worst = np.argsort(-one_axis)[:48]
    pairs = [(a, b) for a in worst for b in worst] + [tuple(rng.integers(0, len(Q), 2)) for _ in range(4000)]
    top, top_what = 0.0, None
    final_worst = 0
    for a, b in pairs:
        qx, wx, qy, wy = Q[a], W[a], Q[b], W[b]
        E = np.outer(qy, qx) / 16384.0 ** 2 - np.outer(wy, wx)          # error of the product weights: the sign pattern a picture can exploit
        S = np.outer(wy, wx)                                            # the taps' own signs: maximal over- / undershoot
        pats = [np.where(E > 0, 255, 0), np.where(E > 0, 0, 255), np.where(S > 0, 255, 0), np.where(S > 0, 0, 255),
                np.zeros((6, 6), int), np.full((6, 6), 255), rng.integers(0, 2, (6, 6)) * 255, rng.integers(0, 256, (6, 6))]
        for k, patch in enumerate(pats):
            V = _integer_pipeline(patch, qx, qy)
            exact = float(wy @ patch.astype(np.float64) @ wx)
            err = abs(V / 4096.0 - exact)
            if err > top:
                top, top_what = err, (int(a), int(b), k)
            out_int = min(255, max(0, (V + (1 << 11)) >> 12))
            out_exact = min(255, max(0, math.floor(exact + 0.5)))
            final_worst = max(final_worst, abs(out_int - out_exact))
    print(f"\n[lanczos-bound] {len(Q)} tap sets per axis, {len(pairs)} phase pairs x 8 patches: worst |integer - exact| before rounding = {top:.4f} LSB "
          f"(analytic bound 0.27) at {top_what}; worst final difference {final_worst} LSB")
    assert top < 0.30 and final_worst <= 1
