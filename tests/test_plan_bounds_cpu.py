"""The LDS sizing bounds of the strip-staging resize kernels (csrc/vpf_plan_bounds.h — the header the launchers include) against the
kernels' exact fp32 tap arithmetic, on the CPU.  A bound one byte or one row short would be silent memory corruption on the device;
here every formula is compiled with gcc as it stands and checked over thousands of (source size, destination size, position) cases.

Tap arithmetic restated from k_bilinear_blend.h make_tap / k_resize.hip ltap_i0 (fma in fp32: the product of two floats is exact in
float64, the sum with -0.5 as well at these magnitudes, one rounding to float32 — what v_fma_f32 does)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


@pytest.fixture(scope="module")
def pb(tmp_path_factory):
    from conftest import native_test_build
    so = str(tmp_path_factory.mktemp("pb") / "libplanbounds.so")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-shared", "-fPIC", "-Wall", "-Werror", *native_test_build()[0], "-I" + os.path.join(ROOT, "videoprocessingframework_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "plan_bounds_capi.c"), "-o", so, "-lm"])
    L = C.CDLL(so)
    for name, args in (("pb_strip_bytes", [C.c_int] + [C.c_uint32] * 4), ("pb_band_slots", [C.c_int, C.c_float]), ("pb_band_rows_exact", [C.c_int, C.c_uint32, C.c_uint32]),
                       ("pb_fused_rowbytes", [C.c_float]), ("pb_fused_rowbytes4", [C.c_float]), ("pb_strip_bytes_px4", [C.c_uint32] * 4), ("pb_tile_rows", [C.c_uint32, C.c_float, C.c_int]),
                       ("pb_tile_rowq", [C.c_float, C.c_int, C.c_int, C.c_int]), ("pb_lzm_span", [C.c_int, C.c_uint32, C.c_uint32, C.c_int]), ("pb_lzm_span_win", [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]),
                       ("pb_lzm_pitch", [C.c_uint32]), ("pb_lzm_rows_ok", [C.c_uint32, C.c_uint32]), ("pb_lzm_rows_two", [C.c_uint32, C.c_uint32])):
        getattr(L, name).argtypes, getattr(L, name).restype = args, C.c_uint32
    L.pb_fused_rows_fit.argtypes, L.pb_fused_rows_fit.restype = [C.c_int, C.c_float, C.c_int], C.c_int
    return L


def _s(d, scale):
    """fma((float)d + 0.5f, scale, -0.5f) for an array of destination indices"""
    d = np.asarray(d)
    return ((d.astype(F) + F(0.5)).astype(np.float64) * np.float64(scale) - 0.5).astype(F)


def lin_taps(d, S, D):
    """make_tap<LINEAR>: (i0, i1) of destination indices d"""
    scale = F(F(S) / F(D))
    s = np.minimum(np.maximum(_s(d, scale), F(0)), F(S - 1))
    i0 = s.astype(np.int64)
    return i0, np.minimum(i0 + 1, S - 1)


def lz_i0(d, S, D):
    """ltap_i0: floor of the unclamped source coordinate (the six taps are i0 - 2 .. i0 + 3)"""
    return np.floor(_s(d, F(F(S) / F(D)))).astype(np.int64)


def size_pairs(rng, n, lo=0.2, hi=4.0):
    out = [(1920, 1280), (1080, 720), (3840, 1920), (1280, 1920), (720, 1080), (1920, 3840), (1920, 416), (2160, 1080), (64, 1000), (7, 300), (1, 9)]
    for d in (255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048):   # chunk boundaries, exact ratios, one off
        for num, den in ((1, 1), (2, 1), (3, 2), (3, 1), (1, 2), (2, 3), (4, 3), (5, 1)):
            if lo <= num / den <= hi:
                out += [(max(1, d * num // den), d), (max(1, d * num // den + 1), d), (max(1, d * num // den - 1), d)]
    while len(out) < n:
        d = int(rng.integers(1, 3000))
        s = max(1, int(d * rng.uniform(lo, hi)))
        out.append((s, d))
    return out


def test_band_slots_cover_every_band(pb):
    """RowBandTask / convert_strip_task stage the contiguous rows [i0(first row of the band), i1(last row)] into `slots` strips"""
    rng = np.random.default_rng(1)
    for sh, dh in size_pairs(rng, 1500, 0.2, 2.0):
        scy = F(F(sh) / F(dh))
        y = np.arange(dh)
        i0, i1 = lin_taps(y, sh, dh)
        for r in (2, 4, 8, 16):
            ya = np.arange(0, dh, r)
            yb = np.minimum(ya + r - 1, dh - 1)
            need = int((i1[yb] - i0[ya] + 1).max())
            assert need <= pb.pb_band_slots(r, scy), (sh, dh, r, need)
            assert need == pb.pb_band_rows_exact(r, sh, dh), (sh, dh, r, need)  # the walk the launchers size the LDS rows with
            if pb.pb_fused_rows_fit(r if r <= 8 else 8, scy, 8):
                rr = r if r <= 8 else 8
                ya = np.arange(0, dh, rr)
                assert int((i1[np.minimum(ya + rr - 1, dh - 1)] - i0[ya] + 1).max()) <= 8, (sh, dh, rr)


@pytest.mark.parametrize("ch", [1, 2, 3])
def test_bilinear_strips_hold_the_span_and_the_tap_window(pb, ch):
    """RowPairTask / RowBandTask: strip = bytes [base, ch * (last + 1)) of the row, base = (ch * first) & ~15; packed RGB reads every tap pair
    as a 12-B window from the dword below (over-read of 6 bytes past the last tap)"""
    rng = np.random.default_rng(2)
    for sw, dw in size_pairs(rng, 1500, 0.05, 16.0):
        for cols in (256, 512):
            rb = pb.pb_strip_bytes(ch, sw, dw, 4096, cols)
            if not rb:
                continue
            xs = np.arange(0, dw, cols)
            xe = np.minimum(xs + cols - 1, dw - 1)
            first, last = lin_taps(xs, sw, dw)[0], lin_taps(xe, sw, dw)[1]
            base = (ch * first) & ~15
            span = ch * (last + 1) - base
            assert int(((span + 15) // 16 * 16).max()) <= rb, (sw, dw, cols)        # what the staging loop writes (whole 16-B units)
            x = np.arange(dw)
            i0 = lin_taps(x, sw, dw)[0]
            a = ch * i0 - base[x // cols]
            reach = (a & ~3) + 12 if ch == 3 else a + 2 * ch                           # window of three dwords / two byte taps
            assert int(reach.max()) <= rb, (sw, dw, cols)


def test_fused_strip_rows_hold_the_converted_window(pb):
    """convert_strip_task: a strip row holds packed RGB of source pixels [first & ~7, ...) in groups of 8 up to the last tap; taps are read as
    12-B windows"""
    rng = np.random.default_rng(4)
    for sw, dw in size_pairs(rng, 1500, 0.2, 3.0):
        if sw % 8:
            continue
        rowbytes = pb.pb_fused_rowbytes(F(F(sw) / F(dw)))
        xs = np.arange(0, dw, 256)
        xe = np.minimum(xs + 255, dw - 1)
        first, last = lin_taps(xs, sw, dw)[0], lin_taps(xe, sw, dw)[1]
        base_px = first & ~7
        conv_end = base_px + (last - base_px) // 8 * 8 + 8                             # conversion runs in groups of 8 pixels through `last`
        assert int((3 * (conv_end - base_px)).max()) <= rowbytes, (sw, dw)
        x = np.arange(dw)
        a = 3 * (lin_taps(x, sw, dw)[0] - base_px[x // 256])
        assert int(((a & ~3) + 12).max()) <= rowbytes, (sw, dw)


def test_band_px4_strips_hold_the_widened_span(pb):
    """RowBandTask<3, ..> since round 6: a strip row holds R G B x dwords of source pixels [first & ~3, last + 1] in whole units of four (Span4:
    12 source bytes -> 16 LDS bytes per lane); a pixel's taps are the dword at 4 (i0 - base_px) and the next one"""
    rng = np.random.default_rng(15)
    for sw, dw in size_pairs(rng, 1500, 0.2, 4.0):
        rb = pb.pb_strip_bytes_px4(sw, dw, 4096, 256)
        if rb == 0:
            continue
        assert rb % 16 == 0
        xs = np.arange(0, dw, 256)
        xe = np.minimum(xs + 255, dw - 1)
        first, last = lin_taps(xs, sw, dw)[0], lin_taps(xe, sw, dw)[1]
        base_px = first & ~3
        nu = (last + 2 - base_px + 3) // 4
        assert int((16 * nu).max()) <= rb, (sw, dw)                                     # what the staging loop writes
        x = np.arange(dw)
        a = 4 * (lin_taps(x, sw, dw)[0] - base_px[x // 256])
        assert int((a + 8).max()) <= 16 * int(nu.max()) and int((a + 8 - 16 * nu[x // 256]).max()) <= 0, (sw, dw)   # both tap dwords lie in staged units


def test_fused_px4_strip_rows_hold_the_converted_window(pb):
    """convert_strip_wg_task since round 6: a strip row holds R G B x dwords of source pixels [first & ~7, ...) in groups of 8 up to the last tap;
    a pixel's taps are the dword at 4 (i0 - base_px) and the NEXT dword (also where i1 == i0: weight 0, but the dword is read)"""
    rng = np.random.default_rng(14)
    for sw, dw in size_pairs(rng, 1500, 0.2, 3.0):
        if sw % 8:
            continue
        rowbytes = pb.pb_fused_rowbytes4(F(F(sw) / F(dw)))
        assert rowbytes % 16 == 0
        xs = np.arange(0, dw, 256)
        xe = np.minimum(xs + 255, dw - 1)
        first, last = lin_taps(xs, sw, dw)[0], lin_taps(xe, sw, dw)[1]
        base_px = first & ~7
        conv_end = base_px + (last - base_px) // 8 * 8 + 8                             # conversion runs in groups of 8 pixels through `last`
        assert int((4 * (conv_end - base_px)).max()) <= rowbytes, (sw, dw)
        x = np.arange(dw)
        a = 4 * (lin_taps(x, sw, dw)[0] - base_px[x // 256])
        assert int((a + 8).max()) <= rowbytes, (sw, dw)


@pytest.mark.parametrize("taps", [2, 6])
def test_tile_windows(pb, taps):
    """TileTask: source rows [R0, R1] of a tile of ty destination rows, and the 16-B units of a staged source row"""
    rng = np.random.default_rng(5)
    for s_, d_ in size_pairs(rng, 800, 0.1, 8.0):
        scale = F(F(s_) / F(d_))
        for ty in (4, 8, 16, 24, 32, 64):
            y0 = np.arange(0, d_, ty)
            yl = np.minimum(y0 + ty - 1, d_ - 1)
            if taps == 6:
                need = (lz_i0(yl, s_, d_) + 3) - (lz_i0(y0, s_, d_) - 2) + 1
            else:
                need = lin_taps(yl, s_, d_)[1] - lin_taps(y0, s_, d_)[0] + 1
            assert int(need.max()) <= pb.pb_tile_rows(ty, scale, taps), (s_, d_, ty)
        for ch in (1, 2, 3):
            rowq = pb.pb_tile_rowq(scale, taps, ch, 1)
            xf = np.arange(0, d_, 64)
            xl = np.minimum(xf + 63, d_ - 1)
            if taps == 6:
                fv, lv = lz_i0(xf, s_, d_) - 2, lz_i0(xl, s_, d_) + 3
                first, last = np.clip(fv, 0, s_ - 1), np.clip(lv, 0, s_ - 1)
                base = (ch * first) & ~15
                nq = (ch * (last + 1) - base + 15) // 16
                assert int((1 + nq).max()) <= rowq, (s_, d_, ch)                                   # pad unit + staged units
                assert int((16 + ch * (lv + 1) - base).max()) <= rowq * 16, (s_, d_, ch)           # replicated right margin
                x = np.arange(d_)
                off = 16 + ch * (lz_i0(x, s_, d_) - 2) - base[x // 64]
                assert int(off.min()) >= 0 and int(((off & ~3) + 4 * ((6 * ch + 3) // 4 + 1)).max()) <= rowq * 16, (s_, d_, ch)
            else:
                first, last = lin_taps(xf, s_, d_)[0], lin_taps(xl, s_, d_)[1]
                base = (ch * first) & ~15
                assert int(((ch * (last + 1) - base + 15) // 16).max()) <= rowq, (s_, d_, ch)


@pytest.mark.parametrize("ch", [1, 2, 3])
def test_lanczos_mfma_windows_ring_and_strip(pb, ch):
    """k_lanczos_mfma.hip: the launcher walks the tiles with the kernel's own fp32 coordinate arithmetic (vpf_bound_lzm_span /
    vpf_bound_lzm_rows_ok).  Against an independent numpy evaluation of every destination BYTE: a non-zero span means every tap of the 16
    destination bytes of every N-tile lies in the 64-B window that starts at the 16-B aligned byte below the tile's first tap (K = 64 of the
    MFMA) and every strip of nt tiles fits the span (the LDS pitch is >= span and 32 mod 64: conflict-free A-operand reads, checked below); a zero span means some tile really does not
    fit (the check is exact up to the last pixel's unused channels); rows_ok means a 16-row destination tile finds its source rows in four
    consecutive 16-row source tiles.  And the headline ratios — exactly 2.0 with three channels among them — pass."""
    rng = np.random.default_rng(77 + ch)
    n_ok = n_no = n_two = 0
    for (S, D) in size_pairs(rng, 500, 0.2, 3.2):
        dwb = D * ch
        b = np.arange(dwb)
        px, c = b // ch, b % ch
        i0 = lz_i0(np.arange(D), S, D)
        lo = np.clip(i0 - 2, 0, S - 1)[px] * ch + c      # lowest / highest source byte a destination byte touches
        hi = np.clip(i0 + 3, 0, S - 1)[px] * ch + c
        first_b = np.minimum((b // 16) * 16, dwb - 1)
        ws = (np.clip(i0 - 2, 0, S - 1)[first_b // ch] * ch) & ~15
        fits = bool((lo >= ws).all() and (hi - ws < 64).all())
        for nt in (4, 8):
            span = pb.pb_lzm_span(ch, S, D, nt)
            if span:
                assert fits, (S, D, int((hi - ws).max()))
                pitch = pb.pb_lzm_pitch(span)
                assert pitch % 64 == 32 and span <= pitch < span + 64
                wst = ws[::16]                                 # window start per tile
                for s0 in range(0, len(wst), nt):              # strips
                    assert wst[min(s0 + nt, len(wst)) - 1] - wst[s0] + 64 <= span, (S, D, nt, s0)
            else:
                assert (hi - ws).max() + ch - 1 >= 64, (S, D)  # conservative by at most the last pixel's unused channels
        n_ok += bool(pb.pb_lzm_span(ch, S, D, 8)); n_no += not pb.pb_lzm_span(ch, S, D, 8)
        if pb.pb_lzm_rows_ok(S, D):
            y0 = np.arange(0, D, 16)
            y1 = np.minimum(y0 + 15, D - 1)
            tmin, tmax = np.clip(i0[y0] - 2, 0, S - 1) >> 4, np.clip(i0[y1] + 3, 0, S - 1) >> 4
            assert (tmax - tmin <= 3).all(), (S, D)
        if pb.pb_lzm_rows_two(S, D):   # the ring of two: exact, both ways
            assert pb.pb_lzm_rows_ok(S, D) and (tmax - tmin <= 1).all(), (S, D)
            n_two += 1
        elif pb.pb_lzm_rows_ok(S, D):
            assert (tmax - tmin > 1).any(), (S, D)
    assert n_ok > 100 and n_no > 20 and n_two > 20
    for (S, D) in ((1080, 2160), (720, 1080), (1280, 1920), (1920, 3840), (640, 960), (540, 1080)):   # the up-scales the benches quote
        assert pb.pb_lzm_rows_two(S, D), (S, D)
    for (S, D) in ((1080, 1080), (1080, 720), (1000, 1080)):
        assert not pb.pb_lzm_rows_two(S, D), (S, D)
    for (S, D) in ((1920, 1280), (3840, 1920), (1280, 1920), (1920, 3840), (1080, 720), (2160, 1080), (720, 1080), (960, 640), (640, 960)):
        assert pb.pb_lzm_span(ch, S, D, 8) and pb.pb_lzm_span(ch, S, D, 4) and pb.pb_lzm_rows_ok(S, D), (S, D)


@pytest.mark.parametrize("ch", [1, 2, 3])
def test_lanczos_mfma_two_chunk_windows(pb, ch):
    """the same walk with 128-B windows (vpf_bound_lzm_span_win, two K chunks in pass 1): a non-zero span means every tap of every N-tile lies
    in [window, window + 128) and every 4-tile strip fits the span; a shape that fits the 64-B windows fits these with strips 64 B longer;
    the network-input shapes (1080p -> 416 / 640 wide, 720p -> 416 / 320, 4K -> 1440) are in, 1080p -> 224 is out"""
    rng = np.random.default_rng(177 + ch)
    n_k2 = 0
    for (S, D) in size_pairs(rng, 400, 1.5, 8.0):
        dwb = D * ch
        b = np.arange(dwb)
        px, c = b // ch, b % ch
        i0 = lz_i0(np.arange(D), S, D)
        lo = np.clip(i0 - 2, 0, S - 1)[px] * ch + c
        hi = np.clip(i0 + 3, 0, S - 1)[px] * ch + c
        first_b = np.minimum((b // 16) * 16, dwb - 1)
        ws = (np.clip(i0 - 2, 0, S - 1)[first_b // ch] * ch) & ~15
        span = pb.pb_lzm_span_win(ch, S, D, 4, 128)
        one = pb.pb_lzm_span(ch, S, D, 4)
        if one:
            assert span == one + 64, (S, D)
        if span:
            assert bool((lo >= ws).all() and (hi - ws < 128).all()), (S, D, int((hi - ws).max()))
            wst = ws[::16]
            for s0 in range(0, len(wst), 4):
                assert wst[min(s0 + 4, len(wst)) - 1] - wst[s0] + 128 <= span, (S, D, s0)
            n_k2 += not one
        else:
            assert (hi - ws).max() + ch - 1 >= 128, (S, D)
    assert n_k2 > 60
    for (S, D) in ((1920, 416), (1920, 640), (1280, 416), (1280, 320), (3840, 1440), (2560, 640)):
        assert pb.pb_lzm_span_win(ch, S, D, 4, 128) and (ch != 3 or not pb.pb_lzm_span(ch, S, D, 4)), (S, D)   # (a 1-channel plane fits 64 B up to ~2.7 x)
    assert not pb.pb_lzm_span_win(ch, 1920, 224, 4, 128)
    # ... which the 192-B windows of the three-chunk form hold (2-tile strips), like 4K -> 416; 1080p -> 120 is out of those too
    for (S, D) in ((1920, 224), (3840, 416), (960, 112)):
        span = pb.pb_lzm_span_win(ch, S, D, 2, 192)
        assert 192 < span <= 512, (S, D, span)
    assert not pb.pb_lzm_span_win(ch, 1920, 120, 2, 192)


def test_lanczos_mfma_pitch_is_conflict_free_for_the_a_operand_reads():
    """ds_read_b128 is served in four groups of 16 lanes, bank = (byte address / 4) mod 64 (MI355X_MICROARCH.md, LDS).  The A operand of pass 1:
    lane (i = lane & 15, g = lane >> 4) reads 16 B at i * pitch + 16 * g + window.  For every pitch = 32 mod 64 and every 16-B aligned window,
    the 16 lanes of each group touch 16 distinct 16-B slots of the 256-B bank period."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for pitch in range(96, 1024, 64):
        for window in range(0, 256, 16):
            for grp in groups:
                slots = {(((l & 15) * pitch + 16 * (l >> 4) + window) // 16) % 16 for l in grp}
                assert len(slots) == 16, (pitch, window)
    bad = {(((l & 15) * 256 + 16 * (l >> 4)) // 16) % 16 for l in groups[0]}
    assert len(bad) < 16   # a power-of-two pitch is what the + 32 avoids
