"""A CPU model of the data flow of the MFMA Lanczos-3 kernel (csrc/k_lanczos_mfma.hip) — TEST INFRASTRUCTURE.

The kernel runs both passes of the separable filter on the integer matrix cores (v_mfma_i32_16x16x64_i8).  What can go wrong in such a
kernel is not the arithmetic — sums of products of bytes — but the bookkeeping around it: which source byte sits in which K slot of
which lane group, where a strip's windows start, which ring slot holds which 16-row tile, the constants that undo the signed-byte
offsets.  This model restates that bookkeeping in numpy with the MFMA as a plain integer matrix product over the same (lane group, slot)
layout, and tests/test_lanczos_mfma_model_cpu.py checks it against the oracle's definition of 8-bit Lanczos (vpf_oracle.c
resize_plane_lanczos, FP32 mode: H exact in Q14, rounded to Q6, V exact in Q20, rounded half up) on the CPU, where a wrong constant
costs seconds instead of a GPU visit.  The weights come from the oracle (vpfo_lanczos_taps_q14).

Layout facts modelled (kernel comments carry the same names):
  N-tile j of a strip   16 consecutive destination BYTES (pixel b / CH, channel b % CH); the strip = NT tiles
  window ws_j           16-B aligned source byte offset below the first tap of the tile's first pixel; all taps of the tile's 16 bytes
                        lie in [ws_j, ws_j + 64 kc)  (host bound, vpf_plan_bounds.h; kc = 2: two chained MFMAs per product)
  pass 1                D[row i][n] = sum_k A[i][k] B[k][n],  A = source bytes - 128 (16 rows x 64 window bytes),
                        B = Q14 weight split into two signed bytes (w = 256 wh + wl): two MFMAs -> HI, LO;  h'' = ((HI + 128) << 8) + LO + 128
                        bits 8..23 of h'' = z + 128 with z = Hr - 8192 = 256 zh + zl: byte 2 = zh (signed), byte 1 ^ 0x80 = zl (signed)
  ring                  the (zl, zh) byte PAIRS of the last four 16-row tiles, two K chunks of two tiles: tile in slot p = (T - t_first) & 3 ->
                        chunk p >> 1, K slot (g, 8 (p & 1) + 2 r + s) <-> source row 16 T + 4 g + r, s = 0: zl, s = 1: zh
  pass 2                D[n][y] = sum over the ring's 64 rows x 2 bytes: X = sum qh zh, Y = sum (ql zh + qh zl) (the weight operands carry
                        qh at the zh slots for X; ql at the zh slots and qh at the zl slots for Y; ql zl is not formed);
                        out = clamp((256 X + Y + 2^19 + 2^11) >> 12)
"""
import numpy as np


def clampi(v, lo, hi):
    return lo if v < lo else (hi if v > hi else v)


def merged_taps(i0, q, size):
    """taps i0 - 2 .. i0 + 3 clamped to [0, size - 1]; weights of taps that land on the same sample are summed into the LAST of them
    (the others are dropped) -> list of (pos, weight)"""
    pos = [clampi(i0 - 2 + k, 0, size - 1) for k in range(6)]
    w = [int(x) for x in q]
    out = []
    for k in range(6):
        if k < 5 and pos[k] == pos[k + 1]:
            w[k + 1] += w[k]
            continue
        out.append((pos[k], w[k]))
    return out


def split_i8(w):
    """w = 256 hi + lo with lo in [-128, 127]"""
    lo = ((w + 128) & 0xff) - 128
    hi = (w - lo) >> 8
    assert -128 <= hi <= 127 and w == 256 * hi + lo
    return hi, lo


def mfma_i8(A, B, c):
    """A [16][64] int8 (row i, K), B [64][16] int8 (K, col n), c scalar -> D [16][16] int32"""
    return A.astype(np.int32) @ B.astype(np.int32) + np.int32(c)


class Model:
    def __init__(self, ch, sw, sh, dw, dh, taps_x, taps_y, nt=8, band_rows=32, pitch=None, garbage_seed=1, kc=1, rt=16, up2=False):
        self.ch, self.sw, self.sh, self.dw, self.dh, self.nt, self.band = ch, sw, sh, dw, dh, nt, band_rows
        self.up2 = up2   # the ring of TWO (up-scales): every tile's taps lie in the source tiles (Tmax - 1, Tmax); one K chunk in pass 2
        self.rt = rt   # destination rows a 16-row tile carries: 16, or 8 (half tiles: rows 8 .. 15 repeat row 7 and are never stored)
        self.kc, self.win = kc, 64 * kc   # K chunks of a pass-1 window (2: the two-chunk windows of strong horizontal down-scales, 4-tile strips)
        self.i0x, self.qx = taps_x
        self.i0y, self.qy = taps_y
        self.P = pitch
        self.rng = np.random.default_rng(garbage_seed)
        self.max_k = 0          # largest window-relative source byte offset seen (< 64 kc required)
        self.max_strip = 0      # largest strip-relative end of a window (<= P required)
        self.max_tile_span = 0  # largest Tmax - Tmin of a destination tile (<= 3 required)

    def run(self, src):
        """src: [sh][>= sw * ch] uint8 -> dst [dh][dw * ch] uint8"""
        ch, nt = self.ch, self.nt
        dwb = self.dw * ch
        dst = np.zeros((self.dh, dwb), np.uint8)
        for ob0 in range(0, dwb, 16 * nt):
            for ya in range(0, self.dh, self.band):
                self.run_wave(src, dst, ob0, ya, min(ya + self.band, self.dh) - 1)
        return dst

    def run_wave(self, src, dst, ob0, ya, yb):
        ch, nt, sw, sh = self.ch, self.nt, self.sw, self.sh
        dwb = self.dw * ch
        # ---- windows
        ws = []
        for j in range(nt):
            b = min(ob0 + 16 * j, dwb - 1)   # tiles past the row end copy the last window (they hold no weights)
            pos0 = clampi(int(self.i0x[b // ch]) - 2, 0, sw - 1)
            ws.append((ch * pos0) & ~15)
        S0 = ws[0]
        # ---- pass-1 weight operands: B1[plane][j][k][n]
        B1 = np.zeros((2, nt, self.win, 16), np.int8)   # (K slot k of the window: chunk k >> 6, slot k & 63 of that chunk's MFMA)
        for b in range(ob0, min(ob0 + 16 * nt, dwb)):
            px, c, j, n = b // ch, b % ch, (b - ob0) >> 4, (b - ob0) & 15
            for pos, w in merged_taps(int(self.i0x[px]), self.qx[6 * px:6 * px + 6], sw):
                k = ch * pos + c - ws[j]
                assert 0 <= k < self.win, (k, b, j)
                self.max_k = max(self.max_k, k)
                assert B1[0, j, k, n] == 0 and B1[1, j, k, n] == 0
                B1[0, j, k, n], B1[1, j, k, n] = split_i8(w)
        self.max_strip = max(self.max_strip, ws[-1] - S0 + self.win)
        P = self.P if self.P else ((ws[-1] - S0 + self.win + 255) & ~255) + 32
        assert ws[-1] - S0 + self.win <= P
        # ---- the march
        ring = np.zeros((2, nt, 4, 16, 16), np.int8)   # [byte s: zl, zh][j][slot p][row i of the tile][n]
        ring[:] = self.rng.integers(-128, 128, ring.shape, dtype=np.int8)  # whatever it held before first use must not matter: poison
        t_first = None
        t_done = None
        rt = self.rt
        GR = 4 * rt                               # destination rows of a group of four tiles
        ngroups = (yb - ya) // GR + 1
        for G in range(ngroups):
            # lane l = (tile l >> 4, row l & 15) of the group: rows past the tile's rt repeat its last one, rows past the band repeat yb
            rows = [min(ya + GR * G + rt * (l >> 4) + min(l & 15, rt - 1), yb) for l in range(64)]
            taps = [merged_taps(int(self.i0y[y]), self.qy[6 * y:6 * y + 6], sh) for y in rows]
            ntile = min(4, (yb - (ya + GR * G)) // rt + 1)
            if t_first is None:
                t_first = taps[0][0][0] >> 4          # first source tile of the band: ring slots are numbered from it
            # vertical weight operands of the group's four destination tiles: Wm[t][X | Y][chunk c][k slot = 16 g + 8 (p & 1) + 2 r + s][y]
            Wm = np.zeros((4, 2, 2, 64, 16), np.int8)
            for l in range(64):
                t, y = l >> 4, l & 15
                tmax_t = taps[(l & ~15) | (rt - 1)][-1][0] >> 4   # the tile's last source tile: what the emit test looks at
                for pos, w in taps[l]:
                    T, g, r = pos >> 4, (pos >> 2) & 3, pos & 3
                    p = (T - t_first) & 3
                    if self.up2:   # the operand has ONE chunk: source tile Tmax - 1 in its first half, Tmax in its second
                        p = T - (tmax_t - 1)
                        assert p in (0, 1), (p, T, tmax_t)
                    slot = 16 * g + 8 * (p & 1) + 2 * r
                    qh, ql = split_i8(w)
                    assert not Wm[t, :, p >> 1, slot:slot + 2, y].any()
                    Wm[t, 0, p >> 1, slot + 1, y] = qh                                  # X: qh against zh
                    Wm[t, 1, p >> 1, slot, y], Wm[t, 1, p >> 1, slot + 1, y] = qh, ql   # Y: qh against zl, ql against zh
            for t in range(ntile):
                tmin = taps[16 * t][0][0] >> 4
                tmax = taps[16 * t + rt - 1][-1][0] >> 4
                self.max_tile_span = max(self.max_tile_span, tmax - tmin)
                assert tmax - tmin <= (1 if self.up2 else 3)
                if t_done is None:
                    t_done = tmin - 1
                while t_done < tmax:
                    t_done += 1
                    self.pass1(src, ring, B1, ws, S0, P, t_done, t_first)
                # every source tile this destination tile needs is one of the last four produced
                assert tmin >= t_done - 3
                self.emit(dst, ring, Wm[t], ob0, ya + GR * G + rt * t, yb, (tmax - t_first - 1) & 1)

    def pass1(self, src, ring, B1, ws, S0, P, T, t_first):
        ch, nt, sw, sh = self.ch, self.nt, self.sw, self.sh
        # staging: rows 16 T .. 16 T + 15 (clamped), bytes [S0, S0 + P): real bytes where the row has them, garbage elsewhere
        stage = self.rng.integers(0, 256, (16, P), dtype=np.uint8)
        rowb = src.shape[1]
        for i in range(16):
            r = clampi(16 * T + i, 0, sh - 1)
            n = max(0, min(P, rowb - S0))
            stage[i, :n] = src[r, S0:S0 + n]
        st8 = (stage ^ 0x80).view(np.int8)
        for j in range(nt):
            HI, LO = np.int32(128), np.int32(128)
            for c in range(self.kc):   # chunk c accumulates onto chunk c - 1 (the MFMA's C operand)
                A = st8[:, ws[j] - S0 + 64 * c: ws[j] - S0 + 64 * c + 64]
                HI = mfma_i8(A, B1[0, j, 64 * c:64 * c + 64], 0) + HI
                LO = mfma_i8(A, B1[1, j, 64 * c:64 * c + 64], 0) + LO
            h2 = ((HI.astype(np.int64) << 8) + LO).astype(np.int64) & 0xffffffff
            zh = ((h2 >> 16) & 0xff).astype(np.uint8).view(np.int8)
            zl = (((h2 >> 8) & 0xff) ^ 0x80).astype(np.uint8).view(np.int8)
            if self.up2:
                # the register file holds two overlapping chunks, chunk k = tiles (k, k + 1) in slots (2 (k & 1), 2 (k & 1) + 1): tile T is
                # the first half of chunk T and the second half of chunk T - 1
                rel = T - t_first
                for s_, z in ((0, zl), (1, zh)):
                    ring[s_, j, 2 * (rel & 1)] = z
                    ring[s_, j, 2 * ((rel & 1) ^ 1) + 1] = z
            else:
                ring[0, j, (T - t_first) & 3], ring[1, j, (T - t_first) & 3] = zl, zh

    def emit(self, dst, ring, W, ob0, y0, yb, chunk=0):
        ch, nt = self.ch, self.nt
        dwb = self.dw * ch
        for j in range(nt):
            # chunk c: A2[n][k slot = 16 g + 8 (p & 1) + 2 r + s] = ring[s, j, p, row 4 g + r, n] for the two tiles p = 2 c, 2 c + 1
            X = np.zeros((16, 16), np.int32)
            Y = np.full((16, 16), (1 << 19) + (1 << 11), np.int32)
            for c in ((chunk,) if self.up2 else range(2)):
                A2 = np.zeros((16, 64), np.int8)
                for g in range(4):
                    for pp in range(2):
                        for r in range(4):
                            for sb in range(2):
                                A2[:, 16 * g + 8 * pp + 2 * r + sb] = ring[sb, j, 2 * c + pp, 4 * g + r, :]
                X = X + mfma_i8(A2, W[0, 0 if self.up2 else c], 0)
                Y = Y + mfma_i8(A2, W[1, 0 if self.up2 else c], 0)
            V = (X.astype(np.int64) << 8) + Y
            assert np.all(np.abs(V) < 2 ** 31)
            out = np.clip(V >> 12, 0, 255).astype(np.uint8)   # [n][y]
            for y in range(self.rt):
                if y0 + y > yb:
                    break
                for n in range(16):
                    b = ob0 + 16 * j + n
                    if b < dwb:
                        dst[y0 + y, b] = out[n, y]
