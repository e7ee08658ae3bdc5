/* vpf_plan_bounds.h — the LDS sizing bounds of the strip-staging resize kernels, as plain C so that the launchers (k_resize.hip,
 * k_convert_resize.hip, k_lanczos_mfma.hip through vpf_lzm_plan.h) and a CPU property test (tests/test_plan_bounds_cpu.py, compiled with gcc) use the SAME formulas.
 *
 * Every kernel of the row-pair / row-band / fused-strip / tiled families copies the source bytes its destination columns touch into wave-private LDS
 * strips whose size the HOST computes from the scale factors.  A bound that is one byte or one row short is silent memory corruption
 * on the device, so each formula here is checked on the CPU against the kernels' exact fp32 tap arithmetic over thousands of
 * (source size, destination size, position) combinations. */
#ifndef VPF_PLAN_BOUNDS_H_
#define VPF_PLAN_BOUNDS_H_
#include <math.h>
#include <stdint.h>

/* strip bytes for the source span of `cols` destination columns of a `ch`-channel plane (bilinear taps): <= (cols - 1) * scale + 3
 * pixels + 16-B alignment slack on both ends + the 12-B tap window's over-read, rounded up to 256; 0 when above `cap` */
static inline uint32_t vpf_bound_strip_bytes(int ch, uint32_t sw, uint32_t dw, uint32_t cap, uint32_t cols) {
  const double scale = (double)sw / (double)dw;
  const double need = ((double)(cols - 1) * scale + 4.0) * ch + 32.0;
  if (need > (double)cap) return 0;
  return ((uint32_t)need + 255u) & ~255u;
}

/* the row-band kernels' strips of packed RGB widened to four bytes per pixel (k_bilinear_blend.h "px4", round 6): pixels [first & ~3, last + 1]
 * in whole units of four — (cols - 1) * scale + 3 pixels of taps + 3 of alignment + 1 for the second tap's dword at the right edge + 3 of unit
 * rounding, 4 B each — rounded up to 64 B (LDS rows are what decides how many workgroups a CU holds: 256-B rounding would cost 1080p -> 720p
 * one of its four); 0 when above `cap` */
static inline uint32_t vpf_bound_strip_bytes_px4(uint32_t sw, uint32_t dw, uint32_t cap, uint32_t cols) {
  const double scale = (double)sw / (double)dw;
  const double need = ((double)(cols - 1) * scale + 11.0) * 4.0;
  if (need > (double)cap) return 0;
  return ((uint32_t)need + 63u) & ~63u;
}

/* source rows a band of `r` destination rows can touch (bilinear): i1(last row) - i0(first row) + 1 <= floor((r - 1) scy) + 3 (+ fp32 slack) */
static inline uint32_t vpf_bound_band_slots(int r, float scy) {
  return (uint32_t)((double)(r - 1) * (double)scy + 0.01) + 3u;
}

/* the same, EXACT: the largest i1(last row) - i0(first row) + 1 over the bands of `r` destination rows of a sh -> dh resize, walked with
 * make_tap's own fp32 arithmetic (k_bilinear_blend.h; scy = (float)sh / (float)dh as the launchers pass it).  The closed form above
 * is one or two rows generous exactly where it costs most — 1080 -> 720 with r = 4 touches 6 rows, the bound says 8 — and LDS rows are
 * what decides how many workgroups a CU holds.  dh / r iterations; the launchers remember the last shapes. */
static inline uint32_t vpf_lin_i0(uint32_t d, float scale, uint32_t size) {
  float s = fmaf((float)d + 0.5f, scale, -0.5f);
  s = fmaxf(s, 0.f);
  s = fminf(s, (float)(size - 1));
  return (uint32_t)(int32_t)s;
}
static inline uint32_t vpf_band_rows_exact(int r, uint32_t sh, uint32_t dh, float scy) {
  uint32_t most = 1;
  for (uint32_t ya = 0; ya < dh; ya += (uint32_t)r) {
    const uint32_t yb = ya + (uint32_t)r - 1 < dh - 1 ? ya + (uint32_t)r - 1 : dh - 1;
    const uint32_t lo = vpf_lin_i0(ya, scy, sh), hi0 = vpf_lin_i0(yb, scy, sh), hi = hi0 + 1 < sh ? hi0 + 1 : sh - 1;
    most = hi - lo + 1 > most ? hi - lo + 1 : most;
  }
  return most;
}

/* fused convert + resize strip kernel: bytes per strip row of packed RGB for a wave's 256 destination columns (source span rounded out to
 * 8-pixel conversion groups on both sides + the tap window's over-read), and whether `r` destination rows fit `strip_rows` source rows */
static inline uint32_t vpf_bound_fused_rowbytes(float scx) {
  return (((uint32_t)(255.0 * (double)scx) + 2 + 8 + 8) * 3 + 16 + 15) & ~15u;
}
/* the same strip with FOUR bytes per pixel (R G B x: k_bilinear_blend.h "px4"; the workgroup-shared strips of round 6): the same pixel span +
 * one pixel for the second tap's dword at the right edge */
static inline uint32_t vpf_bound_fused_rowbytes4(float scx) {
  return (((uint32_t)(255.0 * (double)scx) + 2 + 8 + 8 + 1) * 4 + 15) & ~15u;
}
static inline int vpf_bound_fused_rows_fit(int r, float scy, int strip_rows) {  /* rows touched <= (r - 1) scy + 3 (+ fp32 slack) */
  return (double)scy * (double)(r - 1) + 3.01 <= (double)strip_rows;
}

/* tiled separable kernels (64 destination columns x ty rows per workgroup, `taps` = 6 Lanczos / 2 bilinear, `elem` bytes per sample):
 * source rows a tile can touch, and 16-B units per staged source row (8-bit Lanczos rows carry a pad unit and a replicated right margin) */
static inline uint32_t vpf_bound_tile_rows(uint32_t ty, float scy, int taps) {
  return (uint32_t)((double)(ty - 1) * (double)scy) + (uint32_t)taps + 2;
}
static inline uint32_t vpf_bound_tile_rowq(float scx, int taps, int ch, int elem) {
  return (uint32_t)((((double)scx * 63.0 + (double)taps + 3.0) * ch * elem + 32.0) / 16.0) + 1 + ((taps == 6 && elem == 1) ? 2 : 0);
}

/* MFMA Lanczos kernel (k_lanczos_mfma.hip; a CPU model of its bookkeeping: tests/lanczos_mfma_model.py).  A wave owns `nt` N-tiles of 16
 * destination BYTES each; every tap of a tile's bytes must lie in the 64-B window that starts at the 16-B aligned source byte below the
 * first tap of the tile's first pixel (K = 64 of v_mfma_i32_16x16x64_i8), and a 16-row destination tile must find all its source rows in
 * four consecutive 16-row source tiles (the ring).  Closed-form bounds on floor differences are one too pessimistic exactly at the ratios
 * that matter (2.0 with three channels), so the launcher WALKS the tiles with the kernel's own fp32 coordinate arithmetic — a few hundred
 * fma + floor per plane shape, cached per thread — instead of estimating. */
static inline int32_t vpf_lz_i0(uint32_t d, float scale) { return (int32_t)floorf(fmaf((float)d + 0.5f, scale, -0.5f)); }  /* ltap_i0 of the kernels */
static inline uint32_t vpf_lz_clamp(int32_t i, uint32_t size) { return i < 0 ? 0u : (i > (int32_t)size - 1 ? size - 1u : (uint32_t)i); }
/* 0 when some tile's taps do not fit its window; else the bytes a wave stages per source row: the largest (window of a strip's last tile
 * - window of its first tile) + 64 over all strips of `nt` tiles */
static inline uint32_t vpf_bound_lzm_span_win(int ch, uint32_t sw, uint32_t dw, float scx, int nt, uint32_t win /* window bytes: 64 per K chunk of pass 1 */) {
  const uint32_t dwb = dw * (uint32_t)ch, ntile = (dwb + 15u) / 16u;
  uint32_t span = 0, ws0 = 0;
  for (uint32_t t = 0; t < ntile; t++) {
    const uint32_t b0 = 16u * t, b1 = b0 + 15u < dwb - 1u ? b0 + 15u : dwb - 1u;
    const uint32_t ws = ((uint32_t)ch * vpf_lz_clamp(vpf_lz_i0(b0 / (uint32_t)ch, scx) - 2, sw)) & ~15u;
    /* highest source byte of the tile: the last tap of its last pixel, any channel (the taps of earlier pixels lie below: i0 is monotone) */
    const uint32_t hi = (uint32_t)ch * vpf_lz_clamp(vpf_lz_i0(b1 / (uint32_t)ch, scx) + 3, sw) + (uint32_t)ch - 1u;
    if (hi - ws >= win) return 0;
    if (t % (uint32_t)nt == 0) ws0 = ws;
    if (ws - ws0 + win > span) span = ws - ws0 + win;
  }
  return span;
}
static inline uint32_t vpf_bound_lzm_span(int ch, uint32_t sw, uint32_t dw, float scx, int nt) { return vpf_bound_lzm_span_win(ch, sw, dw, scx, nt, 64u); }
static inline uint32_t vpf_bound_lzm_pitch(uint32_t span) { return ((span + 31u) & ~63u) + 32u; }
/* every 16-row destination tile (tiles start at multiples of 16: bands are whole tiles) spans at most `tiles` 16-row source tiles: four (the
 * ring), or two — the ring of two of the up-scales, vertical factors up to ~0.7 (16 destination rows + 5 rows of taps on <= 17 source rows), walked like everything here */
static inline int vpf_bound_lzm_rows_within(uint32_t sh, uint32_t dh, float scy, uint32_t rt /* destination rows a tile carries: 16, or 8 (half tiles) */, uint32_t tiles) {
  for (uint32_t y0 = 0; y0 < dh; y0 += rt) {
    const uint32_t y1 = y0 + rt - 1u < dh - 1u ? y0 + rt - 1u : dh - 1u;
    if ((vpf_lz_clamp(vpf_lz_i0(y1, scy) + 3, sh) >> 4) - (vpf_lz_clamp(vpf_lz_i0(y0, scy) - 2, sh) >> 4) > tiles - 1u) return 0;
  }
  return 1;
}
static inline int vpf_bound_lzm_rows_ok_rt(uint32_t sh, uint32_t dh, float scy, uint32_t rt) { return vpf_bound_lzm_rows_within(sh, dh, scy, rt, 4u); }
static inline int vpf_bound_lzm_rows_ok(uint32_t sh, uint32_t dh, float scy) { return vpf_bound_lzm_rows_ok_rt(sh, dh, scy, 16u); }
static inline int vpf_bound_lzm_rows_two(uint32_t sh, uint32_t dh, float scy) { return vpf_bound_lzm_rows_within(sh, dh, scy, 16u, 2u); }

#endif /* VPF_PLAN_BOUNDS_H_ */
