/* vpf_plan_bounds.h — the LDS sizing bounds of the strip-staging resize kernels, as plain C so that the launchers (k_resize.hip,
 * k_convert_resize.hip) and a CPU property test (tests/test_plan_bounds_cpu.py, compiled with gcc) use the SAME formulas.
 *
 * Every kernel of the row-pair / row-band / march families copies the source bytes its destination columns touch into wave-private LDS
 * strips whose size the HOST computes from the scale factors.  A bound that is one byte or one row short is silent memory corruption
 * on the device, so each formula here is checked on the CPU against the kernels' exact fp32 tap arithmetic over thousands of
 * (source size, destination size, position) combinations. */
#ifndef VPF_PLAN_BOUNDS_H_
#define VPF_PLAN_BOUNDS_H_
#include <stdint.h>

/* strip bytes for the source span of `cols` destination columns of a `ch`-channel plane (bilinear taps): <= (cols - 1) * scale + 3
 * pixels + 16-B alignment slack on both ends + the 12-B tap window's over-read, rounded up to 256; 0 when above `cap` */
static inline uint32_t vpf_bound_strip_bytes(int ch, uint32_t sw, uint32_t dw, uint32_t cap, uint32_t cols) {
  const double scale = (double)sw / (double)dw;
  const double need = ((double)(cols - 1) * scale + 4.0) * ch + 32.0;
  if (need > (double)cap) return 0;
  return ((uint32_t)need + 255u) & ~255u;
}

/* source rows a band of `r` destination rows can touch (bilinear): i1(last row) - i0(first row) + 1 <= floor((r - 1) scy) + 3 (+ fp32 slack) */
static inline uint32_t vpf_bound_band_slots(int r, float scy) {
  return (uint32_t)((double)(r - 1) * (double)scy + 0.01) + 3u;
}

/* Lanczos march: 16-B units per strip of a plane whose waves own `wcols` destination columns: pad unit (replicated left margin) + base
 * alignment + six-tap span + replicated right margin and the tap run's over-read; 0 when the span does not fit two 1-KiB staging passes */
#define VPF_MARCH_PAD 16u
static inline uint32_t vpf_bound_march_rowq(int ch, uint32_t sw, uint32_t dw, uint32_t wcols) {
  const double scx = (double)sw / (double)dw;
  const uint32_t span_px = (uint32_t)(((double)wcols - 1.0) * scx) + 8;  /* taps of a wave's columns: floor((W - 1) scx) + 6 (+ fp32 slack) */
  if ((uint32_t)ch * span_px > 2018u) return 0;
  return (VPF_MARCH_PAD + 15u + (uint32_t)ch * span_px + 8u + 15u) / 16u;
}

/* fused convert + resize strip kernel: bytes per strip row of packed RGB for a wave's 256 destination columns (source span rounded out to
 * 8-pixel conversion groups on both sides + the tap window's over-read), and whether `r` destination rows fit `strip_rows` source rows */
static inline uint32_t vpf_bound_fused_rowbytes(float scx) {
  return (((uint32_t)(255.0 * (double)scx) + 2 + 8 + 8) * 3 + 16 + 15) & ~15u;
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

#endif /* VPF_PLAN_BOUNDS_H_ */
