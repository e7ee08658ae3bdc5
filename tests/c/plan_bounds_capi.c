/* tests/c/plan_bounds_capi.c — the launchers' LDS sizing formulas (csrc/vpf_plan_bounds.h) behind C symbols, for tests/test_plan_bounds_cpu.py */
#include "vpf_plan_bounds.h"

uint32_t pb_strip_bytes(int ch, uint32_t sw, uint32_t dw, uint32_t cap, uint32_t cols) { return vpf_bound_strip_bytes(ch, sw, dw, cap, cols); }
uint32_t pb_band_slots(int r, float scy) { return vpf_bound_band_slots(r, scy); }
uint32_t pb_band_rows_exact(int r, uint32_t sh, uint32_t dh) { return vpf_band_rows_exact(r, sh, dh, (float)sh / (float)dh); }
uint32_t pb_fused_rowbytes(float scx) { return vpf_bound_fused_rowbytes(scx); }
uint32_t pb_strip_bytes_px4(uint32_t sw, uint32_t dw, uint32_t cap, uint32_t cols) { return vpf_bound_strip_bytes_px4(sw, dw, cap, cols); }
uint32_t pb_fused_rowbytes4(float scx) { return vpf_bound_fused_rowbytes4(scx); }
int pb_fused_rows_fit(int r, float scy, int strip_rows) { return vpf_bound_fused_rows_fit(r, scy, strip_rows); }
uint32_t pb_tile_rows(uint32_t ty, float scy, int taps) { return vpf_bound_tile_rows(ty, scy, taps); }
uint32_t pb_tile_rowq(float scx, int taps, int ch, int elem) { return vpf_bound_tile_rowq(scx, taps, ch, elem); }
uint32_t pb_lzm_span(int ch, uint32_t sw, uint32_t dw, int nt) { return vpf_bound_lzm_span(ch, sw, dw, (float)sw / (float)dw, nt); }
uint32_t pb_lzm_span_win(int ch, uint32_t sw, uint32_t dw, int nt, uint32_t win) { return vpf_bound_lzm_span_win(ch, sw, dw, (float)sw / (float)dw, nt, win); }
uint32_t pb_lzm_pitch(uint32_t span) { return vpf_bound_lzm_pitch(span); }
int pb_lzm_rows_ok(uint32_t sh, uint32_t dh) { return vpf_bound_lzm_rows_ok(sh, dh, (float)sh / (float)dh); }
int pb_lzm_rows_two(uint32_t sh, uint32_t dh) { return vpf_bound_lzm_rows_two(sh, dh, (float)sh / (float)dh); }
