// vpf_coef.h — YUV -> RGB colour matrices in the form the kernels consume (shared by libvpfhip's ABI layer and the measurement lab).
#pragma once
#include "vpf_internal.h"

namespace vpf {

// ------------------------------------------------------------------------------------------
// colour matrices.  Decimal coefficients x 1e6 as published for NPP's colour models / BT.601 /
// BT.709.  Everything the device sees is derived from these integers by one
// correctly rounded division and one narrowing, so host and device agree bit for bit everywhere.
// ------------------------------------------------------------------------------------------
struct Yuv2RgbDec {
  int64_t cy, rv, gu, gv, bu;
  int off;
};
static constexpr Yuv2RgbDec kYuv2Rgb[2][2] = {
    {{1164000, 1596000, -392000, -813000, 2017000, 16},   // BT.601 MPEG : NPP "YCbCr"
     {1000000, 1140000, -394000, -581000, 2032000, 0}},   // BT.601 JPEG : NPP "YUV"
    {{1164384, 1792741, -213249, -532909, 2112402, 16},   // BT.709 MPEG : "709CSC"
     {1000000, 1574800, -187324, -468124, 1855600, 0}}};  // BT.709 JPEG : "709HDTV"

static inline float q6(int64_t v) { return (float)((double)v / 1e6); }

static inline bool coef_yuv2rgb(int cs, int cr, Yuv2RgbCoef* o) {
  if ((cs != VPF_BT_601 && cs != VPF_BT_709) || (cr != VPF_MPEG && cr != VPF_JPEG)) return false;
  const Yuv2RgbDec& m = kYuv2Rgb[cs][cr];
  o->cy = q6(m.cy); o->rv = q6(m.rv); o->gu = q6(m.gu); o->gv = q6(m.gv); o->bu = q6(m.bu);
  const int64_t yoff = -(int64_t)m.off * m.cy;  // luma offset (rounding is done by v_cvt_pk_u8_f32: nearest even)
  o->br = q6(yoff - 128 * m.rv);
  o->bg = q6(yoff - 128 * (m.gu + m.gv));
  o->bb = q6(yoff - 128 * m.bu);
  return true;
}

}  // namespace vpf
