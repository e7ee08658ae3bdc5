// k_convert_resize.hip — fused NV12 / YUV420 -> bilinear -> RGB / BGR / RGB_PLANAR (gfx950): vpf_convert_resize(_batch), the additive
// one-pass form of PySurfaceConverter + PySurfaceResizer (reference call sites: TasksColorCvt.cpp:145-155 then Tasks.cpp:1193).
//   k_convert_strip          general scale factors: convert the source window once into an LDS RGB strip, blend from bytes (band walk)
//   k_convert_resize_lds     per-tap conversion out of LDS strips (odd integer factors: centre-sample shortcut)
//   k_convert_half           exact 2x: quad-structured, integer blend
//   k_convert_resize         gather form: any size / alignment
#include <cstring>

#include "k_bilinear_blend.h"
#include "k_resize_common.h"

namespace vpf {

// ------------------------------------------------------------------------------------------
// (Stores: plain here — non-temporal stores measured 1.44 -> 1.61 us per 4K -> 720p frame in the batched fused kernel,
// while the unfused resize kernels gain from them on large outputs: 1080p -> 4K 16.0 -> 13.8 us.)
// fused NV12 / YUV420 -> bilinear -> RGB / BGR / RGB_PLANAR.  Defined as convert-then-resize: each of
// the four source texels is converted to 8-bit RGB with exactly vpf_convert's arithmetic (including
// its rounding), then interpolated — bit-identical to running the two kernels back to back, but the
// 3 B/px intermediate never exists: 12.4 MB read + 2.8 MB written instead of 65 MB for 4K -> 720p.
// ------------------------------------------------------------------------------------------
template <int SRC>
VPF_DEV void texel_rgb(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t x, uint32_t y, float* rgb) {
  const float yf = (float)f.s[0][(size_t)y * f.sp[0] + x];
  float u, v;
  if constexpr (SRC == FC_NV12) {
    const uint8_t* p = f.s[1] + (size_t)(y >> 1) * f.sp[1] + 2 * (x >> 1);
    u = p[0]; v = p[1];
  } else {
    u = f.s[1][(size_t)(y >> 1) * f.sp[1] + (x >> 1)]; v = f.s[2][(size_t)(y >> 1) * f.sp[2] + (x >> 1)];
  }
  const Chroma k = chroma_terms(c, u, v);
  rgb[0] = (float)sat_rne(__builtin_fmaf(yf, c.cy, k.rc));
  rgb[1] = (float)sat_rne(__builtin_fmaf(yf, c.cy, k.gc));
  rgb[2] = (float)sat_rne(__builtin_fmaf(yf, c.cy, k.bc));
}

template <int SRC, int DST, class BA = BatchArgs>  // BA: the frame table's size (<= 32 / <= 128 frames: vpf_internal.h)
__global__ __launch_bounds__(256) void k_convert_resize(const BA args, const Yuv2RgbCoef c, uint32_t sw,
                                                        uint32_t sh, uint32_t dw, uint32_t dh, float scx, float scy,
                                                        int vec_ok) {
  const FrameDesc f = args.f[blockIdx.z];
  const uint32_t gx = blockIdx.x * 64 + (threadIdx.x & 63);
  const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const uint32_t x0 = gx * 4;
  if (x0 >= dw || y >= dh) return;
  const Tap ty = make_tap<VPF_INTERP_LINEAR>(y, scy, sh);
  float o[3][4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t x = (x0 + k < dw) ? x0 + k : dw - 1;
    const Tap tx = make_tap<VPF_INTERP_LINEAR>(x, scx, sw);
    float p00[3], p01[3], p10[3], p11[3];
    texel_rgb<SRC>(f, c, tx.i0, ty.i0, p00);
    texel_rgb<SRC>(f, c, tx.i1, ty.i0, p01);
    texel_rgb<SRC>(f, c, tx.i0, ty.i1, p10);
    texel_rgb<SRC>(f, c, tx.i1, ty.i1, p11);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) o[ch][k] = bilerp(p00[ch], p01[ch], p10[ch], p11[ch], tx.f, ty.f);
  }
  const uint32_t nv = dw - x0 < 4 ? dw - x0 : 4;
  if constexpr (DST == FC_PLANAR) {
    for (int ch = 0; ch < 3; ch++) {
      uint8_t* out = f.d[ch] + (size_t)y * f.dp[ch] + x0;
      if (vec_ok && nv == 4) stg<false, uint32_t>(out, pack4_trunc(o[ch][0], o[ch][1], o[ch][2], o[ch][3]));
      else for (uint32_t i = 0; i < nv; i++) out[i] = (uint8_t)(uint32_t)(o[ch][i]);
    }
  } else {
    const int a = (DST == FC_BGR) ? 2 : 0, b = (DST == FC_BGR) ? 0 : 2;
    uint8_t* out = f.d[0] + (size_t)y * f.dp[0] + 3 * (size_t)x0;
    if (vec_ok && nv == 4) {
      const float t[12] = {o[a][0], o[1][0], o[b][0], o[a][1], o[1][1], o[b][1], o[a][2], o[1][2], o[b][2], o[a][3], o[1][3], o[b][3]};
      uint32_t d0, d1, d2;
      pack12_trunc(t, d0, d1, d2);
      stg3<false>(out, d0, d1, d2);
    } else {
      for (uint32_t i = 0; i < nv; i++) {
        out[3 * i] = (uint8_t)(uint32_t)(o[a][i]); out[3 * i + 1] = (uint8_t)(uint32_t)(o[1][i]); out[3 * i + 2] = (uint8_t)(uint32_t)(o[b][i]);
      }
    }
  }
}

// LDS-staged fused kernel: the wave stages the luma spans of source rows y0,y1 and the chroma spans of rows y0>>1,
// y1>>1 (4 coalesced strips), then converts the four taps of each destination pixel from LDS.  Bit-identical to
// k_convert_resize.  NV12: chroma strip holds interleaved UV; YUV420: U strip then V strip.
// This kernel is VALU-bound, not HBM-bound: four full conversions (incl. the u8 rounding that keeps it bit-identical to
// convert-then-resize) per destination pixel is ~500 VALU instructions per wave of 256 px, i.e. ~2.9 us per 4K->720p
// frame of pure issue time on 1024 SIMDs; measured 3.2 us batched.
constexpr uint32_t kFusedRowBytes = 2048;  // cap; the launch sizes the strips for its own scale factor (dyn_strip)

template <int SRC, int DST, int IT>
VPF_DEV void convert_resize_lds_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, float scx,
                                     float scy, int vec_ok, uint32_t rowq, uint32_t bx, uint32_t by) {
  // per wave, NS strips of rowq x 16 B in dynamic LDS: 0,1 luma rows; 2,3 chroma rows (NV12: UV interleaved | YUV420: U);
  // 4,5 V rows (YUV420 only)
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t y = by * 4 + wv;
  if (y >= dh) return;
  const uint32_t xs = bx * 256, xe = (xs + 255 < dw - 1) ? xs + 255 : dw - 1;
  const Tap ty = make_tap<VPF_INTERP_LINEAR>(y, scy, sh);
  const uint32_t first = make_tap<VPF_INTERP_LINEAR>(xs, scx, sw).i0, last = make_tap<VPF_INTERP_LINEAR>(xe, scx, sw).i1;
  const uint32_t ybase = first & ~15u, ynq = (last + 1 - ybase + 15) / 16;
  uint32_t cbase, cnq;
  if constexpr (SRC == FC_NV12) {  // chroma bytes [2*(first>>1), 2*(last>>1)+2)
    cbase = (2 * (first >> 1)) & ~15u; cnq = (2 * (last >> 1) + 2 - cbase + 15) / 16;
  } else {                         // chroma bytes [first>>1, (last>>1)+1)
    cbase = (first >> 1) & ~15u; cnq = ((last >> 1) + 1 - cbase + 15) / 16;
  }
  constexpr int NS = (SRC == FC_NV12) ? 4 : 6;
  // both source rows usually sit on ONE chroma row when the upper one is even: wave-uniform, so the second chroma
  // strip is neither loaded nor converted (its chroma terms are the first row's)
  // Exact-alignment shortcuts (bit-identical: fma(0, anything finite, t) == t).  With an odd integer scale factor (4K ->
  // 720p is 3x) every destination pixel centre falls on a source pixel centre: fy == 0 for the whole row (wave-uniform:
  // the second source row is neither loaded nor converted) and fx == 0 in every lane (checked per pixel with a wave
  // vote: the second tap is not converted).  That is 4 instead of 16 conversions per lane and half the source rows.
  const bool row1 = __builtin_amdgcn_readfirstlane(__float_as_uint(ty.f)) != 0u;
  const bool one_crow = !row1 || (ty.i0 >> 1) == (ty.i1 >> 1);
  Span<IT> sp_[NS];  // every strip's loads are in flight before the first LDS write
  sp_[0].load(f.s[0] + (size_t)ty.i0 * f.sp[0], ybase, ynq, lane);
  if (row1) sp_[1].load(f.s[0] + (size_t)ty.i1 * f.sp[0], ybase, ynq, lane);
  sp_[2].load(f.s[1] + (size_t)(ty.i0 >> 1) * f.sp[1], cbase, cnq, lane);
  if (!one_crow) sp_[3].load(f.s[1] + (size_t)(ty.i1 >> 1) * f.sp[1], cbase, cnq, lane);
  if constexpr (SRC != FC_NV12) {
    sp_[4].load(f.s[2] + (size_t)(ty.i0 >> 1) * f.sp[2], cbase, cnq, lane);
    if (!one_crow) sp_[5].load(f.s[2] + (size_t)(ty.i1 >> 1) * f.sp[2], cbase, cnq, lane);
  }
  u32x4* const wstrip = dyn_strip + wv * NS * rowq;
  auto strip_at = [&](int k) { return wstrip + k * rowq; };
  sp_[0].store(strip_at(0), ynq, lane);
  if (row1) sp_[1].store(strip_at(1), ynq, lane);
  sp_[2].store(strip_at(2), cnq, lane);
  if (!one_crow) sp_[3].store(strip_at(3), cnq, lane);
  if constexpr (SRC != FC_NV12) {
    sp_[4].store(strip_at(4), cnq, lane);
    if (!one_crow) sp_[5].store(strip_at(5), cnq, lane);
  }
  wave_lds_sync();
  const uint32_t x0 = xs + lane * 4;
  if (x0 >= dw) return;
  // Two horizontally adjacent taps are converted together on the packed-fp32 pipe (v_pk_fma_f32: two independent
  // IEEE fmas per instruction, so every component is bit-identical to vpf_convert's scalar fma chain).  Measured
  // against the scalar spelling it is a wash (a packed op costs two issue slots): kept for the shorter instruction stream.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 cy2 = {c.cy, c.cy}, rv2 = {c.rv, c.rv}, gu2 = {c.gu, c.gu}, gv2 = {c.gv, c.gv}, bu2 = {c.bu, c.bu};
  const f32x2 br2 = {c.br, c.br}, bg2 = {c.bg, c.bg}, bb2 = {c.bb, c.bb};
  struct Chroma2 { f32x2 rc, gc, bc; };
  auto chroma2 = [&](uint32_t a0, uint32_t a1, int r) {  // chroma terms of taps i0, i1 on chroma strip r
    f32x2 u, v;
    if constexpr (SRC == FC_NV12) {
      const uint8_t* p = reinterpret_cast<const uint8_t*>(strip_at(2 + r));
      const uint32_t d0 = *reinterpret_cast<const uint16_t*>(p + a0), d1 = *reinterpret_cast<const uint16_t*>(p + a1);
      u = f32x2{ubyte<0>(d0), ubyte<0>(d1)}; v = f32x2{ubyte<1>(d0), ubyte<1>(d1)};
    } else {
      const uint8_t* pu = reinterpret_cast<const uint8_t*>(strip_at(2 + r));
      const uint8_t* pv = reinterpret_cast<const uint8_t*>(strip_at(4 + r));
      u = f32x2{(float)pu[a0], (float)pu[a1]}; v = f32x2{(float)pv[a0], (float)pv[a1]};
    }
    Chroma2 k;
    k.rc = __builtin_elementwise_fma(v, rv2, br2);
    k.gc = __builtin_elementwise_fma(u, gu2, __builtin_elementwise_fma(v, gv2, bg2));
    k.bc = __builtin_elementwise_fma(u, bu2, bb2);
    return k;
  };
  auto rnd = [](f32x2 t) { return f32x2{(float)sat_rne(t[0]), (float)sat_rne(t[1])}; };
  // horizontal lerp of one source row: top[c] = fma(fx, p1[c] - p0[c], p0[c]) on the converted + rounded taps
  auto row_pair = [&](int r, const Chroma2& kk, uint32_t l0, uint32_t l1, float fx, float t3[3]) {
    const uint8_t* yp = reinterpret_cast<const uint8_t*>(strip_at(r));
    const f32x2 yv = {(float)yp[l0], (float)yp[l1]};
    const f32x2 rr = rnd(__builtin_elementwise_fma(yv, cy2, kk.rc)), gg = rnd(__builtin_elementwise_fma(yv, cy2, kk.gc)), bb = rnd(__builtin_elementwise_fma(yv, cy2, kk.bc));
    t3[0] = __builtin_fmaf(fx, rr[1] - rr[0], rr[0]); t3[1] = __builtin_fmaf(fx, gg[1] - gg[0], gg[0]); t3[2] = __builtin_fmaf(fx, bb[1] - bb[0], bb[0]);
  };
  auto chroma1 = [&](uint32_t a0, int cr) {  // chroma terms of the first tap only
    float u, v;
    if constexpr (SRC == FC_NV12) {
      const uint32_t d0 = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(strip_at(2 + cr)) + a0);
      u = ubyte<0>(d0); v = ubyte<1>(d0);
    } else {
      u = (float)reinterpret_cast<const uint8_t*>(strip_at(2 + cr))[a0]; v = (float)reinterpret_cast<const uint8_t*>(strip_at(4 + cr))[a0];
    }
    return chroma_terms(c, u, v);
  };
  auto row_single = [&](int r, const Chroma& kk, uint32_t l0, float t3[3]) {  // fx == 0 in every lane: top == first tap
    const float yv = (float)reinterpret_cast<const uint8_t*>(strip_at(r))[l0];
    t3[0] = (float)sat_rne(__builtin_fmaf(yv, c.cy, kk.rc)); t3[1] = (float)sat_rne(__builtin_fmaf(yv, c.cy, kk.gc)); t3[2] = (float)sat_rne(__builtin_fmaf(yv, c.cy, kk.bc));
  };
  const uint32_t nv = dw - x0 < 4 ? dw - x0 : 4;
  const uint32_t kx = (uint32_t)scx;
  if (!row1 && (float)kx == scx && (kx & 1u) && sw == kx * dw) {
    // Odd integer scale factors (kernel-uniform on x, row-uniform on y): every destination pixel IS one converted source
    // pixel — (x + 0.5) * k - 0.5 = k x + (k - 1) / 2 exactly, in float as well (all values < 2^24).  The general path
    // would round it to 8 bits (sat_rne), go back to float, add 0.5, clamp and truncate — which returns the same integer —
    // so the conversion's own v_cvt_pk_u8_f32 writes the destination bytes directly: ~25 instead of ~60 VALU per pixel.
    float v[3][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i0 = kx * ((x0 + k < dw) ? x0 + k : dw - 1) + (kx >> 1);
      const uint32_t l0 = i0 - ybase, a0 = (SRC == FC_NV12 ? (i0 & ~1u) : (i0 >> 1)) - cbase;
      const Chroma ka = chroma1(a0, 0);
      const float yv = (float)reinterpret_cast<const uint8_t*>(strip_at(0))[l0];
      v[0][k] = __builtin_fmaf(yv, c.cy, ka.rc); v[1][k] = __builtin_fmaf(yv, c.cy, ka.gc); v[2][k] = __builtin_fmaf(yv, c.cy, ka.bc);
    }
    if constexpr (DST == FC_PLANAR) {
      for (int ch = 0; ch < 3; ch++) {
        uint8_t* out = f.d[ch] + (size_t)y * f.dp[ch] + x0;
        if (vec_ok && nv == 4) stg<false, uint32_t>(out, pack4<1>(v[ch][0], v[ch][1], v[ch][2], v[ch][3]));
        else for (uint32_t i = 0; i < nv; i++) out[i] = (uint8_t)sat_rne(v[ch][i]);
      }
    } else {
      const int a = (DST == FC_BGR) ? 2 : 0, b = (DST == FC_BGR) ? 0 : 2;
      uint8_t* out = f.d[0] + (size_t)y * f.dp[0] + 3 * (size_t)x0;
      if (vec_ok && nv == 4) {
        stg3<false>(out, pack4<1>(v[a][0], v[1][0], v[b][0], v[a][1]), pack4<1>(v[1][1], v[b][1], v[a][2], v[1][2]),
                    pack4<1>(v[b][2], v[a][3], v[1][3], v[b][3]));
      } else {
        for (uint32_t i = 0; i < nv; i++) {
          out[3 * i] = (uint8_t)sat_rne(v[a][i]); out[3 * i + 1] = (uint8_t)sat_rne(v[1][i]); out[3 * i + 2] = (uint8_t)sat_rne(v[b][i]);
        }
      }
    }
    return;
  }
  float o[3][4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t x = (x0 + k < dw) ? x0 + k : dw - 1;
    const Tap tx = make_tap<VPF_INTERP_LINEAR>(x, scx, sw);
    const uint32_t l0 = tx.i0 - ybase, l1 = tx.i1 - ybase;
    const uint32_t a0 = (SRC == FC_NV12 ? (tx.i0 & ~1u) : (tx.i0 >> 1)) - cbase, a1 = (SRC == FC_NV12 ? (tx.i1 & ~1u) : (tx.i1 >> 1)) - cbase;
    const bool tap1 = __builtin_amdgcn_ballot_w64(tx.f != 0.f) != 0;  // wave-uniform
    float top[3], bot[3];
    if (tap1) {
      const Chroma2 ka = chroma2(a0, a1, 0);
      row_pair(0, ka, l0, l1, tx.f, top);
      if (row1) {
        if (one_crow) row_pair(1, ka, l0, l1, tx.f, bot);  // both rows sit on one chroma row: its terms are reused
        else row_pair(1, chroma2(a0, a1, 1), l0, l1, tx.f, bot);
      }
    } else {
      const Chroma ka = chroma1(a0, 0);
      row_single(0, ka, l0, top);
      if (row1) {
        if (one_crow) row_single(1, ka, l0, bot);
        else row_single(1, chroma1(a0, 1), l0, bot);
      }
    }
    if (row1) {
#pragma unroll
      for (int ch = 0; ch < 3; ch++) o[ch][k] = __builtin_fmaf(ty.f, bot[ch] - top[ch], top[ch]) + 0.5f;
    } else {
#pragma unroll
      for (int ch = 0; ch < 3; ch++) o[ch][k] = top[ch] + 0.5f;
    }
  }
  if constexpr (DST == FC_PLANAR) {
    for (int ch = 0; ch < 3; ch++) {
      uint8_t* out = f.d[ch] + (size_t)y * f.dp[ch] + x0;
      if (vec_ok && nv == 4) stg<false, uint32_t>(out, pack4_trunc(o[ch][0], o[ch][1], o[ch][2], o[ch][3]));
      else for (uint32_t i = 0; i < nv; i++) out[i] = (uint8_t)(uint32_t)(o[ch][i]);
    }
  } else {
    const int a = (DST == FC_BGR) ? 2 : 0, b = (DST == FC_BGR) ? 0 : 2;
    uint8_t* out = f.d[0] + (size_t)y * f.dp[0] + 3 * (size_t)x0;
    if (vec_ok && nv == 4) {
      const float t[12] = {o[a][0], o[1][0], o[b][0], o[a][1], o[1][1], o[b][1], o[a][2], o[1][2], o[b][2], o[a][3], o[1][3], o[b][3]};
      uint32_t d0, d1, d2;
      pack12_trunc(t, d0, d1, d2);
      stg3<false>(out, d0, d1, d2);
    } else {
      for (uint32_t i = 0; i < nv; i++) {
        out[3 * i] = (uint8_t)(uint32_t)(o[a][i]); out[3 * i + 1] = (uint8_t)(uint32_t)(o[1][i]); out[3 * i + 2] = (uint8_t)(uint32_t)(o[b][i]);
      }
    }
  }
}
template <int SRC, int DST, int IT, class BA = BatchArgs>
__global__ __launch_bounds__(256) void k_convert_resize_lds(const BA args, const Yuv2RgbCoef c, uint32_t sw, uint32_t sh,
                                                            uint32_t dw, uint32_t dh, float scx, float scy, int vec_ok, uint32_t rowq) {
  const BlockId b = picture_order();  // XCD-aware numbering (k_resize_common.h)
  convert_resize_lds_task<SRC, DST, IT>(args.f[b.z], c, sw, sh, dw, dh, scx, scy, vec_ok, rowq, b.x, b.y);
}
// single-frame entry: scalar arguments, what the first loads need in front (see VPF_ONE_SRC_PARAMS in vpf_internal.h)
template <int SRC, int DST, int IT>
__global__ __launch_bounds__(256) void k_convert_resize_lds_one(const uint8_t* s0, const uint8_t* s1, uint32_t sp0, uint32_t sp1, uint32_t sw,
                                                                uint32_t sh, uint32_t dw, uint32_t dh, float scx, float scy, uint32_t rowq,
                                                                int vec_ok, const uint8_t* s2, uint32_t sp2, VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  convert_resize_lds_task<SRC, DST, IT>(VPF_ONE_FRAME, c, sw, sh, dw, dh, scx, scy, vec_ok, rowq, blockIdx.x, blockIdx.y);
}


// ------------------------------------------------------------------------------------------
// The per-tap kernel as a ROW BAND (round 6): factors beyond ~2x (4K -> 1600 x 900, 1080p -> 800 x 450: every destination row has source rows
// of its own, four conversions per destination pixel whatever the kernel) spent a quarter of their ~100 VALU instructions per pixel on work
// that does not depend on the row — make_tap and the tap offsets of the four columns of a lane (17 per pixel) — and on a horizontal lerp
// evaluated row by row in scalar fp32.  Here a wave walks down R destination rows of its 256 columns: the column taps are computed once, the
// NEXT row's four (six) source strips are requested before this row is converted (the wave hides its own memory latency and pays its fixed
// part once per R rows), and everything runs on PAIRS ACROSS THE TWO SOURCE ROWS ({upper row, lower row} of one tap: v_pk_fma_f32), so the
// horizontal lerp of both rows is one v_pk_add_f32 + one v_pk_fma_f32 per channel.  Every component goes through the operations of
// convert_resize_lds_task in its order (fma(0, finite, p) == p covers the right picture edge): the same bytes.  No zero-weight shortcuts:
// odd integer factors keep convert_resize_lds_task, whose shortcuts skip whole rows and taps.
// ------------------------------------------------------------------------------------------
template <int SRC, int DST, int IT, int R>
VPF_DEV void convert_resize_band_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, float scx,
                                      float scy, int vec_ok, uint32_t rowq, uint32_t bx, uint32_t by) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t y0 = (by * 4 + wv) * R;
  if (y0 >= dh || bx * 256 >= dw) return;
  const uint32_t xs = bx * 256, xe = (xs + 255 < dw - 1) ? xs + 255 : dw - 1;
  const uint32_t first = make_tap<VPF_INTERP_LINEAR>(xs, scx, sw).i0, last = make_tap<VPF_INTERP_LINEAR>(xe, scx, sw).i1;
  const uint32_t ybase = first & ~15u, ynq = (last + 1 - ybase + 15) / 16;
  uint32_t cbase, cnq;
  if constexpr (SRC == FC_NV12) {
    cbase = (2 * (first >> 1)) & ~15u; cnq = (2 * (last >> 1) + 2 - cbase + 15) / 16;
  } else {
    cbase = (first >> 1) & ~15u; cnq = ((last >> 1) + 1 - cbase + 15) / 16;
  }
  constexpr int NS = (SRC == FC_NV12) ? 4 : 6;
  u32x4* const wstrip = dyn_strip + wv * NS * rowq;
  auto strip_at = [&](int k) { return reinterpret_cast<const uint8_t*>(wstrip + k * rowq); };
  // ---- the column side, once for the R rows
  const uint32_t x0 = xs + lane * 4;
  const bool draws = x0 < dw;
  const uint32_t nv = !draws ? 0u : dw - x0 < 4 ? dw - x0 : 4;
  uint32_t l0[4], l1[4], a0[4], a1[4];
  float fx[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const Tap tx = make_tap<VPF_INTERP_LINEAR>((x0 + k < dw) ? x0 + k : dw - 1, scx, sw);
    l0[k] = tx.i0 - ybase; l1[k] = tx.i1 - ybase; fx[k] = tx.f;
    a0[k] = (SRC == FC_NV12 ? (tx.i0 & ~1u) : (tx.i0 >> 1)) - cbase; a1[k] = (SRC == FC_NV12 ? (tx.i1 & ~1u) : (tx.i1 >> 1)) - cbase;
  }
  // ---- the row side: strips of one destination row requested (all loads in flight together), then written to LDS
  Span<IT> sp_[NS];
  uint32_t r_i0 = 0, r_i1 = 0;
  float r_f = 0.f;
  bool row1 = false, one_crow = true;
  auto request = [&](uint32_t y) {
    const Tap ty = make_tap<VPF_INTERP_LINEAR>(y, scy, sh);
    r_i0 = __builtin_amdgcn_readfirstlane(ty.i0); r_i1 = __builtin_amdgcn_readfirstlane(ty.i1);
    r_f = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ty.f)));
    row1 = r_f != 0.f;
    one_crow = !row1 || (r_i0 >> 1) == (r_i1 >> 1);
    sp_[0].load(f.s[0] + (size_t)r_i0 * f.sp[0], ybase, ynq, lane);
    if (row1) sp_[1].load(f.s[0] + (size_t)r_i1 * f.sp[0], ybase, ynq, lane);
    sp_[2].load(f.s[1] + (size_t)(r_i0 >> 1) * f.sp[1], cbase, cnq, lane);
    if (!one_crow) sp_[3].load(f.s[1] + (size_t)(r_i1 >> 1) * f.sp[1], cbase, cnq, lane);
    if constexpr (SRC != FC_NV12) {
      sp_[4].load(f.s[2] + (size_t)(r_i0 >> 1) * f.sp[2], cbase, cnq, lane);
      if (!one_crow) sp_[5].load(f.s[2] + (size_t)(r_i1 >> 1) * f.sp[2], cbase, cnq, lane);
    }
  };
  auto commit = [&]() {
    sp_[0].store(wstrip, ynq, lane);
    if (row1) sp_[1].store(wstrip + rowq, ynq, lane);
    sp_[2].store(wstrip + 2 * rowq, cnq, lane);
    if (!one_crow) sp_[3].store(wstrip + 3 * rowq, cnq, lane);
    if constexpr (SRC != FC_NV12) {
      sp_[4].store(wstrip + 4 * rowq, cnq, lane);
      if (!one_crow) sp_[5].store(wstrip + 5 * rowq, cnq, lane);
    }
    wave_lds_sync();
  };
  const f32x2 cy2 = {c.cy, c.cy}, rv2 = {c.rv, c.rv}, gu2 = {c.gu, c.gu}, gv2 = {c.gv, c.gv}, bu2 = {c.bu, c.bu};
  const f32x2 br2 = {c.br, c.br}, bg2 = {c.bg, c.bg}, bb2 = {c.bb, c.bb};
  auto rnd = [](f32x2 t) { return f32x2{(float)sat_rne(t[0]), (float)sat_rne(t[1])}; };
  request(y0);
#pragma unroll 1
  for (uint32_t r = 0; r < (uint32_t)R; r++) {
    const uint32_t y = y0 + r;
    commit();
    const bool row1_k = row1, one_crow_k = one_crow;
    const float fy = r_f;
    const bool more = r + 1 < (uint32_t)R && y + 1 < dh;
    if (more) request(y + 1);  // in flight while this row is converted
    if (draws) {
      const int lr = row1_k ? 1 : 0, cr = one_crow_k ? 0 : 1;  // the strip that stands for the lower source row (the upper one itself where there is none: weight 0 / never blended)
      const uint8_t* const y_up = strip_at(0);
      const uint8_t* const y_lo = strip_at(lr);
      float o[3][4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // chroma terms of tap t on {upper, lower} source row
        f32x2 rc[2], gc[2], bc[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const uint32_t a = t ? a1[k] : a0[k];
          f32x2 u, v;
          if constexpr (SRC == FC_NV12) {
            const uint32_t du = *reinterpret_cast<const uint16_t*>(strip_at(2) + a), dl = *reinterpret_cast<const uint16_t*>(strip_at(2 + cr) + a);
            u = f32x2{ubyte<0>(du), ubyte<0>(dl)}; v = f32x2{ubyte<1>(du), ubyte<1>(dl)};
          } else {
            u = f32x2{(float)strip_at(2)[a], (float)strip_at(2 + cr)[a]}; v = f32x2{(float)strip_at(4)[a], (float)strip_at(4 + cr)[a]};
          }
          rc[t] = __builtin_elementwise_fma(v, rv2, br2);
          gc[t] = __builtin_elementwise_fma(u, gu2, __builtin_elementwise_fma(v, gv2, bg2));
          bc[t] = __builtin_elementwise_fma(u, bu2, bb2);
        }
        const f32x2 yv0 = {(float)y_up[l0[k]], (float)y_lo[l0[k]]}, yv1 = {(float)y_up[l1[k]], (float)y_lo[l1[k]]};
        const f32x2 fx2 = {fx[k], fx[k]};
        const f32x2 r0 = rnd(__builtin_elementwise_fma(yv0, cy2, rc[0])), r1 = rnd(__builtin_elementwise_fma(yv1, cy2, rc[1]));
        const f32x2 g0 = rnd(__builtin_elementwise_fma(yv0, cy2, gc[0])), g1 = rnd(__builtin_elementwise_fma(yv1, cy2, gc[1]));
        const f32x2 b0 = rnd(__builtin_elementwise_fma(yv0, cy2, bc[0])), b1 = rnd(__builtin_elementwise_fma(yv1, cy2, bc[1]));
        const f32x2 hr = __builtin_elementwise_fma(fx2, r1 - r0, r0), hg = __builtin_elementwise_fma(fx2, g1 - g0, g0), hb = __builtin_elementwise_fma(fx2, b1 - b0, b0);  // {top, bot}
        if (row1_k) {
          o[0][k] = __builtin_fmaf(fy, hr[1] - hr[0], hr[0]) + 0.5f; o[1][k] = __builtin_fmaf(fy, hg[1] - hg[0], hg[0]) + 0.5f; o[2][k] = __builtin_fmaf(fy, hb[1] - hb[0], hb[0]) + 0.5f;
        } else {
          o[0][k] = hr[0] + 0.5f; o[1][k] = hg[0] + 0.5f; o[2][k] = hb[0] + 0.5f;
        }
      }
      if constexpr (DST == FC_PLANAR) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          uint8_t* out = f.d[ch] + (size_t)y * f.dp[ch] + x0;
          if (vec_ok && nv == 4) stg<false, uint32_t>(out, pack4_trunc(o[ch][0], o[ch][1], o[ch][2], o[ch][3]));
          else for (uint32_t i = 0; i < nv; i++) out[i] = (uint8_t)(uint32_t)(o[ch][i]);
        }
      } else {
        constexpr int a = (DST == FC_BGR) ? 2 : 0, b = (DST == FC_BGR) ? 0 : 2;
        uint8_t* out = f.d[0] + (size_t)y * f.dp[0] + 3 * (size_t)x0;
        if (vec_ok && nv == 4) {
          const float t[12] = {o[a][0], o[1][0], o[b][0], o[a][1], o[1][1], o[b][1], o[a][2], o[1][2], o[b][2], o[a][3], o[1][3], o[b][3]};
          uint32_t d0, d1, d2;
          pack12_trunc(t, d0, d1, d2);
          stg3<false>(out, d0, d1, d2);
        } else {
          for (uint32_t i = 0; i < nv; i++) {
            out[3 * i] = (uint8_t)(uint32_t)(o[a][i]); out[3 * i + 1] = (uint8_t)(uint32_t)(o[1][i]); out[3 * i + 2] = (uint8_t)(uint32_t)(o[b][i]);
          }
        }
      }
    }
    if (!more) break;
    wave_lds_sync();  // this row's LDS reads are done before the next row's strips overwrite them
  }
}
template <int SRC, int DST, int IT, int R, class BA = BatchArgs>
__global__ __launch_bounds__(256) void k_convert_resize_band(const BA args, const Yuv2RgbCoef c, uint32_t sw, uint32_t sh,
                                                             uint32_t dw, uint32_t dh, float scx, float scy, int vec_ok, uint32_t rowq) {
  const BlockId b = picture_order();  // XCD-aware numbering (k_resize_common.h)
  convert_resize_band_task<SRC, DST, IT, R>(args.f[b.z], c, sw, sh, dw, dh, scx, scy, vec_ok, rowq, b.x, b.y);
}

// ------------------------------------------------------------------------------------------
// Exact 2x down-scale (4K -> 1080p, 1080p -> 540p ...): s = (d + 0.5) * 2 - 0.5 = 2d + 0.5 exactly, so every destination
// pixel is the bilerp with fx = fy = 0.5 of the 2 x 2 block (2x..2x+1, 2y..2y+1), which shares ONE chroma sample.  That
// structure needs no tap arithmetic, no LDS gathers and one chroma evaluation per destination pixel: a lane converts
// 16 x 2 source pixels exactly like the NV12 -> RGB kernels (dwordx4 loads, convert4) and averages them; ~70 VALU per
// destination pixel instead of ~120 in the general kernel.  Bit-identical to it (same fma order on the same values).
// Requires NV12 or YUV420 (chroma planes 8-B aligned), sw == 2 dw, sh == 2 dh, sw % 16 == 0 (% 32 for packed outputs), 16-B aligned source rows, 8-B (planar) / 16-B
// (packed) aligned destination rows.
// ------------------------------------------------------------------------------------------
template <int DST, int SRC>
VPF_DEV void convert_half_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t sw, uint32_t dh, uint32_t chunks_x, uint32_t n_tasks) {
  __shared__ u32x4 tile[DST == FC_PLANAR ? 1 : 4 * 96];  // 1.5 KiB per wave: 64 lanes x 24 packed bytes
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = blockIdx.x * 4 + wv;
  if (wt >= n_tasks) return;
  const uint32_t y = wt / chunks_x, chunk = wt - y * chunks_x;
  const uint32_t xs = chunk * 1024 + lane * 16;  // first source pixel of the lane; destination pixel xs / 2
  const bool act = xs < sw;
  // With fx = fy = 0.5 the blend of four 8-bit taps is (p00 + p01 + p10 + p11 + 2) >> 2: bilerp()'s float chain is exact
  // on these operands (halves and quarters of small integers) and truncates the same quotient.  So the four converted
  // taps are rounded straight into the bytes of one dword (v_cvt_pk_u8_f32, vpf_convert's rounding) and ONE
  // v_dot4_u32_u8 with weights 64 and addend 128 leaves the pixel in byte 1: 64 * (sum + 2) >> 8.  ~40 instead of ~77
  // VALU per destination pixel — the kernel was VALU-bound (tools/gpu_pmc_fused.sh 3840 2160 1920 1080).
  uint32_t o[3][8];  // channel value of destination pixel i in byte 1
  if (act) {
    const u32x4 ya = ldg<true, u32x4>(f.s[0] + (size_t)(2 * y) * f.sp[0] + xs);
    const u32x4 yb = ldg<true, u32x4>(f.s[0] + (size_t)(2 * y + 1) * f.sp[0] + xs);
    const u32x4 uv = load_uv16<SRC, true>(f, y, xs);
    auto avg = [](const float* t, const float* b, int i) { return __builtin_amdgcn_udot4(pack4<1>(t[i], t[i + 1], b[i], b[i + 1]), 0x40404040u, 128u, false); };
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const Chroma k0 = chroma_terms(c, ubyte<0>(uv[j]), ubyte<1>(uv[j])), k1 = chroma_terms(c, ubyte<2>(uv[j]), ubyte<3>(uv[j]));
      const Quad qa = convert4(c, ya[j], k0, k1), qb = convert4(c, yb[j], k0, k1);
#pragma unroll
      for (int e = 0; e < 2; e++) {
        o[0][2 * j + e] = avg(qa.r, qb.r, 2 * e); o[1][2 * j + e] = avg(qa.g, qb.g, 2 * e); o[2][2 * j + e] = avg(qa.b, qb.b, 2 * e);
      }
    }
  }
  // byte 1 of four registers -> one dword (three v_perm_b32)
  auto gather4 = [](uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    return __builtin_amdgcn_perm(__builtin_amdgcn_perm(v3, v2, 0x0c0c0501u), __builtin_amdgcn_perm(v1, v0, 0x0c0c0501u), 0x05040100u);
  };
  const uint32_t xd = xs >> 1;
  if constexpr (DST == FC_PLANAR) {
    if (!act) return;
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
      stg<true, u32x2>(f.d[ch] + (size_t)y * f.dp[ch] + xd, u32x2{gather4(o[ch][0], o[ch][1], o[ch][2], o[ch][3]), gather4(o[ch][4], o[ch][5], o[ch][6], o[ch][7])});
  } else {
    constexpr int a = (DST == FC_BGR) ? 2 : 0, b = (DST == FC_BGR) ? 0 : 2;
    uint32_t* t = reinterpret_cast<uint32_t*>(tile + wv * 96);
    if (act) {
#pragma unroll
      for (int g = 0; g < 2; g++) {  // 4 px -> 3 dwords, twice
        const int q = 4 * g;
        t[lane * 6 + 3 * g] = gather4(o[a][q], o[1][q], o[b][q], o[a][q + 1]);
        t[lane * 6 + 3 * g + 1] = gather4(o[1][q + 1], o[b][q + 1], o[a][q + 2], o[1][q + 2]);
        t[lane * 6 + 3 * g + 2] = gather4(o[b][q + 2], o[a][q + 3], o[1][q + 3], o[b][q + 3]);
      }
    }
    wave_lds_sync();
    uint8_t* row = f.d[0] + (size_t)y * f.dp[0];
    const uint32_t row_bytes = 3 * (sw >> 1);
#pragma unroll
    for (int k = 0; k < 2; k++) {  // the wave's 1536 B leave as one dense 1-KiB store and one 512-B store
      const uint32_t idx = k * 64 + lane, off = chunk * 1536 + idx * 16;
      if (idx < 96 && off < row_bytes) stg<true, u32x4>(row + off, (tile + wv * 96)[idx]);
    }
  }
}
template <int DST, int SRC, class BA = BatchArgs>
__global__ __launch_bounds__(256) void k_convert_half(const BA args, const Yuv2RgbCoef c, uint32_t sw, uint32_t dh,
                                                      uint32_t chunks_x, uint32_t n_tasks) {
  convert_half_task<DST, SRC>(args.f[blockIdx.y], c, sw, dh, chunks_x, n_tasks);
}
template <int DST, int SRC>  // single-frame entry: scalar arguments
__global__ __launch_bounds__(256) void k_convert_half_one(VPF_ONE_SRC_PARAMS, uint32_t sw, uint32_t dh, uint32_t chunks_x, uint32_t n_tasks,
                                                          VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  convert_half_task<DST, SRC>(VPF_ONE_FRAME, c, sw, dh, chunks_x, n_tasks);
}

// ------------------------------------------------------------------------------------------
// General scale factors (1080p -> 720p ...): convert ONCE, blend from bytes.  k_convert_resize_lds converts the four taps of every
// destination pixel — 4 conversions (incl. the 8-bit rounding that keeps the result identical to convert-then-resize) per output pixel,
// 481 VALU instructions per 256-px wave at 1.5x, VALU-bound at 0.27 of the HBM roofline.  Here a wave owns R destination rows x 256
// columns: it first converts the source window those rows touch ((R - 1) scy + 2 rows x (255 scx + 2) pixels: 1.6 source pixels per
// destination pixel at 1.5x with R = 4, instead of 4) into a wave-private LDS strip of packed 8-bit RGB — straight from global
// memory, 8 pixels per lane, every row's loads in flight before the first is converted, one chroma evaluation per chroma sample —
// and then runs the plain row-pair bilinear blend (rowpair_blend4, packed fp32) on those bytes.  Same conversions, same rounding, same
// blend -> bit-identical to the two-step chain and to the other fused kernels.
// Requires 8-B aligned source planes, sw % 8 == 0, and a window that fits 8 LDS rows; odd-integer factors keep k_convert_resize_lds
// (whose centre-sample shortcut is HBM-bound already), exact 2x keeps k_convert_half.
// ------------------------------------------------------------------------------------------
constexpr int kStripRows = 8;  // source rows a wave's strip can hold
#ifdef VPF_LAB_FORMS  // superseded by k_convert_strip_wg (round 5): built into tools/lab/libvpfhip_forms.so only (VPF_TUNE_NV12_RGB_VARIANT = 47)
template <int SRC, int DST, int R>
VPF_DEV void convert_strip_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, float scx, float scy,
                                int vec_ok, uint32_t rowq_rows, uint32_t bx, uint32_t by) {
  const uint32_t rowq = rowq_rows & 0xffffu, srows = rowq_rows >> 16;  // 16-B units per strip row | strip rows per wave (<= kStripRows: what this launch's bands touch)
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t ya = (by * 4 + wv) * R;
  const uint32_t xs = bx * 256;
  if (ya >= dh || xs >= dw) return;
  const uint32_t yb = (ya + R - 1 < dh - 1) ? ya + R - 1 : dh - 1, xe = (xs + 255 < dw - 1) ? xs + 255 : dw - 1;
  const uint32_t first = make_tap<VPF_INTERP_LINEAR>(xs, scx, sw).i0, last = make_tap<VPF_INTERP_LINEAR>(xe, scx, sw).i1;
  const uint32_t base_px = first & ~7u;
  const uint32_t r_lo = make_tap<VPF_INTERP_LINEAR>(ya, scy, sh).i0, r_hi = make_tap<VPF_INTERP_LINEAR>(yb, scy, sh).i1;  // the launcher guarantees r_hi - r_lo < kStripRows
  uint8_t* const strip = reinterpret_cast<uint8_t*>(dyn_strip + (size_t)wv * srows * rowq);
  const uint32_t rowbytes = rowq * 16;
  constexpr int CMAX = kStripRows / 2 + 1;  // chroma rows under kStripRows luma rows
  const uint32_t c_lo = r_lo >> 1;
  for (uint32_t px0 = base_px + lane * 8; px0 <= last; px0 += 512) {  // one trip unless scx > 2
    u32x2 yq[CMAX][2], cq[CMAX];
    uint32_t vq[CMAX];  // YUV420: the V bytes (cq holds U then)
#pragma unroll
    for (int ci = 0; ci < CMAX; ci++) {
      const uint32_t crow = c_lo + ci;
      if (2 * crow > r_hi) break;
      if constexpr (SRC == FC_NV12) {
        cq[ci] = ldg<false, u32x2>(f.s[1] + (size_t)crow * f.sp[1] + px0);
      } else {
        cq[ci] = u32x2{ldg<false, uint32_t>(f.s[1] + (size_t)crow * f.sp[1] + (px0 >> 1)), 0u};
        vq[ci] = ldg<false, uint32_t>(f.s[2] + (size_t)crow * f.sp[2] + (px0 >> 1));
      }
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const uint32_t rr = 2 * crow + hf;
        if (rr >= r_lo && rr <= r_hi) yq[ci][hf] = ldg<false, u32x2>(f.s[0] + (size_t)rr * f.sp[0] + px0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // all of the window's loads are requested before the first conversion
#pragma unroll
    for (int ci = 0; ci < CMAX; ci++) {
      const uint32_t crow = c_lo + ci;
      if (2 * crow > r_hi) break;
      uint32_t uv[2];  // U V U V bytes of pixel pairs 0, 1 | 2, 3
      if constexpr (SRC == FC_NV12) {
        uv[0] = cq[ci][0]; uv[1] = cq[ci][1];
      } else {
        uv[0] = __builtin_amdgcn_perm(vq[ci], cq[ci][0], 0x05010400u); uv[1] = __builtin_amdgcn_perm(vq[ci], cq[ci][0], 0x07030602u);
      }
      Chroma k[4];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        k[2 * j] = chroma_terms(c, ubyte<0>(uv[j]), ubyte<1>(uv[j]));
        k[2 * j + 1] = chroma_terms(c, ubyte<2>(uv[j]), ubyte<3>(uv[j]));
      }
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const uint32_t rr = 2 * crow + hf;
        if (rr < r_lo || rr > r_hi) continue;
        uint32_t d[6];  // 8 px -> 24 bytes R G B R G B ..., vpf_convert's rounding (v_cvt_pk_u8_f32)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          // two pixels share a chroma sample: their three channel fmas run as pixel pairs on the packed-fp32 pipe (same IEEE fma per
          // component as convert4 -> same bits)
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const uint32_t yd = yq[ci][hf][j];
          const f32x2 cy2 = {c.cy, c.cy}, ya2 = {ubyte<0>(yd), ubyte<1>(yd)}, yb2 = {ubyte<2>(yd), ubyte<3>(yd)};
          const Chroma &ka = k[2 * j], &kb = k[2 * j + 1];
          const f32x2 ra = __builtin_elementwise_fma(ya2, cy2, f32x2{ka.rc, ka.rc}), ga = __builtin_elementwise_fma(ya2, cy2, f32x2{ka.gc, ka.gc}),
                      ba = __builtin_elementwise_fma(ya2, cy2, f32x2{ka.bc, ka.bc});
          const f32x2 rb = __builtin_elementwise_fma(yb2, cy2, f32x2{kb.rc, kb.rc}), gb = __builtin_elementwise_fma(yb2, cy2, f32x2{kb.gc, kb.gc}),
                      bb = __builtin_elementwise_fma(yb2, cy2, f32x2{kb.bc, kb.bc});
          d[3 * j] = pack4<1>(ra[0], ga[0], ba[0], ra[1]);
          d[3 * j + 1] = pack4<1>(ga[1], ba[1], rb[0], gb[0]);
          d[3 * j + 2] = pack4<1>(bb[0], rb[1], gb[1], bb[1]);
        }
        u32x2* w = reinterpret_cast<u32x2*>(strip + (size_t)(rr - r_lo) * rowbytes + 3 * (px0 - base_px));
        w[0] = u32x2{d[0], d[1]}; w[1] = u32x2{d[2], d[3]}; w[2] = u32x2{d[4], d[5]};
      }
    }
  }
  wave_lds_sync();
  const Tap row_taps = band_row_taps(ya, yb, scy, sh);  // every lane still active here
  const uint32_t x0 = xs + lane * 4;
  if (x0 >= dw) return;
  const uint32_t nv = dw - x0 < 4 ? dw - x0 : 4;
  const ColTaps<3> T = make_col_taps<3>(3 * base_px, x0, dw, sw, scx);  // once for the R rows
  // the band walk of RowBandTask: every strip row's horizontal lerp is evaluated once and shared by the destination rows that blend it
  band_blend_rows<3, R>(strip, rowbytes, r_lo, ya, yb, row_taps, T, [&](uint32_t y, const float* o) {  // o: pixel-major R G B, + 0.5 added
    if constexpr (DST == FC_PLANAR) {
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        uint8_t* out = f.d[ch] + (size_t)y * f.dp[ch] + x0;
        if (vec_ok && nv == 4) stg<false, uint32_t>(out, pack4_trunc(o[ch], o[3 + ch], o[6 + ch], o[9 + ch]));
        else for (uint32_t j = 0; j < nv; j++) out[j] = (uint8_t)(uint32_t)o[3 * j + ch];
      }
    } else {
      constexpr int a = (DST == FC_BGR) ? 2 : 0, b = (DST == FC_BGR) ? 0 : 2;
      uint8_t* out = f.d[0] + (size_t)y * f.dp[0] + 3 * (size_t)x0;
      if (vec_ok && nv == 4) {
        const float t[12] = {o[a], o[1], o[b], o[3 + a], o[4], o[3 + b], o[6 + a], o[7], o[6 + b], o[9 + a], o[10], o[9 + b]};
        uint32_t d0, d1, d2;
        pack12_trunc(t, d0, d1, d2);
        stg3<false>(out, d0, d1, d2);
      } else {
        for (uint32_t j = 0; j < nv; j++) {
          out[3 * j] = (uint8_t)(uint32_t)o[3 * j + a]; out[3 * j + 1] = (uint8_t)(uint32_t)o[3 * j + 1]; out[3 * j + 2] = (uint8_t)(uint32_t)o[3 * j + b];
        }
      }
    }
  });
}
template <int SRC, int DST, int R, class BA = BatchArgs>
__global__ __launch_bounds__(256) void k_convert_strip(const BA args, const Yuv2RgbCoef c, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh,
                                                       float scx, float scy, int vec_ok, uint32_t rowq) {
  VPF_WAVE_TIMER(4);
  const BlockId b = picture_order();  // XCD-aware numbering (k_resize_common.h): bands above each other and chunks next to each other share one L2
  convert_strip_task<SRC, DST, R>(args.f[b.z], c, sw, sh, dw, dh, scx, scy, vec_ok, rowq, b.x, b.y);
}
#endif  // VPF_LAB_FORMS

// ------------------------------------------------------------------------------------------
// The same, with the strip shared by the WORKGROUP (round 5).  In k_convert_strip every wave converts its own window with 8 pixels per lane
// over the window's columns: 49 of 64 lanes at 1.5x (386 px), 17 of 64 at a 2x up-scale (130 px), each of them walking all the window's
// rows — at 1080p -> 4K the conversions took as long as the blend and the kernel LOST to convert-then-resize (11.5 against 7.4 us per
// frame).  Here the four waves of a workgroup own four bands above each other (4 R destination rows x 256 columns), the source
// window of all of them is ONE strip, and its conversion is dealt out in UNITS of (chroma row, 8-pixel group) = 8 px x 2 luma rows over all
// 256 lanes: every source row is converted once per workgroup (the rows two neighbouring bands share were converted twice), a chroma sample's
// terms are evaluated once per strip, and the lanes are full whatever the window's width (2x up, R = 16: 306 units, 60 % of two trips;
// 1.5x down, R = 4: 637 units, 83 % of three trips).  One s_barrier between the conversion and the blend.  Same conversions, same
// rounding, same blend: the bytes of k_convert_strip, of convert-then-resize and of the oracle.
// ------------------------------------------------------------------------------------------
template <int SRC>
VPF_DEV void convert_unit8(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t crow, uint32_t px0, bool row_a, bool row_b, uint8_t* wa /* strip byte of (row 2 crow, px0) */,
                           uint32_t rowbytes, u32x2 ya, u32x2 yb, u32x2 cq, uint32_t vq) {
  (void)f; (void)crow; (void)px0;
  uint32_t uv[2];  // U V U V bytes of pixel pairs 0, 1 | 2, 3
  if constexpr (SRC == FC_NV12) {
    uv[0] = cq[0]; uv[1] = cq[1];
  } else {
    uv[0] = __builtin_amdgcn_perm(vq, cq[0], 0x05010400u); uv[1] = __builtin_amdgcn_perm(vq, cq[0], 0x07030602u);
  }
  Chroma k[4];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    k[2 * j] = chroma_terms(c, ubyte<0>(uv[j]), ubyte<1>(uv[j]));
    k[2 * j + 1] = chroma_terms(c, ubyte<2>(uv[j]), ubyte<3>(uv[j]));
  }
#pragma unroll
  for (int hf = 0; hf < 2; hf++) {
    if (!(hf ? row_b : row_a)) continue;
    const u32x2 yq = hf ? yb : ya;
    uint32_t d[8];  // 8 px -> 8 dwords R G B x (px4 strips, k_bilinear_blend.h), vpf_convert's rounding (v_cvt_pk_u8_f32): three cvt per pixel as for packed bytes
#pragma unroll
    for (int j = 0; j < 2; j++) {  // (the pixel-pair form of convert_strip_task: same IEEE fma per component as convert4 -> same bits)
      const uint32_t yd = yq[j];
      const f32x2 cy2 = {c.cy, c.cy}, ya2 = {ubyte<0>(yd), ubyte<1>(yd)}, yb2 = {ubyte<2>(yd), ubyte<3>(yd)};
      const Chroma &ka = k[2 * j], &kb = k[2 * j + 1];
      const f32x2 ra = __builtin_elementwise_fma(ya2, cy2, f32x2{ka.rc, ka.rc}), ga = __builtin_elementwise_fma(ya2, cy2, f32x2{ka.gc, ka.gc}),
                  ba = __builtin_elementwise_fma(ya2, cy2, f32x2{ka.bc, ka.bc});
      const f32x2 rb = __builtin_elementwise_fma(yb2, cy2, f32x2{kb.rc, kb.rc}), gb = __builtin_elementwise_fma(yb2, cy2, f32x2{kb.gc, kb.gc}),
                  bb = __builtin_elementwise_fma(yb2, cy2, f32x2{kb.bc, kb.bc});
      d[4 * j] = pack3(ra[0], ga[0], ba[0]);
      d[4 * j + 1] = pack3(ra[1], ga[1], ba[1]);
      d[4 * j + 2] = pack3(rb[0], gb[0], bb[0]);
      d[4 * j + 3] = pack3(rb[1], gb[1], bb[1]);
    }
    u32x4* w = reinterpret_cast<u32x4*>(wa + (hf ? rowbytes : 0u));
    w[0] = u32x4{d[0], d[1], d[2], d[3]}; w[1] = u32x4{d[4], d[5], d[6], d[7]};
  }
}
template <int SRC, int DST, int R>
VPF_DEV void convert_strip_wg_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, float scx, float scy,
                                   int vec_ok, uint32_t rowq, uint32_t bx, uint32_t by) {
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, tid = threadIdx.x;
  const uint32_t Y0 = by * (4 * R), xs = bx * 256;  // the grid covers the picture exactly: Y0 < dh, xs < dw
  const uint32_t Y1 = (Y0 + 4 * R - 1 < dh - 1) ? Y0 + 4 * R - 1 : dh - 1, xe = (xs + 255 < dw - 1) ? xs + 255 : dw - 1;
  const uint32_t first = make_tap<VPF_INTERP_LINEAR>(xs, scx, sw).i0, last = make_tap<VPF_INTERP_LINEAR>(xe, scx, sw).i1;
  const uint32_t base_px = first & ~7u;
  // the workgroup's source rows (the launcher sized the strip for them: vpf_band_rows_exact(4 R, ..))
  const uint32_t R_lo = __builtin_amdgcn_readfirstlane(make_tap<VPF_INTERP_LINEAR>(Y0, scy, sh).i0), R_hi = __builtin_amdgcn_readfirstlane(make_tap<VPF_INTERP_LINEAR>(Y1, scy, sh).i1);
  uint8_t* const strip = reinterpret_cast<uint8_t*>(dyn_strip);
  const uint32_t rowbytes = rowq * 16;
  const uint32_t c_lo = R_lo >> 1, ncr = (R_hi >> 1) - c_lo + 1, ng = ((last - base_px) >> 3) + 1, units = ncr * ng;
  const float rng = 1.0f / (float)ng;
  // unit u -> (chroma row ci = u / ng, group g): (u + 0.5) / ng is at least 0.5 / ng from an integer, far more than the fp32 error for u < 2^14
  struct Unit { uint32_t px0 = 0; bool act = false, ra = false, rb = false; uint8_t* w = nullptr; u32x2 ya = {0u, 0u}, yb = {0u, 0u}, cq = {0u, 0u}; uint32_t vq = 0, crow = 0; };
  auto fetch = [&](uint32_t u, Unit& q) {
    q.act = u < units;
    if (!q.act) return;
    const uint32_t ci = (uint32_t)(((float)u + 0.5f) * rng), g = u - ci * ng;
    q.crow = c_lo + ci; q.px0 = base_px + 8 * g;
    const uint32_t r0 = 2 * q.crow;
    q.ra = r0 >= R_lo; q.rb = r0 + 1 <= R_hi;  // (r0 <= R_hi and r0 + 1 >= R_lo hold for every chroma row of the window)
    if constexpr (SRC == FC_NV12) {
      q.cq = ldg<false, u32x2>(f.s[1] + (size_t)q.crow * f.sp[1] + q.px0);
    } else {
      q.cq = u32x2{ldg<false, uint32_t>(f.s[1] + (size_t)q.crow * f.sp[1] + (q.px0 >> 1)), 0u};
      q.vq = ldg<false, uint32_t>(f.s[2] + (size_t)q.crow * f.sp[2] + (q.px0 >> 1));
    }
    if (q.ra) q.ya = ldg<false, u32x2>(f.s[0] + (size_t)r0 * f.sp[0] + q.px0);
    if (q.rb) q.yb = ldg<false, u32x2>(f.s[0] + (size_t)(r0 + 1) * f.sp[0] + q.px0);
    // strip byte of (row r0, px0); r0 may be R_lo - 1 (that row is not written then: only row r0 + 1 is) — a signed offset
    q.w = strip + ((int32_t)(r0 - R_lo) * (int32_t)rowbytes + (int32_t)(4 * (q.px0 - base_px)));
  };
  for (uint32_t u0 = tid; u0 < units; u0 += 512) {  // two units per lane in flight
    Unit q0, q1;
    fetch(u0, q0);
    fetch(u0 + 256, q1);
    __builtin_amdgcn_sched_barrier(0);  // both units' loads are requested before the first conversion
    if (q0.act) convert_unit8<SRC>(f, c, q0.crow, q0.px0, q0.ra, q0.rb, q0.w, rowbytes, q0.ya, q0.yb, q0.cq, q0.vq);
    if (q1.act) convert_unit8<SRC>(f, c, q1.crow, q1.px0, q1.ra, q1.rb, q1.w, rowbytes, q1.ya, q1.yb, q1.cq, q1.vq);
  }
  __syncthreads();
  const uint32_t ya = Y0 + wv * R;
  if (ya > Y1) return;
  const uint32_t yb = (ya + R - 1 < Y1) ? ya + R - 1 : Y1;
  const Tap row_taps = band_row_taps(ya, yb, scy, sh);  // every lane of the wave still active here
  const uint32_t x0 = xs + lane * 4;
  if (x0 >= dw) return;
  const uint32_t nv = dw - x0 < 4 ? dw - x0 : 4;
  const ColTapsX T = make_col_taps_x(base_px, x0, dw, sw, scx);  // once for the R rows (px4 strip: a tap is one aligned dword)
  band_blend_rows<3, R>(strip, rowbytes, R_lo, ya, yb, row_taps, T, [&](uint32_t y, const float* o) {  // o: pixel-major R G B, + 0.5 added
    if constexpr (DST == FC_PLANAR) {
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        uint8_t* out = f.d[ch] + (size_t)y * f.dp[ch] + x0;
        if (vec_ok && nv == 4) stg<false, uint32_t>(out, pack4_trunc(o[ch], o[3 + ch], o[6 + ch], o[9 + ch]));
        else for (uint32_t j = 0; j < nv; j++) out[j] = (uint8_t)(uint32_t)o[3 * j + ch];
      }
    } else {
      constexpr int a = (DST == FC_BGR) ? 2 : 0, b = (DST == FC_BGR) ? 0 : 2;
      uint8_t* out = f.d[0] + (size_t)y * f.dp[0] + 3 * (size_t)x0;
      if (vec_ok && nv == 4) {
        const float t[12] = {o[a], o[1], o[b], o[3 + a], o[4], o[3 + b], o[6 + a], o[7], o[6 + b], o[9 + a], o[10], o[9 + b]};
        uint32_t d0, d1, d2;
        pack12_trunc(t, d0, d1, d2);
        stg3<false>(out, d0, d1, d2);
      } else {
        for (uint32_t j = 0; j < nv; j++) {
          out[3 * j] = (uint8_t)(uint32_t)o[3 * j + a]; out[3 * j + 1] = (uint8_t)(uint32_t)o[3 * j + 1]; out[3 * j + 2] = (uint8_t)(uint32_t)o[3 * j + b];
        }
      }
    }
  });
}
template <int SRC, int DST, int R, class BA = BatchArgs>
__global__ __launch_bounds__(256) void k_convert_strip_wg(const BA args, const Yuv2RgbCoef c, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh,
                                                          float scx, float scy, int vec_ok, uint32_t rowq) {
  VPF_WAVE_TIMER(5);
  const BlockId b = picture_order();
  convert_strip_wg_task<SRC, DST, R>(args.f[b.z], c, sw, sh, dw, dh, scx, scy, vec_ok, rowq, b.x, b.y);
}

hipError_t launch_convert_resize(hipStream_t st, int src_fc, int dst_fc, const Yuv2RgbCoef& c, uint32_t sw, uint32_t sh,
                                 uint32_t n, const BatchArgsL& a, uint32_t dw, uint32_t dh) {
  const float scx = (float)sw / (float)dw, scy = (float)sh / (float)dh;
  int vec_ok = 1;
  // up to 32 frames travel in the small frame table, more in the large one: two instantiations of every batch kernel (vpf_internal.h)
  const bool small = n <= (uint32_t)kSmallBatch;
  const BatchArgs as = small_batch(a, small ? n : 0u);
#define VPF_UNPAREN(...) __VA_ARGS__
#define VPF_LAUNCH_BA(K, TARGS, GRID, BLK, LDS, ST, ...) do { \
    if (small) VPF_LAUNCH((K<VPF_UNPAREN TARGS, BatchArgs>), GRID, BLK, LDS, ST, as, __VA_ARGS__); \
    else VPF_LAUNCH((K<VPF_UNPAREN TARGS, BatchArgsL>), GRID, BLK, LDS, ST, a, __VA_ARGS__); } while (0)
  const uint32_t rowb = lds_strip_bytes(1, sw, dw, a.f[0].s[0], a.f[0].sp[0], kFusedRowBytes);
  bool lds_ok = rowb != 0;
  for (uint32_t i = 0; i < n; i++) {
    const FrameDesc& f = a.f[i];
    for (int k = 0; k < (dst_fc == FC_PLANAR ? 3 : 1); k++) vec_ok &= ((((uintptr_t)f.d[k] | f.dp[k]) & 3) == 0);
    for (int k = 0; k < (src_fc == FC_NV12 ? 2 : 3); k++) lds_ok = lds_ok && !(((uintptr_t)f.s[k] | f.sp[k]) & 15);
  }
  // exact 2x from NV12: the quad-structured kernel (no taps, no gathers); tuning 40 / 9 keep the general kernels
  // (packed rows leave as 16-B stores: 3 * dw must be a multiple of 16)
  if ((src_fc == FC_NV12 || src_fc == FC_YUV420) && sw == 2 * dw && sh == 2 * dh && sw % (dst_fc == FC_PLANAR ? 16 : 32) == 0 && lds_ok && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 40) {
    bool ok16 = true;
    for (uint32_t i = 0; i < n; i++)
      for (int k = 0; k < (dst_fc == FC_PLANAR ? 3 : 1); k++) ok16 = ok16 && !(((uintptr_t)a.f[i].d[k] | a.f[i].dp[k]) & (dst_fc == FC_PLANAR ? 7 : 15));
    if (ok16) {
      const uint32_t chunks = (sw + 1023) / 1024, tasks = chunks * dh;
      dim3 hgrid((tasks + 3) / 4, n);
#define VPF_HALF1(D, S) do { if (n == 1) VPF_LAUNCH((k_convert_half_one<D, S>), hgrid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), sw, dh, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c); \
                            else VPF_LAUNCH_BA(k_convert_half, (D, S), hgrid, dim3(256), 0, st, c, sw, dh, chunks, tasks); } while (0)
#define VPF_HALF(S) do { if (dst_fc == FC_RGB) VPF_HALF1(FC_RGB, S); else if (dst_fc == FC_BGR) VPF_HALF1(FC_BGR, S); else VPF_HALF1(FC_PLANAR, S); } while (0)
      if (src_fc == FC_NV12) VPF_HALF(FC_NV12); else VPF_HALF(FC_YUV420);
#undef VPF_HALF
#undef VPF_HALF1
      return hipGetLastError();
    }
  }
  // general factors: convert the window once into an LDS RGB strip, blend from bytes (k_convert_strip); odd integer factors on either
  // axis keep the kernels below (their zero-weight shortcuts skip whole rows / taps)
  {
    const bool odd_x = sw % dw == 0 && ((sw / dw) & 1), odd_y = sh % dh == 0 && ((sh / dh) & 1);
    bool ok8 = (src_fc == FC_NV12 || src_fc == FC_YUV420) && sw % 8 == 0 && !odd_x && !odd_y && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 40 &&
               tuning(VPF_TUNE_NV12_RGB_VARIANT) != 9 && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 49 && sw < (1u << 22) && sh < (1u << 22);
    for (uint32_t i = 0; i < n && ok8; i++)
      for (int k = 0; k < (src_fc == FC_NV12 ? 2 : 3); k++) ok8 = ok8 && !(((uintptr_t)a.f[i].s[k] | a.f[i].sp[k]) & 7);
    if (ok8) {
#ifdef VPF_LAB_FORMS  // the per-wave strips of rounds 2-4 (k_convert_strip): variant 47, and where the workgroup strips do not apply
      const uint32_t rowbytes = vpf_bound_fused_rowbytes(scx);  // a wave's source span + alignment + tap-window slack (vpf_plan_bounds.h)
      int r = 0;
      if (vpf_bound_fused_rows_fit(8, scy, kStripRows)) r = 8;  // rows a wave's R destination rows can touch: <= (R - 1) scy + 3 (+ fp32 slack)
      else if (vpf_bound_fused_rows_fit(4, scy, kStripRows)) r = 4;
      else if (vpf_bound_fused_rows_fit(2, scy, kStripRows)) r = 2;
      if (dh < 64) r = r ? 2 : 0;  // short pictures: more, smaller tasks
      if (r == 8 && (uint64_t)((dw + 255) / 256) * ((dh + 31) / 32) * n < 2048) r = 4;  // keep the chip covered
      // strip rows per wave: what the bands of this shape really touch (a walk with the kernel's tap arithmetic, vpf_plan_bounds.h), not
      // the kStripRows the band height was chosen against — 1080p -> 720p: 6 rows, five workgroups per CU instead of four
      static thread_local struct { int r; uint32_t sh, dh, rows; } seen = {0, 0, 0, 0};
      if (r && !(seen.r == r && seen.sh == sh && seen.dh == dh)) { seen.r = r; seen.sh = sh; seen.dh = dh; seen.rows = vpf_band_rows_exact(r, sh, dh, scy); }
      const uint32_t srows = r ? (seen.rows < (uint32_t)kStripRows ? seen.rows : (uint32_t)kStripRows) : (uint32_t)kStripRows;
      const uint32_t lds1 = 4u * srows * rowbytes;
      // conversions per destination pixel: scx x ((r - 1) scy + 2) / r source pixels against the four taps of the per-tap kernel —
      // measured break-even near 2x (4K -> 1600x900, 2.4x: 9.5 us here vs 5.2 us per-tap; 1080p -> 720p: 2.36 vs 3.21; 1080p -> 4K: 15.0 vs 23.6)
      const double conv_per_px = r ? (double)scx * ((r - 1) * (double)scy + 2.0) / r : 1e9;
#endif  // VPF_LAB_FORMS
      // Round 5: the strip shared by the workgroup (k_convert_strip_wg: the source window of 4 R destination rows converted once, dealt
      // out over all 256 lanes).  Band height: the largest of 16 (up-scales) / 8 / 4 / 2 whose strip leaves four workgroups per CU
      // (<= 40 KiB of the CU's 160: round 6's four-byte pixels make 1080p -> 720p at R = 4 a 39.5-KiB strip) and whose launch still covers the chip.  (Lab builds: VPF_TUNE_NV12_RGB_VARIANT = 47 keeps the per-wave strips.)
#ifdef VPF_LAB_FORMS
      if (tuning(VPF_TUNE_NV12_RGB_VARIANT) != 47)
#endif
      {
        static thread_local struct { uint32_t sh, dh, rows[4]; } wseen = {0, 0, {0, 0, 0, 0}};
        if (!(wseen.sh == sh && wseen.dh == dh)) {
          wseen.sh = sh; wseen.dh = dh;
          for (int k = 0; k < 4; k++) wseen.rows[k] = vpf_band_rows_exact(4 * (2 << k), sh, dh, scy);  // the workgroup's 4 R rows: R = 2, 4, 8, 16
        }
        int rw = 0;
        uint32_t wrows = 0;
        const uint32_t rowbytes4 = vpf_bound_fused_rowbytes4(scx);  // the workgroup's strip holds four bytes per pixel (R G B x)
        for (int k = 3; k >= 0; k--) {
          const int cand = 2 << k;
          if (cand == 16 && scy > 1.0f) continue;
          if ((uint64_t)wseen.rows[k] * rowbytes4 > 40u * 1024u) continue;
          if (cand > 2 && (uint64_t)((dw + 255) / 256) * ((dh + 4 * cand - 1) / (4 * cand)) * n < (cand >= 8 ? 2048u : 512u)) continue;  // keep the chip covered
          rw = cand; wrows = wseen.rows[k];
          break;
        }
        // source pixels converted per destination pixel: rows of the strip x its width / (4 R x 256); the per-tap kernel converts four
        auto wconv_of = [&](int cand, uint32_t rows) { return cand ? (double)rows * ((double)scx * 255.0 + 18.0) / (4.0 * cand * 256.0) : 1e9; };
        const double wmax = tuning(VPF_TUNE_NV12_RGB_VARIANT) == 48 ? 8.0 : 3.0;
        // Second pass (round 6: the four-byte pixels pushed factors of 1.55-1.9 and small single frames out of the first): 4- or 2-row bands
        // with a strip of up to 53 KiB (three workgroups per CU) however few workgroups the launch has — the shapes the per-wave strips of
        // rounds 2-4 (lab builds: k_convert_strip) used to take, with the rows neighbouring bands share converted once.
        if (!rw || wconv_of(rw, wrows) > wmax) {
          rw = 0;
          for (int k = 1; k >= 0; k--) {
            const int cand = 2 << k;
            if ((uint64_t)wseen.rows[k] * rowbytes4 > 53u * 1024u || (dh < 64 && cand > 2)) continue;
            if (wconv_of(cand, wseen.rows[k]) <= wmax) { rw = cand; wrows = wseen.rows[k]; break; }
          }
        }
        const double wconv = wconv_of(rw, wrows);
        if (rw && wconv <= wmax) {
          const uint32_t ldsw = wrows * rowbytes4;
          dim3 wgrid((dw + 255) / 256, (dh + 4 * rw - 1) / (4 * rw), n);
#define VPF_WG1(S, D, RR) VPF_LAUNCH_BA(k_convert_strip_wg, (S, D, RR), wgrid, dim3(256), ldsw, st, c, sw, sh, dw, dh, scx, scy, vec_ok, rowbytes4 / 16)
#define VPF_WG(S, D) do { if (rw == 16) VPF_WG1(S, D, 16); else if (rw == 8) VPF_WG1(S, D, 8); else if (rw == 4) VPF_WG1(S, D, 4); else VPF_WG1(S, D, 2); } while (0)
#define VPF_WGD(S) do { if (dst_fc == FC_RGB) VPF_WG(S, FC_RGB); else if (dst_fc == FC_BGR) VPF_WG(S, FC_BGR); else VPF_WG(S, FC_PLANAR); } while (0)
          if (src_fc == FC_NV12) VPF_WGD(FC_NV12); else VPF_WGD(FC_YUV420);
#undef VPF_WGD
#undef VPF_WG
#undef VPF_WG1
          return hipGetLastError();
        }
      }
#ifdef VPF_LAB_FORMS
      if (r && lds1 <= 64u * 1024u && conv_per_px <= 3.0) {
        dim3 sgrid((dw + 255) / 256, (dh + 4 * r - 1) / (4 * r), n);
#define VPF_STRIP1(S, D, RR) VPF_LAUNCH_BA(k_convert_strip, (S, D, RR), sgrid, dim3(256), lds1, st, c, sw, sh, dw, dh, scx, scy, vec_ok, (rowbytes / 16) | (srows << 16))
#define VPF_STRIP(S, D) do { if (r == 8) VPF_STRIP1(S, D, 8); else if (r == 4) VPF_STRIP1(S, D, 4); else VPF_STRIP1(S, D, 2); } while (0)
#define VPF_STRIPD(S) do { if (dst_fc == FC_RGB) VPF_STRIP(S, FC_RGB); else if (dst_fc == FC_BGR) VPF_STRIP(S, FC_BGR); else VPF_STRIP(S, FC_PLANAR); } while (0)
        if (src_fc == FC_NV12) VPF_STRIPD(FC_NV12); else VPF_STRIPD(FC_YUV420);
#undef VPF_STRIPD
#undef VPF_STRIP
#undef VPF_STRIP1
        return hipGetLastError();
      }
#endif  // VPF_LAB_FORMS
    }
  }
  const uint32_t lds = 4 * (src_fc == FC_NV12 ? 4 : 6) * rowb;
  // factors beyond ~2x on a launch that covers the chip with four rows per wave: the per-tap kernel as a row band (k_convert_resize_band);
  // odd integer factors keep the row-at-a-time kernel below and its zero-weight shortcuts.  VPF_TUNE_NV12_RGB_VARIANT = 40 / 9 keep it too.
  {
    const bool odd_x = sw % dw == 0 && ((sw / dw) & 1), odd_y = sh % dh == 0 && ((sh / dh) & 1);
    const int v = tuning(VPF_TUNE_NV12_RGB_VARIANT);
    if ((src_fc == FC_NV12 || src_fc == FC_YUV420) && lds_ok && !odd_x && !odd_y && v != 40 && v != 9 && sw < (1u << 22) && sh < (1u << 22) &&
        ((uint64_t)((dw + 255) / 256) * ((dh + 15) / 16) * n >= 2048u || v == 49)) {  // 49: whatever the launch size (tests, A/B runs)
      dim3 bgrid((dw + 255) / 256, (dh + 15) / 16, n);
#define VPF_BAND1(S, D, I) VPF_LAUNCH_BA(k_convert_resize_band, (S, D, I, 4), bgrid, dim3(256), lds, st, c, sw, sh, dw, dh, scx, scy, vec_ok, rowb / 16)
#define VPF_BAND(S, D) do { if (rowb <= 1024) VPF_BAND1(S, D, 1); else VPF_BAND1(S, D, 2); } while (0)
#define VPF_BANDD(S) do { if (dst_fc == FC_RGB) VPF_BAND(S, FC_RGB); else if (dst_fc == FC_BGR) VPF_BAND(S, FC_BGR); else VPF_BAND(S, FC_PLANAR); } while (0)
      if (src_fc == FC_NV12) VPF_BANDD(FC_NV12); else VPF_BANDD(FC_YUV420);
#undef VPF_BANDD
#undef VPF_BAND
#undef VPF_BAND1
      return hipGetLastError();
    }
  }
  dim3 grid(((dw + 3) / 4 + 63) / 64, (dh + 3) / 4, n);
#define VPF_GOL1(S, D, I) do { if (n == 1) VPF_LAUNCH((k_convert_resize_lds_one<S, D, I>), grid, dim3(256), lds, st, a.f[0].s[0], a.f[0].s[1], a.f[0].sp[0], a.f[0].sp[1], \
                                                      sw, sh, dw, dh, scx, scy, rowb / 16, vec_ok, a.f[0].s[2], a.f[0].sp[2], VPF_ONE_DST_ARGS(a.f[0]), c); \
                               else VPF_LAUNCH_BA(k_convert_resize_lds, (S, D, I), grid, dim3(256), lds, st, c, sw, sh, dw, dh, scx, scy, vec_ok, rowb / 16); } while (0)
#define VPF_GOL(S, D) do { if (rowb <= 1024) VPF_GOL1(S, D, 1); else VPF_GOL1(S, D, 2); } while (0)
#define VPF_GO(S, D) VPF_LAUNCH_BA(k_convert_resize, (S, D), grid, dim3(256), 0, st, c, sw, sh, dw, dh, scx, scy, vec_ok)
#define VPF_PICK(S, D) do { if (lds_ok) VPF_GOL(S, D); else VPF_GO(S, D); } while (0)
  if (src_fc == FC_NV12) {
    if (dst_fc == FC_RGB) VPF_PICK(FC_NV12, FC_RGB); else if (dst_fc == FC_BGR) VPF_PICK(FC_NV12, FC_BGR); else VPF_PICK(FC_NV12, FC_PLANAR);
  } else if (src_fc == FC_YUV420) {
    if (dst_fc == FC_RGB) VPF_PICK(FC_YUV420, FC_RGB); else if (dst_fc == FC_BGR) VPF_PICK(FC_YUV420, FC_BGR); else VPF_PICK(FC_YUV420, FC_PLANAR);
  } else {
    return hipErrorInvalidValue;
  }
#undef VPF_PICK
#undef VPF_GO
#undef VPF_GOL
#undef VPF_GOL1
#undef VPF_LAUNCH_BA
#undef VPF_UNPAREN
  return hipGetLastError();
}

}  // namespace vpf
VPF_WAVE_TIMES_EXPORT(vpf_lab_wave_times_fused)
