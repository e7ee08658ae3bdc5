// k_remap.hip — per-pixel remap of packed RGB (gfx950).  Replaces nppiRemap_8u_C3R (reference: NppRemapSurfacePacked3C_Impl::Run,
// src/TC/src/Tasks.cpp:1555-1602; the reference asks for NPPI_INTER_LINEAR, :1590).
//   k_remap3_p4, k_remap3   4 px per lane with 12-B tap windows (all requested up front) / generic gather form; `_b`: one pair of maps
//                           over up to 32 frames per dispatch (vpf_remap_batch)
#include "k_bilinear_blend.h"

namespace vpf {

// ------------------------------------------------------------------------------------------
// remap: dst(x,y) = bilinear(src, xmap[y][x], ymap[y][x]); out-of-range -> dst untouched [A9].
// One lane per destination pixel: the map reads (8 B/px) are coalesced, texels are gathers.
// ------------------------------------------------------------------------------------------
VPF_DEV void remap3_task(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh, const float* __restrict__ xmap, uint32_t xp,
                         const float* __restrict__ ymap, uint32_t yp, uint8_t* dst, uint32_t dp, uint32_t dw, uint32_t dh) {
  const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const float sx = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(xmap) + (size_t)y * xp)[x];
  const float sy = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(ymap) + (size_t)y * yp)[x];
  if (!(sx >= 0.f && sx <= (float)(sw - 1) && sy >= 0.f && sy <= (float)(sh - 1))) return;
  const uint32_t x0 = (uint32_t)(int)sx, y0 = (uint32_t)(int)sy;
  const uint32_t x1 = (x0 + 1 < sw) ? x0 + 1 : sw - 1, y1 = (y0 + 1 < sh) ? y0 + 1 : sh - 1;
  const float fx = sx - (float)x0, fy = sy - (float)y0;
  const uint8_t *r0 = src + (size_t)y0 * sp, *r1 = src + (size_t)y1 * sp;
  uint8_t* o = dst + (size_t)y * dp + 3 * (size_t)x;
#pragma unroll
  for (int c = 0; c < 3; c++)
    o[c] = (uint8_t)sat_trunc(bilerp(r0[3 * x0 + c], r0[3 * x1 + c], r1[3 * x0 + c], r1[3 * x1 + c], fx, fy));
}

__global__ __launch_bounds__(256) void k_remap3(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh,
                                                const float* __restrict__ xmap, uint32_t xp,
                                                const float* __restrict__ ymap, uint32_t yp, uint8_t* dst, uint32_t dp,
                                                uint32_t dw, uint32_t dh) {
  remap3_task(src, sp, sw, sh, xmap, xp, ymap, yp, dst, dp, dw, dh);
}
// the same map applied to up to 32 frames in one dispatch (vpf_remap_batch): blockIdx.z = frame
__global__ __launch_bounds__(256) void k_remap3_b(const BatchArgs args, uint32_t sw, uint32_t sh, const float* __restrict__ xmap, uint32_t xp,
                                                  const float* __restrict__ ymap, uint32_t yp, uint32_t dw, uint32_t dh) {
  const FrameDesc& f = args.f[blockIdx.z];
  remap3_task(f.s[0], f.sp[0], sw, sh, xmap, xp, ymap, yp, f.d[0], f.dp[0], dw, dh);
}

// fast remap: lane = 4 consecutive destination pixels, wave = 256 pixels of one row.  Maps come in as two 16-B loads, the
// two source texels of a row (6 contiguous bytes at an arbitrary byte offset) as ONE 12-B load from the enclosing 4-B
// aligned address (v_alignbyte_b32 extracts them), and four valid pixels leave as one 12-B store.  Same arithmetic as
// k_remap3, spelled for the VALU (the kernel sits between the VALU and the HBM roofline, tools/gpu_pmc_remap.sh):
//   * out-of-range coordinates are pulled to the border with one v_med3_f32 and the pixel is computed like any other
//     (just not stored) instead of being steered around the arithmetic;
//   * sx - (float)(int)sx for sx >= 0 is v_fract_f32 (the subtraction is exact, so the bits are the same);
//   * source offsets are 32-bit (v_mad_u32_u24; the launcher checks the surface is < 4 GiB) on a scalar base pointer;
//   * blends of 8-bit samples stay inside [0, 255.5], and the pack is v_cvt_pk_u8_f32 under round-toward-zero (pack12_trunc).
// Requires 4-B aligned src rows, 16-B aligned map rows, dw % 4 == 0, sw >= 4, sh * sp < 2^32.  Never reads past the
// dword-rounded end of the last source row.

// The taps of one pixel column in one source row are 6 bytes starting `o` bytes into the surface; they are fetched as the
// 12-B window that starts at the aligned address below them (tap_window) and cut out with v_alignbyte_b32 (window_taps).
// A window that runs over the end of a row into the next one is harmless (the second tap of the last column has weight
// fx == 0 and fma(0, finite, p0) == p0); only at the very end of the pixel data (`last` = its dword-rounded end - 12) must it
// slide left so that it never leaves the allocation — SLIDE, chosen per wave: the first tap then starts up to 9 bytes
// into the window and bytes past it read as zero.
struct TapWindow { uint32_t e0, e1, e2; };
template <bool SLIDE>
VPF_DEV TapWindow tap_window(const uint8_t* __restrict__ src, uint32_t o, uint32_t last) {
  const uint8_t* p = src + (SLIDE ? min(o & ~3u, last) : (o & ~3u));
  return TapWindow{ldg<false, uint32_t>(p), ldg<false, uint32_t>(p + 4), ldg<false, uint32_t>(p + 8)};
}
template <bool SLIDE>
VPF_DEV void window_taps(TapWindow w, uint32_t o, uint32_t last, float* t0, float* t1) {
  uint32_t lead = o & 3u;
  if constexpr (SLIDE) {
    lead = o - min(o & ~3u, last);
    const uint32_t q = lead >> 2;
    w = TapWindow{q == 0 ? w.e0 : (q == 1 ? w.e1 : w.e2), q == 0 ? w.e1 : (q == 1 ? w.e2 : 0u), q == 0 ? w.e2 : 0u};
    lead &= 3u;
  }
  const uint32_t lo = __builtin_amdgcn_alignbyte(w.e1, w.e0, lead), hi = __builtin_amdgcn_alignbyte(w.e2, w.e1, lead);
  t0[0] = ubyte<0>(lo); t0[1] = ubyte<1>(lo); t0[2] = ubyte<2>(lo);
  t1[0] = ubyte<3>(lo); t1[1] = ubyte<0>(hi); t1[2] = ubyte<1>(hi);
}
// four pixels of one lane -> 12 packed bytes.  All eight windows are requested before the first one is used
// (sched_barrier keeps the compiler from sinking the loads to their uses, which would serialise eight memory round
// trips per wave: 22 -> 28 us per 4K frame); pixels are packed as they are produced to keep the register count at 8
// waves per SIMD.
template <bool SLIDE>
VPF_DEV void remap_blend4(const uint8_t* __restrict__ src, const uint32_t* o0, const uint32_t* o1, uint32_t last, const float* fx,
                          const float* fy, uint32_t* d) {
  float o[12];
  if constexpr (SLIDE) {  // one wave per frame at most: pixel by pixel, so that this path does not set the kernel's register count
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float a0[3], a1[3], b0[3], b1[3];
      window_taps<true>(tap_window<true>(src, o0[k], last), o0[k], last, a0, a1);
      window_taps<true>(tap_window<true>(src, o1[k], last), o1[k], last, b0, b1);
#pragma unroll
      for (int c = 0; c < 3; c++) o[3 * k + c] = bilerp(a0[c], a1[c], b0[c], b1[c], fx[k], fy[k]);
      __builtin_amdgcn_sched_barrier(0);
    }
    pack12_trunc(o, d[0], d[1], d[2]);
  } else {
    TapWindow w0[4], w1[4];
    uint32_t lead[4];  // the pitch is a multiple of 4: both rows of a pixel share the lead
#pragma unroll
    for (int k = 0; k < 4; k++) { w0[k] = tap_window<false>(src, o0[k], last); w1[k] = tap_window<false>(src, o1[k], last); lead[k] = o0[k] & 3u; }
    __builtin_amdgcn_sched_barrier(0);
    d[0] = d[1] = d[2] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float a0[3], a1[3], b0[3], b1[3];
      window_taps<false>(w0[k], lead[k], last, a0, a1);
      window_taps<false>(w1[k], lead[k], last, b0, b1);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int j = 3 * k + c;  // byte j of the 12
        d[j >> 2] |= (uint32_t)bilerp(a0[c], a1[c], b0[c], b1[c], fx[k], fy[k]) << (8 * (j & 3));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Cache policy (4K, us per frame): plain map loads + non-temporal destination stores 22.2; all plain 23.7; non-temporal
// map loads 25.2 (+ NT stores 23.6).  Assigning each XCD a horizontal band of the picture (so that vertically adjacent
// tiles share an L2) was slower as well: 23.4 vs 21.9; so was padding the grid width to a multiple of 8 (a column of tiles per
// XCD): 23.0 vs 22.2.
VPF_DEV void remap3_p4_task(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh, const float* __restrict__ xmap, uint32_t xp,
                            const float* __restrict__ ymap, uint32_t yp, uint8_t* dst, uint32_t dp, uint32_t dw, uint32_t dh, int vec_ok) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const uint32_t x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const f32x4 sx4 = ldg<false, f32x4>(reinterpret_cast<const uint8_t*>(xmap) + (size_t)y * xp + 4 * (size_t)x);
  const f32x4 sy4 = ldg<false, f32x4>(reinterpret_cast<const uint8_t*>(ymap) + (size_t)y * yp + 4 * (size_t)x);
  const float wmax = (float)(sw - 1), hmax = (float)(sh - 1);
  const uint32_t last = (((sh - 1) * sp + 3 * sw + 3) & ~3u) - 12;  // the last window that stays inside the pixel data (rounded up to a dword)
  float fx[4], fy[4];
  uint32_t o0[4], o1[4];  // first tap of the two source rows, bytes into the surface
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float sx = sx4[k], sy = sy4[k];
    // in range <=> pulling the coordinate to the border leaves it unchanged (NaN compares unequal; -0.0 == 0.0)
    const float cx = __builtin_amdgcn_fmed3f(sx, 0.f, wmax), cy = __builtin_amdgcn_fmed3f(sy, 0.f, hmax);
    ok[k] = (bool)((int)(cx == sx) & (int)(cy == sy));
    const uint32_t x0 = (uint32_t)(int)cx, y0 = (uint32_t)(int)cy;  // a NaN coordinate becomes 0 here (v_cvt_i32_f32)
    fx[k] = __builtin_amdgcn_fractf(cx); fy[k] = __builtin_amdgcn_fractf(cy);
    o0[k] = mad24(y0, sp, 3 * x0); o1[k] = o0[k] + (y0 + 1 < sh ? sp : 0u);
  }
  // o0 <= o1: the lower row decides whether a window could leave the surface; wave-uniform, true for one wave at most
  const uint32_t omax = max(max(o1[0], o1[1]), max(o1[2], o1[3]));
  uint32_t d[3];
  if (__builtin_amdgcn_ballot_w64((omax & ~3u) > last) != 0) remap_blend4<true>(src, o0, o1, last, fx, fy, d);
  else remap_blend4<false>(src, o0, o1, last, fx, fy, d);
  uint8_t* out = dst + (size_t)y * dp + 3 * (size_t)x;
  if (vec_ok && ok[0] && ok[1] && ok[2] && ok[3]) {
    stg3<true>(out, d[0], d[1], d[2]);
  } else {
#pragma unroll
    for (int j = 0; j < 12; j++)
      if (ok[j / 3]) out[j] = (uint8_t)(d[j >> 2] >> (8 * (j & 3)));
  }
}

__global__ __launch_bounds__(256, 8) void k_remap3_p4(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh,
                                                   const float* __restrict__ xmap, uint32_t xp, const float* __restrict__ ymap,
                                                   uint32_t yp, uint8_t* dst, uint32_t dp, uint32_t dw, uint32_t dh, int vec_ok) {
  remap3_p4_task(src, sp, sw, sh, xmap, xp, ymap, yp, dst, dp, dw, dh, vec_ok);
}
__global__ __launch_bounds__(256, 8) void k_remap3_p4_b(const BatchArgs args, uint32_t sw, uint32_t sh, const float* __restrict__ xmap, uint32_t xp,
                                                     const float* __restrict__ ymap, uint32_t yp, uint32_t dw, uint32_t dh, int vec_ok) {
  const FrameDesc& f = args.f[blockIdx.z];  // the maps are shared: frame i + 1 finds them in L2 / Infinity Cache
  remap3_p4_task(f.s[0], f.sp[0], sw, sh, xmap, xp, ymap, yp, f.d[0], f.dp[0], dw, dh, vec_ok);
}

// one map applied to n <= kSmallBatch frames in one dispatch
hipError_t launch_remap_batch(hipStream_t st, uint32_t sw, uint32_t sh, const float* xmap, uint32_t xp, const float* ymap, uint32_t yp, uint32_t dw,
                              uint32_t dh, uint32_t n, const BatchArgs& a) {
  const int tune = tuning(VPF_TUNE_NV12_RGB_VARIANT);
  bool fast = tune != 9 && (dw % 4 == 0) && !(((uintptr_t)xmap | xp | (uintptr_t)ymap | yp) & 15) && sw >= 4;
  int vec_ok = 1;
  for (uint32_t i = 0; i < n; i++) {
    const FrameDesc& f = a.f[i];
    fast = fast && !(((uintptr_t)f.s[0] | f.sp[0]) & 3) && f.sp[0] < (1u << 24) && (uint64_t)sh * f.sp[0] < (1ull << 32);
    vec_ok &= ((((uintptr_t)f.d[0] | f.dp[0]) & 3) == 0);
  }
  if (fast) {
    VPF_LAUNCH(k_remap3_p4_b, dim3((dw / 4 + 63) / 64, (dh + 3) / 4, n), dim3(256), 0, st, a, sw, sh, xmap, xp, ymap, yp, dw, dh, vec_ok);
    return hipGetLastError();
  }
  VPF_LAUNCH(k_remap3_b, dim3((dw + 63) / 64, (dh + 3) / 4, n), dim3(256), 0, st, a, sw, sh, xmap, xp, ymap, yp, dw, dh);
  return hipGetLastError();
}

hipError_t launch_remap(hipStream_t st, uint32_t sw, uint32_t sh, const uint8_t* src, uint32_t sp, const float* xmap,
                        uint32_t xp, const float* ymap, uint32_t yp, uint32_t dw, uint32_t dh, uint8_t* dst,
                        uint32_t dp) {
  const int tune = tuning(VPF_TUNE_NV12_RGB_VARIANT);
  const bool fast = tune != 9 && (dw % 4 == 0) && !(((uintptr_t)src | sp) & 3) &&
                    !(((uintptr_t)xmap | xp | (uintptr_t)ymap | yp) & 15) && sw >= 4 && sp < (1u << 24) &&
                    (uint64_t)sh * sp < (1ull << 32);  // 32-bit source offsets (v_mad_u32_u24)
  const int vec_ok = ((((uintptr_t)dst | dp) & 3) == 0);
  if (fast) {
    dim3 fgrid((dw / 4 + 63) / 64, (dh + 3) / 4);
    VPF_LAUNCH(k_remap3_p4, fgrid, dim3(256), 0, st, src, sp, sw, sh, xmap, xp, ymap, yp, dst, dp, dw, dh, vec_ok);
    return hipGetLastError();
  }
  dim3 grid((dw + 63) / 64, (dh + 3) / 4);
  VPF_LAUNCH(k_remap3, grid, dim3(256), 0, st, src, sp, sw, sh, xmap, xp, ymap, yp, dst, dp, dw, dh);
  return hipGetLastError();
}

}  // namespace vpf
