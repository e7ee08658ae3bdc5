// k_rgb2yuv.hip — RGB / BGR / RGB_PLANAR -> YUV444 / YUV420 / YCBCR for gfx950.
//
// Replaces the NPP calls behind bgr_yuv444, bgr_ycbcr, rgb_yuv444, rgb_planar_yuv444, rgb_yuv420
// (reference: src/TC/src/TasksColorCvt.cpp:626-672,686-717,731-772,786-830,887-931).  BT.601 only, as
// in the reference; JPEG range = NPP "YUV" model, MPEG range = NPP "YCbCr" model.
//
// One lane owns a 2x2 pixel quad: four luma samples, and either four chroma pairs (4:4:4) or one
// (4:2:0, mean of the quad: matrix row applied to 0.25 * the exact integer channel sums).
// 4.5-6 B/px of streaming traffic; the quad shape keeps the 4:2:0 decimation in registers.
#include "vpf_device.h"

namespace vpf {

VPF_DEV float mrow(const Rgb2YuvCoef& c, int k, float r, float g, float b) {
  return __builtin_fmaf(r, c.m[k][0], __builtin_fmaf(g, c.m[k][1], __builtin_fmaf(b, c.m[k][2], c.d[k])));
}

template <int SRC /*FC_RGB, FC_BGR, FC_PLANAR*/, bool SUB /*4:2:0 output*/>
__global__ __launch_bounds__(256) void k_rgb_yuv_quad(const BatchArgs args, const Rgb2YuvCoef c, uint32_t w,
                                                      uint32_t h) {
  const FrameDesc f = args.f[blockIdx.z];
  const uint32_t qx = blockIdx.x * 64 + (threadIdx.x & 63), qy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const uint32_t x0 = 2 * qx, y0 = 2 * qy;
  if (x0 >= w || y0 >= h) return;
  const uint32_t x1 = (x0 + 1 < w) ? x0 + 1 : x0, y1 = (y0 + 1 < h) ? y0 + 1 : y0;  // edge quads replicate
  const uint32_t xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
  float rs = 0.f, gs = 0.f, bs = 0.f;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const uint32_t x = xs[t], y = ys[t];
    float r, g, b;
    if constexpr (SRC == FC_PLANAR) {
      r = f.s[0][(size_t)y * f.sp[0] + x]; g = f.s[1][(size_t)y * f.sp[1] + x]; b = f.s[2][(size_t)y * f.sp[2] + x];
    } else {
      const uint8_t* p = f.s[0] + (size_t)y * f.sp[0] + 3 * (size_t)x;
      r = p[SRC == FC_BGR ? 2 : 0]; g = p[1]; b = p[SRC == FC_BGR ? 0 : 2];
    }
    rs += r; gs += g; bs += b;  // exact: small integers
    const bool dup = (t == 1 && x1 == x0) || (t == 2 && y1 == y0) || (t == 3 && (x1 == x0 || y1 == y0));
    if (!dup) {
      f.d[0][(size_t)y * f.dp[0] + x] = (uint8_t)sat_trunc(mrow(c, 0, r, g, b));
      if constexpr (!SUB) {
        f.d[1][(size_t)y * f.dp[1] + x] = (uint8_t)sat_trunc(mrow(c, 1, r, g, b));
        f.d[2][(size_t)y * f.dp[2] + x] = (uint8_t)sat_trunc(mrow(c, 2, r, g, b));
      }
    }
  }
  if constexpr (SUB) {
    rs *= 0.25f; gs *= 0.25f; bs *= 0.25f;  // exact in fp32
    f.d[1][(size_t)qy * f.dp[1] + qx] = (uint8_t)sat_trunc(mrow(c, 1, rs, gs, bs));
    f.d[2][(size_t)qy * f.dp[2] + qx] = (uint8_t)sat_trunc(mrow(c, 2, rs, gs, bs));
  }
}

// ------------------------------------------------------------------------------------------
// fast path: one lane = 4 px x 2 rows.  Packed sources come in as three dwords per row (12 B, dwordx3), planar as
// one dword per plane per row; luma (and 4:4:4 chroma) leave as one dword per row, 4:2:0 chroma as two bytes per
// plane.  Every access of a wave is contiguous (768 B loads, 256 B / 128 B stores).
// Requires w % 4 == 0, h even, 4-B aligned planes (2-B for subsampled chroma).  Same arithmetic as the quad kernel.
// ------------------------------------------------------------------------------------------
template <int SRC>
VPF_DEV void load_rgb4(const FrameDesc& f, uint32_t x, uint32_t y, float* r, float* g, float* b) {
  if constexpr (SRC == FC_PLANAR) {
    const uint32_t rd = ldg<false, uint32_t>(f.s[0] + (size_t)y * f.sp[0] + x);
    const uint32_t gd = ldg<false, uint32_t>(f.s[1] + (size_t)y * f.sp[1] + x);
    const uint32_t bd = ldg<false, uint32_t>(f.s[2] + (size_t)y * f.sp[2] + x);
    r[0] = ubyte<0>(rd); r[1] = ubyte<1>(rd); r[2] = ubyte<2>(rd); r[3] = ubyte<3>(rd);
    g[0] = ubyte<0>(gd); g[1] = ubyte<1>(gd); g[2] = ubyte<2>(gd); g[3] = ubyte<3>(gd);
    b[0] = ubyte<0>(bd); b[1] = ubyte<1>(bd); b[2] = ubyte<2>(bd); b[3] = ubyte<3>(bd);
  } else {
    const uint8_t* p = f.s[0] + (size_t)y * f.sp[0] + 3 * (size_t)x;
    const uint32_t d0 = ldg<false, uint32_t>(p), d1 = ldg<false, uint32_t>(p + 4), d2 = ldg<false, uint32_t>(p + 8);
    // d0 = c0a c1a c2a c0b ; d1 = c1b c2b c0c c1c ; d2 = c2c c0d c1d c2d
    float* c0 = (SRC == FC_BGR) ? b : r;
    float* c2 = (SRC == FC_BGR) ? r : b;
    c0[0] = ubyte<0>(d0); g[0] = ubyte<1>(d0); c2[0] = ubyte<2>(d0);
    c0[1] = ubyte<3>(d0); g[1] = ubyte<0>(d1); c2[1] = ubyte<1>(d1);
    c0[2] = ubyte<2>(d1); g[2] = ubyte<3>(d1); c2[2] = ubyte<0>(d2);
    c0[3] = ubyte<1>(d2); g[3] = ubyte<2>(d2); c2[3] = ubyte<3>(d2);
  }
}

template <int SRC, bool SUB>
__global__ __launch_bounds__(256) void k_rgb_yuv_p4(const BatchArgs args, const Rgb2YuvCoef c, uint32_t w, uint32_t h,
                                                    uint32_t groups_x) {
  const FrameDesc f = args.f[blockIdx.z];
  const uint32_t gx = blockIdx.x * 64 + (threadIdx.x & 63), rp = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (gx >= groups_x || rp >= (h >> 1)) return;
  const uint32_t x = gx * 4;
  float r[2][4], g[2][4], b[2][4];
  load_rgb4<SRC>(f, x, 2 * rp, r[0], g[0], b[0]);
  load_rgb4<SRC>(f, x, 2 * rp + 1, r[1], g[1], b[1]);
#pragma unroll
  for (int row = 0; row < 2; row++) {
    const size_t y = 2 * rp + row;
    stg<false, uint32_t>(f.d[0] + y * f.dp[0] + x, pack4_trunc(mrow(c, 0, r[row][0], g[row][0], b[row][0]), mrow(c, 0, r[row][1], g[row][1], b[row][1]),
                                                               mrow(c, 0, r[row][2], g[row][2], b[row][2]), mrow(c, 0, r[row][3], g[row][3], b[row][3])));
    if constexpr (!SUB) {
#pragma unroll
      for (int k = 1; k < 3; k++)
        stg<false, uint32_t>(f.d[k] + y * f.dp[k] + x, pack4_trunc(mrow(c, k, r[row][0], g[row][0], b[row][0]), mrow(c, k, r[row][1], g[row][1], b[row][1]),
                                                                   mrow(c, k, r[row][2], g[row][2], b[row][2]), mrow(c, k, r[row][3], g[row][3], b[row][3])));
    }
  }
  if constexpr (SUB) {
    // two quads: px {0,1} and {2,3} of both rows; sums of small integers are exact in fp32, as is the 0.25 scale
    float qr[2], qg[2], qb[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      qr[q] = 0.25f * (r[0][2 * q] + r[0][2 * q + 1] + r[1][2 * q] + r[1][2 * q + 1]);
      qg[q] = 0.25f * (g[0][2 * q] + g[0][2 * q + 1] + g[1][2 * q] + g[1][2 * q + 1]);
      qb[q] = 0.25f * (b[0][2 * q] + b[0][2 * q + 1] + b[1][2 * q] + b[1][2 * q + 1]);
    }
#pragma unroll
    for (int k = 1; k < 3; k++) {
      const uint32_t v = sat_trunc(mrow(c, k, qr[0], qg[0], qb[0])) | (sat_trunc(mrow(c, k, qr[1], qg[1], qb[1])) << 8);
      stg<false, uint16_t>(f.d[k] + (size_t)rp * f.dp[k] + (x >> 1), (uint16_t)v);
    }
  }
}

// ------------------------------------------------------------------------------------------
// r16: 16 px per lane, dense 1-KiB non-temporal accesses.  4:4:4 outputs: a wave = one row x 1024 px (3 loads, 3
// stores).  4:2:0: a wave = one row pair (6 loads in flight, two 1-KiB luma stores + two 512-B chroma stores).
// Packed sources arrive through the wave-private LDS transpose (load side of store_run48).  Same arithmetic as above.
// Requires w % 16 == 0, h even for 4:2:0, 16-B aligned planes / pitches (8-B for subsampled chroma).
// ------------------------------------------------------------------------------------------
template <int SRC, bool SUB>
VPF_DEV void rgb_yuv_r16_task(const FrameDesc& f, const Rgb2YuvCoef& c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  constexpr int ROWS = SUB ? 2 : 1;
  __shared__ u32x4 tile[SRC == FC_PLANAR ? 1 : 4 * 192 * ROWS];
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = blockIdx.x * 4 + wv;
  if (wt >= n_tasks) return;
  const uint32_t rg = wt / chunks_x, chunk = wt - rg * chunks_x;
  const uint32_t y0 = rg * ROWS, x = chunk * 1024 + lane * 16;
  uint32_t p0[ROWS][4], p1[ROWS][4], p2[ROWS][4];  // channel planes in memory order, 4 px per dword
  if constexpr (SRC == FC_PLANAR) {
    if (x >= w) return;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const u32x4 a = ldg<true, u32x4>(f.s[0] + (size_t)(y0 + r) * f.sp[0] + x), b = ldg<true, u32x4>(f.s[1] + (size_t)(y0 + r) * f.sp[1] + x),
                  d = ldg<true, u32x4>(f.s[2] + (size_t)(y0 + r) * f.sp[2] + x);
#pragma unroll
      for (int g = 0; g < 4; g++) { p0[r][g] = a[g]; p1[r][g] = b[g]; p2[r][g] = d[g]; }
    }
  } else {
    u32x4* t = tile + wv * 192 * ROWS;
    const uint32_t row_bytes = 3 * w;
    u32x4 q[ROWS][3];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        uint32_t off = chunk * 3072 + (k * 64 + lane) * 16;
        off = off < row_bytes ? off : row_bytes - 16;
        q[r][k] = ldg<true, u32x4>(f.s[0] + (size_t)(y0 + r) * f.sp[0] + off);
      }
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) t[r * 192 + k * 64 + lane] = q[r][k];
    wave_sync();
    if (x >= w) return;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      uint32_t d[12];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const u32x4 v = t[r * 192 + lane * 3 + j];
        d[4 * j] = v[0]; d[4 * j + 1] = v[1]; d[4 * j + 2] = v[2]; d[4 * j + 3] = v[3];
      }
#pragma unroll
      for (int g = 0; g < 4; g++) deint4(d[3 * g], d[3 * g + 1], d[3 * g + 2], p0[r][g], p1[r][g], p2[r][g]);
    }
  }
  auto R = [&](int r, int g) { return SRC == FC_BGR ? p2[r][g] : p0[r][g]; };
  auto B = [&](int r, int g) { return SRC == FC_BGR ? p0[r][g] : p2[r][g]; };
  auto row4 = [&](int k, uint32_t rd, uint32_t gd, uint32_t bd) {
    return pack4_trunc(mrow(c, k, ubyte<0>(rd), ubyte<0>(gd), ubyte<0>(bd)), mrow(c, k, ubyte<1>(rd), ubyte<1>(gd), ubyte<1>(bd)),
                       mrow(c, k, ubyte<2>(rd), ubyte<2>(gd), ubyte<2>(bd)), mrow(c, k, ubyte<3>(rd), ubyte<3>(gd), ubyte<3>(bd)));
  };
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
#pragma unroll
    for (int k = 0; k < (SUB ? 1 : 3); k++) {
      u32x4 o;
#pragma unroll
      for (int g = 0; g < 4; g++) o[g] = row4(k, R(r, g), p1[r][g], B(r, g));
      stg<true, u32x4>(f.d[k] + (size_t)(y0 + r) * f.dp[k] + x, o);
    }
  }
  if constexpr (SUB) {
    uint32_t cu[2] = {0, 0}, cv[2] = {0, 0};
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const uint32_t r0 = R(0, g), r1 = R(1, g), g0 = p1[0][g], g1 = p1[1][g], b0 = B(0, g), b1 = B(1, g);
      // two quads: px {0,1} and {2,3} of both rows; sums of small integers and the 0.25 scale are exact in fp32
      const float qr0 = 0.25f * (ubyte<0>(r0) + ubyte<1>(r0) + ubyte<0>(r1) + ubyte<1>(r1)), qr1 = 0.25f * (ubyte<2>(r0) + ubyte<3>(r0) + ubyte<2>(r1) + ubyte<3>(r1));
      const float qg0 = 0.25f * (ubyte<0>(g0) + ubyte<1>(g0) + ubyte<0>(g1) + ubyte<1>(g1)), qg1 = 0.25f * (ubyte<2>(g0) + ubyte<3>(g0) + ubyte<2>(g1) + ubyte<3>(g1));
      const float qb0 = 0.25f * (ubyte<0>(b0) + ubyte<1>(b0) + ubyte<0>(b1) + ubyte<1>(b1)), qb1 = 0.25f * (ubyte<2>(b0) + ubyte<3>(b0) + ubyte<2>(b1) + ubyte<3>(b1));
      const uint32_t u2 = sat_trunc(mrow(c, 1, qr0, qg0, qb0)) | (sat_trunc(mrow(c, 1, qr1, qg1, qb1)) << 8);
      const uint32_t v2 = sat_trunc(mrow(c, 2, qr0, qg0, qb0)) | (sat_trunc(mrow(c, 2, qr1, qg1, qb1)) << 8);
      cu[g >> 1] |= u2 << (16 * (g & 1));
      cv[g >> 1] |= v2 << (16 * (g & 1));
    }
    stg<true, u32x2>(f.d[1] + (size_t)rg * f.dp[1] + (x >> 1), u32x2{cu[0], cu[1]});
    stg<true, u32x2>(f.d[2] + (size_t)rg * f.dp[2] + (x >> 1), u32x2{cv[0], cv[1]});
  }
}
template <int SRC, bool SUB>
__global__ __launch_bounds__(256) void k_rgb_yuv_r16(const BatchArgs args, const Rgb2YuvCoef c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  rgb_yuv_r16_task<SRC, SUB>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
template <int SRC, bool SUB>  // single-frame entry: scalar arguments (see VPF_ONE_SRC_PARAMS in vpf_internal.h)
__global__ __launch_bounds__(256) void k_rgb_yuv_r16_one(VPF_ONE_SRC_PARAMS, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks, VPF_ONE_DST_PARAMS, const Rgb2YuvCoef c) {
  rgb_yuv_r16_task<SRC, SUB>(VPF_ONE_FRAME, c, w, h, chunks_x, n_tasks);
}

static bool rgb2yuv_r16_ok(const BatchArgs& a, uint32_t n, int src_fc, bool sub, uint32_t w, uint32_t h) {
  const int tv = tuning(VPF_TUNE_NV12_RGB_VARIANT);
  if (tv == 9 || tv == 40 || (w & 15) || (sub && (h & 1))) return false;
  const int ns = (src_fc == FC_PLANAR) ? 3 : 1;
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < ns; k++)
      if (((uintptr_t)a.f[i].s[k] | a.f[i].sp[k]) & 15) return false;
    for (int k = 0; k < 3; k++)
      if (((uintptr_t)a.f[i].d[k] | a.f[i].dp[k]) & ((k && sub) ? 7 : 15)) return false;
  }
  return true;
}

static bool rgb2yuv_fast_ok(const BatchArgs& a, uint32_t n, int src_fc, bool sub, uint32_t w, uint32_t h) {
  if (tuning(VPF_TUNE_NV12_RGB_VARIANT) == 9 || (w & 3) || (h & 1)) return false;
  const int ns = (src_fc == FC_PLANAR) ? 3 : 1;
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < ns; k++)
      if (((uintptr_t)a.f[i].s[k] | a.f[i].sp[k]) & 3) return false;
    for (int k = 0; k < 3; k++)
      if (((uintptr_t)a.f[i].d[k] | a.f[i].dp[k]) & ((k && sub) ? 1 : 3)) return false;
  }
  return true;
}

hipError_t launch_rgb_to_yuv(hipStream_t st, int src_fc, int dst_fc, const Rgb2YuvCoef& c, uint32_t w, uint32_t h,
                             uint32_t n, const BatchArgs& a) {
  const bool sub = (dst_fc == FC_YUV420);
  // (a lone 4K frame used to take the narrower p4 kernel here — a row-pair wave halves the wave count, and p4 won by 8 %;
  // with the scalar-argument single-frame entry r16 is the faster one again: 0.51 vs 0.47 of 8 TB/s)
  if (rgb2yuv_r16_ok(a, n, src_fc, sub, w, h)) {
    const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * (sub ? h / 2 : h);
    dim3 rgrid((tasks + 3) / 4, n);
#define VPF_R16(S)                                                                                    \
  if (n == 1 && sub) VPF_LAUNCH((k_rgb_yuv_r16_one<S, true>), rgrid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), w, h, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c); \
  else if (n == 1) VPF_LAUNCH((k_rgb_yuv_r16_one<S, false>), rgrid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), w, h, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c); \
  else if (sub) VPF_LAUNCH((k_rgb_yuv_r16<S, true>), rgrid, dim3(256), 0, st, a, c, w, h, chunks, tasks);   \
  else VPF_LAUNCH((k_rgb_yuv_r16<S, false>), rgrid, dim3(256), 0, st, a, c, w, h, chunks, tasks);      \
  return hipGetLastError();
    switch (src_fc) {
      case FC_RGB: VPF_R16(FC_RGB)
      case FC_BGR: VPF_R16(FC_BGR)
      case FC_PLANAR: VPF_R16(FC_PLANAR)
      default: return hipErrorInvalidValue;
    }
#undef VPF_R16
  }
  if (rgb2yuv_fast_ok(a, n, src_fc, sub, w, h)) {
    dim3 fgrid((w / 4 + 63) / 64, (h / 2 + 3) / 4, n);
#define VPF_FAST(S)                                                                                  \
  if (sub) VPF_LAUNCH((k_rgb_yuv_p4<S, true>), fgrid, dim3(256), 0, st, a, c, w, h, w / 4);           \
  else VPF_LAUNCH((k_rgb_yuv_p4<S, false>), fgrid, dim3(256), 0, st, a, c, w, h, w / 4);              \
  return hipGetLastError();
    switch (src_fc) {
      case FC_RGB: VPF_FAST(FC_RGB)
      case FC_BGR: VPF_FAST(FC_BGR)
      case FC_PLANAR: VPF_FAST(FC_PLANAR)
      default: return hipErrorInvalidValue;
    }
#undef VPF_FAST
  }
  dim3 grid(((w + 1) / 2 + 63) / 64, ((h + 1) / 2 + 3) / 4, n);
#define VPF_GO(S)                                                                                            \
  if (sub) VPF_LAUNCH((k_rgb_yuv_quad<S, true>), grid, dim3(256), 0, st, a, c, w, h);                 \
  else VPF_LAUNCH((k_rgb_yuv_quad<S, false>), grid, dim3(256), 0, st, a, c, w, h);                    \
  return hipGetLastError();
  switch (src_fc) {
    case FC_RGB: VPF_GO(FC_RGB)
    case FC_BGR: VPF_GO(FC_BGR)
    case FC_PLANAR: VPF_GO(FC_PLANAR)
    default: return hipErrorInvalidValue;
  }
#undef VPF_GO
}

}  // namespace vpf
