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

hipError_t launch_rgb_to_yuv(hipStream_t st, int src_fc, int dst_fc, const Rgb2YuvCoef& c, uint32_t w, uint32_t h,
                             uint32_t n, const BatchArgs& a) {
  dim3 grid(((w + 1) / 2 + 63) / 64, ((h + 1) / 2 + 3) / 4, n);
  const bool sub = (dst_fc == FC_YUV420);
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
