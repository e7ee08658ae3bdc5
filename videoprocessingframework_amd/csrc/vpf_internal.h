// vpf_internal.h — shared between the translation units of libvpfhip (gfx950 only).
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vpf_hip.h"

namespace vpf {

// ------------------------------------------------------------------------------------------
// YUV -> RGB coefficients in the form the kernels consume.  Biases fold the luma offset and the
// chroma -128, so a pixel costs one FMA per channel:
//   rc = fma(V, rv, br);  gc = fma(U, gu, fma(V, gv, bg));  bc = fma(U, bu, bb)     (per chroma sample)
//   R = sat_rne(fma(Y, cy, rc)); G = sat_rne(fma(Y, cy, gc)); B = sat_rne(fma(Y, cy, bc))
// sat_rne = saturate to [0,255], round to nearest even (v_cvt_pk_u8_f32)
// ------------------------------------------------------------------------------------------
struct Yuv2RgbCoef {
  float cy, rv, gu, gv, bu, br, bg, bb;
};
// RGB -> YUV: out_k = sat_trunc(fma(R, m[k][0], fma(G, m[k][1], fma(B, m[k][2], d[k]))))
struct Rgb2YuvCoef {
  float m[3][3];
  float d[3];
};
bool make_yuv2rgb(int color_space, int color_range, Yuv2RgbCoef* out);
bool make_rgb2yuv(int color_range, Rgb2YuvCoef* out);

// The frame table of a batched launch travels in the kernarg segment (no device-side table to manage).  Two sizes since round 5:
//   BatchArgs   32 frames x 72 B = 2.3 KiB — every converter and remap kernel, and every resize / fused launch of up to 32 frames (one
//               frame per dispatch included: the unmodified per-Execute() API);
//   BatchArgsL  128 frames = 9 KiB — resize and fused launches of 33 .. 128 frames.  A batch of small planes is 25-50 us of kernel per 32
//               frames; every dispatch boundary costs ~3 us of idle chip plus a tail of partly empty CUs of about one wave life
//               (profiles/r05_wave_times.txt): four times the frames per dispatch amortise both.
// The kernarg segment is write-combined host memory: a launch pays ~0.2 us of HOST time per KiB of argument (9 KiB: + 1.3-2 us per call,
// profiles/r05_abi_launch_rate_9k_kernarg.txt) — nothing against a 128-frame dispatch, 40-70 % of a single frame's issue time.  Hence
// two instantiations of the resize / fused batch kernels (template parameter BA) instead of one large table for everybody.
constexpr int kSmallBatch = 32;
constexpr int kMaxBatch = 128;
struct FrameDesc {
  const uint8_t* s[3];
  uint8_t* d[3];
  uint32_t sp[3];
  uint32_t dp[3];
};
template <int CAP>
struct BatchArgsT {
  FrameDesc f[CAP];
};
using BatchArgs = BatchArgsT<kSmallBatch>;
using BatchArgsL = BatchArgsT<kMaxBatch>;
// the small table of a launch of n <= kSmallBatch frames (host side: the launchers pass BatchArgsL around and cut it down at the launch)
inline BatchArgs small_batch(const BatchArgsL& a, uint32_t n) {
  BatchArgs s;
  std::memcpy(s.f, a.f, (size_t)(n < (uint32_t)kSmallBatch ? n : (uint32_t)kSmallBatch) * sizeof(FrameDesc));
  return s;
}
// Single-frame kernel entries (one Execute() = one launch) take the frame as SCALAR arguments, source side and task
// counts first: the library is built with -amdgpu-kernarg-preload-count=16, so the dispatcher hands those to the wave in
// SGPRs and its first loads do not wait for a scalar-cache round trip to the kernarg segment (a by-value BatchArgs is
// never preloaded).  Worth 0.5-0.8 us per launch on kernels that last 4-7 us.
#define VPF_ONE_SRC_PARAMS const uint8_t *s0, const uint8_t *s1, const uint8_t *s2, uint32_t sp0, uint32_t sp1, uint32_t sp2
#define VPF_ONE_DST_PARAMS uint8_t *d0, uint8_t *d1, uint8_t *d2, uint32_t dp0, uint32_t dp1, uint32_t dp2
#define VPF_ONE_FRAME FrameDesc{{s0, s1, s2}, {d0, d1, d2}, {sp0, sp1, sp2}, {dp0, dp1, dp2}}
#define VPF_ONE_SRC_ARGS(f) (f).s[0], (f).s[1], (f).s[2], (f).sp[0], (f).sp[1], (f).sp[2]
#define VPF_ONE_DST_ARGS(f) (f).d[0], (f).d[1], (f).d[2], (f).dp[0], (f).dp[1], (f).dp[2]

enum FmtClass : int {
  FC_NV12 = 0,    // Y + interleaved UV, 4:2:0
  FC_YUV420 = 1,  // Y + U + V planes, 4:2:0
  FC_YUV444 = 2,  // three full planes
  FC_RGB = 3,     // packed R,G,B
  FC_BGR = 4,     // packed B,G,R
  FC_PLANAR = 5,  // three full planes R,G,B
};

// launchers (one per translation unit); all asynchronous on `st`
hipError_t launch_yuv_to_rgb(hipStream_t st, int src_fc, int dst_fc, const Yuv2RgbCoef& c, uint32_t w,
                             uint32_t h, uint32_t n, const BatchArgs& a, int variant, bool dst_reused = false);
hipError_t launch_rgb_to_yuv(hipStream_t st, int src_fc, int dst_fc /*FC_YUV444|FC_YUV420*/,
                             const Rgb2YuvCoef& c, uint32_t w, uint32_t h, uint32_t n, const BatchArgs& a);
hipError_t launch_relayout(hipStream_t st, int src_fmt, int dst_fmt, uint32_t w, uint32_t h, uint32_t n,
                           const BatchArgs& a);
hipError_t launch_resize(hipStream_t st, int channels, int interp, uint32_t sw, uint32_t sh, const uint8_t* src,
                         uint32_t spitch, uint32_t dw, uint32_t dh, uint8_t* dst, uint32_t dpitch);
hipError_t launch_resize_f32(hipStream_t st, int channels, int interp, uint32_t sw, uint32_t sh, const uint8_t* src,
                             uint32_t spitch, uint32_t dw, uint32_t dh, uint8_t* dst, uint32_t dpitch);
// one plane of a resize: `ch` interleaved channels (for float surfaces: floats per pixel), `k` = plane index in FrameDesc
struct ResizeJob {
  int ch, k;
  uint32_t sw, sh, dw, dh;
};
// all planes of a format over n <= kMaxBatch same-shape frames in as few launches as possible (k_resize.hip)
hipError_t launch_resize_jobs(hipStream_t st, bool f32, int interp, int njobs, const ResizeJob* jobs, uint32_t n, const BatchArgsL& a);
// 8-bit Lanczos-3 on the matrix cores (k_lanczos_mfma.hip): every plane in `jobs` over n frames in ONE dispatch; false when it does not apply
// (window / ring bounds of vpf_plan_bounds.h, 16-B aligned rows) and nothing was launched
bool launch_lanczos_mfma(hipStream_t st, int njobs, const ResizeJob* jobs, uint32_t n, const BatchArgsL& a);
// the caller-owned table workspace of the vpf_resize_ws / vpf_resize_batch_ws call the current thread is inside (nullptr: none)
void set_lanczos_workspace(vpf_workspace* ws);
uint64_t lanczos_table_bytes_bound(int ch, uint32_t dw, uint32_t dh);  // upper bound of one plane's table bytes under any launch shape
hipError_t launch_remap(hipStream_t st, uint32_t sw, uint32_t sh, const uint8_t* src, uint32_t spitch,
                        const float* xmap, uint32_t xpitch, const float* ymap, uint32_t ypitch, uint32_t dw,
                        uint32_t dh, uint8_t* dst, uint32_t dpitch);
hipError_t launch_remap_batch(hipStream_t st, uint32_t sw, uint32_t sh, const float* xmap, uint32_t xpitch, const float* ymap, uint32_t ypitch,
                              uint32_t dw, uint32_t dh, uint32_t n, const BatchArgs& a);
hipError_t launch_convert_resize(hipStream_t st, int src_fc, int dst_fc, const Yuv2RgbCoef& c, uint32_t sw,
                                 uint32_t sh, uint32_t n, const BatchArgsL& a, uint32_t dw, uint32_t dh);

int tuning(int key);

// Diagnostics (both off by default, one relaxed atomic load per call when off):
//   VPF_HIP_LOG=1    launch / runtime errors on stderr;  VPF_HIP_LOG=2  also which kernel every launch selected
//   VPF_HIP_ROCTX=1  a roctx range around every C-ABI entry and a marker per kernel selection (rocprofv3 --marker-trace shows them
//                    next to the kernels) — the counterpart of the reference's NvtxMark (src/TC/inc/Tasks.hpp:27-52, USE_NVTX);
//                    librocprofiler-sdk-roctx is dlopen()ed on first use, so there is no link-time dependency
int log_level();
void note_kernel(const char* kernel_expr);  // called by VPF_LAUNCH when log_level() >= 2 or roctx is on
bool trace_on();
void trace_push(const char* name);
void trace_pop();
struct Mark {
  bool on;
  explicit Mark(const char* name) : on(trace_on()) { if (on) trace_push(name); }
  ~Mark() { if (on) trace_pop(); }
  Mark(const Mark&) = delete;
  Mark& operator=(const Mark&) = delete;
};

// hipGetLastError() is sticky per thread: an unrelated earlier failure (e.g. a caller's bad memcpy) would be
// reported by the check that follows a launch.  Clear it first so the check sees this launch only.
#define VPF_LAUNCH(K, ...)                                               \
  do {                                                                   \
    if (vpf::log_level() >= 2 || vpf::trace_on()) vpf::note_kernel(#K);  \
    (void)hipGetLastError();                                             \
    hipLaunchKernelGGL(K, __VA_ARGS__);                                  \
  } while (0)

}  // namespace vpf
