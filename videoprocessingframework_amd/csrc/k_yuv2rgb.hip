// k_yuv2rgb.hip — NV12 / YUV420 / YUV444 -> RGB / BGR / RGB_PLANAR for gfx950 (MI355X).
//
// Replaces the NPP calls behind nv12_rgb / nv12_bgr / yuv420_rgb / yuv420_bgr / yuv444_rgb /
// yuv444_bgr (reference: src/TC/src/TasksColorCvt.cpp:53-108,122-182,322-369,383-430,444-550) and
// fuses nv12_rgb + rgb8_deinterleave (:1059-1088) into one pass for NV12 -> RGB_PLANAR.
//
// This is HBM-bound pointwise work (4.5 B/px: 1.5 read, 3 written), so the design is about memory
// instructions, not math:
//   * a wave owns a strip of one or more ROW PAIRS so each UV sample is read once and serves its
//     2x2 luma quad out of registers (no second trip to memory, no LDS needed for the stencil);
//   * p4 kernels: a lane owns 4 pixels -> 1 dword of Y per row, 1 dword of UV, and writes 12 B per
//     row.  A wave's loads are 256 contiguous bytes, its stores 768 contiguous bytes: every memory
//     instruction is fully coalesced with no cross-lane traffic at all;
//   * p16 kernels: a lane owns 16 pixels -> dwordx4 loads; the 48 B/row it produces are transposed
//     through a wave-private LDS tile so every global store is a dense 1 KiB dwordx4 wave store;
//   * math is fp32 FMA on v_cvt_f32_ubyteN operands, rounded + saturated + byte-packed by
//     v_cvt_pk_u8_f32: 1 cvt + 3 FMA + 3 pack ops per pixel;
//   * `BatchArgs` carries up to 32 frames in the kernarg segment, blockIdx.y selects the frame, so
//     one dispatch streams ~600 MB and the ~2 us kernel boundary is amortised.
#include "k_yuv2rgb_tasks.h"

namespace vpf {

// batched p4 entry (the 4-B-aligned / irregular-size fast path): 1 row pair per task, cvt_pk pack, non-temporal loads + stores
template <int SRC, int DST>
__global__ __launch_bounds__(256) void k_yuv420_rgb_p4(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w,
                                                       uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  yuv420_rgb_p4_task<SRC, DST, 1, 1, true, true, 0>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
template <int SRC, int DST>  // single-frame entry of the default p4 form (irregular sizes / alignments): scalar arguments
__global__ __launch_bounds__(256) void k_yuv420_rgb_p4_one(VPF_ONE_SRC_PARAMS, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks,
                                                           VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  yuv420_rgb_p4_task<SRC, DST, 1, 1, true, true, 0>(VPF_ONE_FRAME, c, w, h, chunks_x, n_tasks);
}

// batched p16 entry: up to 32 frame descriptors by value in the kernarg segment, blockIdx.y = frame.  NTS = non-temporal
// stores (false: allocating stores, VPF_EXEC_DST_REUSED); BALLAST_KB pads the LDS footprint so that 4 instead of 6 workgroups are
// resident per CU (+1-2 % on launches that run long enough, tools/write_probe.hip)
template <int DST, bool NTS, int BALLAST_KB, int SRC>
__global__ __launch_bounds__(256) void k_nv12_rgb_p16(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w,
                                                      uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  p16_task<DST, 1, true, NTS, true, 4, BALLAST_KB, false, SRC>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
// single-frame entry (one Execute() = one launch): the frame arrives as scalar kernel arguments, source side first, which
// the dispatcher preloads into SGPRs (-amdgpu-kernarg-preload-count): the wave's first loads no longer wait for a
// scalar-cache round trip to the kernarg segment — worth 0.5-0.8 us on a kernel that lasts 6-7 us
template <int DST, bool NTS, int SRC>
__global__ __launch_bounds__(256) void k_nv12_rgb_p16_one(VPF_ONE_SRC_PARAMS, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks,
                                                          VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  p16_task<DST, 1, true, NTS, true, 4, 0, false, SRC>(VPF_ONE_FRAME, c, w, h, chunks_x, n_tasks);
}

// the same conversion with the blocks numbered straight through the picture (p16x_task): the default for packed outputs of frames at least 1024 px wide
template <int DST, bool NTS, int BALLAST_KB, int SRC>
__global__ __launch_bounds__(256) void k_nv12_rgb_p16x(const BatchArgs args, const Yuv2RgbCoef c, uint32_t bpr, uint32_t n_blocks) {
  p16x_task<DST, NTS, BALLAST_KB, SRC>(args.f[blockIdx.y], c, bpr, n_blocks);
}
template <int DST, bool NTS, int SRC>
__global__ __launch_bounds__(256) void k_nv12_rgb_p16x_one(VPF_ONE_SRC_PARAMS, uint32_t bpr, uint32_t n_blocks, VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  p16x_task<DST, NTS, 0, SRC>(VPF_ONE_FRAME, c, bpr, n_blocks);
}

template <bool NTS, int SRC = FC_NV12>
__global__ __launch_bounds__(256) void k_nv12_planar_r16(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h,
                                                         uint32_t chunks_x, uint32_t n_tasks) {
  planar_r16_task<NTS, SRC>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
template <bool NTS, int SRC>  // single-frame entry: scalar arguments (see VPF_ONE_SRC_PARAMS)
__global__ __launch_bounds__(256) void k_nv12_planar_r16_one(VPF_ONE_SRC_PARAMS, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks,
                                                             VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  planar_r16_task<NTS, SRC>(VPF_ONE_FRAME, c, w, h, chunks_x, n_tasks);
}


// ---------------------------------------------------------------------------------------------
// YUV444 (three full planes) -> RGB/BGR/PLANAR, 4 px per lane, one row per task.
// Requires w % 4 == 0 and 4-byte aligned planes.
// ---------------------------------------------------------------------------------------------
template <int DST, int PACK>
__global__ __launch_bounds__(256) void k_yuv444_rgb_p4(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w,
                                                       uint32_t h, uint32_t groups_x) {
  const FrameDesc& f = args.f[blockIdx.z];
  const uint32_t gx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (gx >= groups_x || y >= h) return;
  const uint32_t x = gx * 4;
  const uint32_t yd = ldg<false, uint32_t>(f.s[0] + (size_t)y * f.sp[0] + x);
  const uint32_t ud = ldg<false, uint32_t>(f.s[1] + (size_t)y * f.sp[1] + x);
  const uint32_t vd = ldg<false, uint32_t>(f.s[2] + (size_t)y * f.sp[2] + x);
  const Chroma k0 = chroma_terms(c, ubyte<0>(ud), ubyte<0>(vd)), k1 = chroma_terms(c, ubyte<1>(ud), ubyte<1>(vd));
  const Chroma k2 = chroma_terms(c, ubyte<2>(ud), ubyte<2>(vd)), k3 = chroma_terms(c, ubyte<3>(ud), ubyte<3>(vd));
  Quad q;
  const float y0 = ubyte<0>(yd), y1 = ubyte<1>(yd), y2 = ubyte<2>(yd), y3 = ubyte<3>(yd);
  q.r[0] = __builtin_fmaf(y0, c.cy, k0.rc); q.g[0] = __builtin_fmaf(y0, c.cy, k0.gc); q.b[0] = __builtin_fmaf(y0, c.cy, k0.bc);
  q.r[1] = __builtin_fmaf(y1, c.cy, k1.rc); q.g[1] = __builtin_fmaf(y1, c.cy, k1.gc); q.b[1] = __builtin_fmaf(y1, c.cy, k1.bc);
  q.r[2] = __builtin_fmaf(y2, c.cy, k2.rc); q.g[2] = __builtin_fmaf(y2, c.cy, k2.gc); q.b[2] = __builtin_fmaf(y2, c.cy, k2.bc);
  q.r[3] = __builtin_fmaf(y3, c.cy, k3.rc); q.g[3] = __builtin_fmaf(y3, c.cy, k3.gc); q.b[3] = __builtin_fmaf(y3, c.cy, k3.bc);
  if constexpr (DST == FC_PLANAR) {
    stg<false, uint32_t>(f.d[0] + (size_t)y * f.dp[0] + x, pack4<PACK>(q.r[0], q.r[1], q.r[2], q.r[3]));
    stg<false, uint32_t>(f.d[1] + (size_t)y * f.dp[1] + x, pack4<PACK>(q.g[0], q.g[1], q.g[2], q.g[3]));
    stg<false, uint32_t>(f.d[2] + (size_t)y * f.dp[2] + x, pack4<PACK>(q.b[0], q.b[1], q.b[2], q.b[3]));
  } else {
    uint32_t d0, d1, d2;
    pack_rgb12<DST, PACK>(q, d0, d1, d2);
    stg3<false>(f.d[0] + (size_t)y * f.dp[0] + 3 * (size_t)x, d0, d1, d2);
  }
}

// YUV444 -> RGB/BGR/PLANAR, r16: one row x 1024 px per wave, three dense 1-KiB plane loads, three dense 1-KiB stores
// (packed outputs through store_run48).  Requires w % 16 == 0 and 16-B aligned planes / pitches.
template <int DST>
VPF_DEV void yuv444_rgb_r16_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  __shared__ u32x4 tile[DST == FC_PLANAR ? 1 : 4 * 192];
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = blockIdx.x * 4 + wv;
  if (wt >= n_tasks) return;
  const uint32_t y = wt / chunks_x, chunk = wt - y * chunks_x;
  const uint32_t x = chunk * 1024 + lane * 16;
  if (DST == FC_PLANAR && x >= w) return;
  const uint32_t xc = x < w ? x : w - 16;  // clamped lanes compute a duplicate that store_run48 never writes
  const u32x4 yq = ldg<true, u32x4>(f.s[0] + (size_t)y * f.sp[0] + xc);
  const u32x4 uq = ldg<true, u32x4>(f.s[1] + (size_t)y * f.sp[1] + xc);
  const u32x4 vq = ldg<true, u32x4>(f.s[2] + (size_t)y * f.sp[2] + xc);
  uint32_t o[12];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t yd = yq[j], ud = uq[j], vd = vq[j];
    const Chroma k0 = chroma_terms(c, ubyte<0>(ud), ubyte<0>(vd)), k1 = chroma_terms(c, ubyte<1>(ud), ubyte<1>(vd));
    const Chroma k2 = chroma_terms(c, ubyte<2>(ud), ubyte<2>(vd)), k3 = chroma_terms(c, ubyte<3>(ud), ubyte<3>(vd));
    Quad q;
    const float y0 = ubyte<0>(yd), y1 = ubyte<1>(yd), y2 = ubyte<2>(yd), y3 = ubyte<3>(yd);
    q.r[0] = __builtin_fmaf(y0, c.cy, k0.rc); q.g[0] = __builtin_fmaf(y0, c.cy, k0.gc); q.b[0] = __builtin_fmaf(y0, c.cy, k0.bc);
    q.r[1] = __builtin_fmaf(y1, c.cy, k1.rc); q.g[1] = __builtin_fmaf(y1, c.cy, k1.gc); q.b[1] = __builtin_fmaf(y1, c.cy, k1.bc);
    q.r[2] = __builtin_fmaf(y2, c.cy, k2.rc); q.g[2] = __builtin_fmaf(y2, c.cy, k2.gc); q.b[2] = __builtin_fmaf(y2, c.cy, k2.bc);
    q.r[3] = __builtin_fmaf(y3, c.cy, k3.rc); q.g[3] = __builtin_fmaf(y3, c.cy, k3.gc); q.b[3] = __builtin_fmaf(y3, c.cy, k3.bc);
    if constexpr (DST == FC_PLANAR) {
      o[j] = pack4<1>(q.r[0], q.r[1], q.r[2], q.r[3]);
      o[4 + j] = pack4<1>(q.g[0], q.g[1], q.g[2], q.g[3]);
      o[8 + j] = pack4<1>(q.b[0], q.b[1], q.b[2], q.b[3]);
    } else {
      pack_rgb12<DST, 1>(q, o[3 * j], o[3 * j + 1], o[3 * j + 2]);
    }
  }
  if constexpr (DST == FC_PLANAR) {
#pragma unroll
    for (int k = 0; k < 3; k++)
      stg<true, u32x4>(f.d[k] + (size_t)y * f.dp[k] + x, u32x4{o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]});
  } else {
    store_run48(tile + wv * 192, f.d[0] + (size_t)y * f.dp[0], chunk * 3072, 3 * w, lane, o);
  }
}
template <int DST>
__global__ __launch_bounds__(256) void k_yuv444_rgb_r16(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  yuv444_rgb_r16_task<DST>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
template <int DST>  // single-frame entry: scalar arguments (see VPF_ONE_SRC_PARAMS in vpf_internal.h)
__global__ __launch_bounds__(256) void k_yuv444_rgb_r16_one(VPF_ONE_SRC_PARAMS, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks, VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  yuv444_rgb_r16_task<DST>(VPF_ONE_FRAME, c, w, h, chunks_x, n_tasks);
}

// ---------------------------------------------------------------------------------------------
// generic: any size, any alignment.  One thread per 2x2 quad, byte accesses, full bounds checks.
// ---------------------------------------------------------------------------------------------
template <int SRC, int DST>
__global__ __launch_bounds__(256) void k_yuv_rgb_generic(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w,
                                                         uint32_t h) {
  const FrameDesc& f = args.f[blockIdx.z];
  const uint32_t qx = blockIdx.x * 64 + (threadIdx.x & 63), qy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const uint32_t x0 = 2 * qx, y0 = 2 * qy;
  if (x0 >= w || y0 >= h) return;
  float u = 0, v = 0;
  if constexpr (SRC == FC_NV12) {
    const uint8_t* p = f.s[1] + (size_t)qy * f.sp[1] + x0;
    u = p[0]; v = p[1];
  } else if constexpr (SRC == FC_YUV420) {
    u = f.s[1][(size_t)qy * f.sp[1] + qx]; v = f.s[2][(size_t)qy * f.sp[2] + qx];
  }
  Chroma k = chroma_terms(c, u, v);
  for (uint32_t dy = 0; dy < 2; dy++)
    for (uint32_t dx = 0; dx < 2; dx++) {
      const uint32_t x = x0 + dx, y = y0 + dy;
      if (x >= w || y >= h) continue;
      if constexpr (SRC == FC_YUV444) {
        k = chroma_terms(c, (float)f.s[1][(size_t)y * f.sp[1] + x], (float)f.s[2][(size_t)y * f.sp[2] + x]);
      }
      const float yf = (float)f.s[0][(size_t)y * f.sp[0] + x];
      const uint8_t r = (uint8_t)sat_rne(__builtin_fmaf(yf, c.cy, k.rc));
      const uint8_t g = (uint8_t)sat_rne(__builtin_fmaf(yf, c.cy, k.gc));
      const uint8_t b = (uint8_t)sat_rne(__builtin_fmaf(yf, c.cy, k.bc));
      if constexpr (DST == FC_PLANAR) {
        f.d[0][(size_t)y * f.dp[0] + x] = r; f.d[1][(size_t)y * f.dp[1] + x] = g; f.d[2][(size_t)y * f.dp[2] + x] = b;
      } else {
        uint8_t* o = f.d[0] + (size_t)y * f.dp[0] + 3 * (size_t)x;
        o[0] = (DST == FC_BGR) ? b : r; o[1] = g; o[2] = (DST == FC_BGR) ? r : b;
      }
    }
}


// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool aligned_all(const BatchArgs& a, uint32_t n, int nsrc, int ndst, uint32_t src_al, uint32_t dst_al,
                        uint32_t chroma_al) {
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < nsrc; k++) {
      uint32_t al = (k == 0) ? src_al : chroma_al;
      if (((uintptr_t)a.f[i].s[k] | a.f[i].sp[k]) & (al - 1)) return false;
    }
    for (int k = 0; k < ndst; k++)
      if (((uintptr_t)a.f[i].d[k] | a.f[i].dp[k]) & (dst_al - 1)) return false;
  }
  return true;
}

// Kernel selection for 4:2:0 sources.  `variant` is the tuning hint (include/vpf_hip.h): 0 = policy below; a named kernel
// (4 / 8 / 12 / 30 / 37 / 44 / 45 / 46) is honoured where it applies and falls back down the same chain where it does not; 40 = the
// narrower p4 path, 9 = the any-input generic kernel.  Every kernel here writes the same pixels.
//   p16x (45 | 46: 4 workgroups / CU)                                               packed outputs, w >= 1024: blocks numbered straight through the picture
//   p16  (8: non-temporal stores | 12: allocating stores | 30: 4 workgroups / CU)   packed outputs; w % 16 == 0, h even, 16-B aligned
//   r16  (37: non-temporal | 44: allocating stores)                                 planar outputs; same conditions
//        (the straight numbering applied to r16 measured 1 % SLOWER on the 1080p planar batch, 0.779 vs 0.787 in same-box A/B: not kept)
//   p4   (4)                                                                        4-B aligned planes, any size
//   generic (9)                                                                     anything
// Policy (profiles/r01_bench_sweep.log, r02_bench_sweep.log, tools/lab): packed -> p16x (p16 below 1024 px), with the 4-workgroup cap when the launch is a batch (>= 4 frames:
// a narrower chip-wide write frontier is worth 1-2 %; short single-frame launches want all the waves they can get); planar -> r16
// (three 1-KiB plane stores per wave), except a lone frame of >= 3 Mpx where the row-pair p16 form wins (kernel 7.3 vs 7.9 us at 4K,
// but 4.3 vs 3.6 at 1080p: tools/gpu_planar_single.sh).  The alternatives that were measured and dropped (more row pairs per task,
// one-store-per-wave forms, XCD swizzles, other occupancy caps, byte-stream stores) live in tools/lab/k_lab.hip.
template <int SRC, int DST>
static hipError_t launch_420(hipStream_t st, const Yuv2RgbCoef& c, uint32_t w, uint32_t h, uint32_t n,
                             const BatchArgs& a, int variant) {
  const int nsrc = (SRC == FC_NV12) ? 2 : 3, ndst = (DST == FC_PLANAR) ? 3 : 1;
  const bool even = (w % 4 == 0) && (h % 2 == 0);
  const bool p16_ok = even && (w % 16 == 0) && aligned_all(a, n, nsrc, ndst, 16, 16, SRC == FC_NV12 ? 16 : 8);
  const bool p4_ok = aligned_all(a, n, nsrc, ndst, 4, 4, SRC == FC_NV12 ? 4 : 2);
  const bool big_single = n < 4 && (size_t)w * h >= (size_t)3 << 20;
  const bool p16x_ok = p16_ok && DST != FC_PLANAR && w >= 1024 && (uint64_t)(w / 16) * (h / 2) < (1u << 30);
  // values of the shared tuning key that name no NV12 / YUV420 -> RGB kernel (43 = a resize-only hint, anything unknown) mean the default policy
  if (variant != 4 && variant != 8 && variant != 9 && variant != 12 && variant != 30 && variant != 37 && variant != 40 && variant != 44 && variant != 45 && variant != 46) variant = 0;
  if (variant == 0) variant = p16_ok ? (DST == FC_PLANAR ? (big_single ? 8 : 37) : (n >= 4 ? (p16x_ok ? 46 : 30) : 8)) : 4;  // (single frames: p16x measured level with p16, 0.710 vs 0.717)
  if ((variant == 45 || variant == 46) && !p16x_ok) variant = variant == 46 ? 30 : 8;  // narrow frames / planar outputs: the chunk-per-row form
  const bool is_p16 = variant == 8 || variant == 12 || variant == 30, is_r16 = variant == 37 || variant == 44;
  if ((is_p16 || is_r16) && !p16_ok) variant = 4;
  if (is_r16 && DST != FC_PLANAR) variant = 4;
  if (variant != 9 && !p4_ok) variant = 9;  // p16_ok implies p4_ok
  const FrameDesc& f0 = a.f[0];
  if (variant == 37 || variant == 44) {
    const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * h;  // one task per row per chunk (h even)
    dim3 grid((tasks + 3) / 4, n);
    if (n == 1 && variant == 37) VPF_LAUNCH((k_nv12_planar_r16_one<true, SRC>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
    else if (n == 1) VPF_LAUNCH((k_nv12_planar_r16_one<false, SRC>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
    else if (variant == 37) VPF_LAUNCH((k_nv12_planar_r16<true, SRC>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
    else VPF_LAUNCH((k_nv12_planar_r16<false, SRC>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);  // allocating stores
    return hipGetLastError();
  }
  if (variant == 45 || variant == 46) {
    if constexpr (DST != FC_PLANAR) {
      const uint32_t bpr = w / 16, nb = bpr * (h / 2);
      dim3 grid((nb + 255) / 256, n);
      if (n == 1) VPF_LAUNCH((k_nv12_rgb_p16x_one<DST, true, SRC>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), bpr, nb, VPF_ONE_DST_ARGS(f0), c);
      else if (variant == 45) VPF_LAUNCH((k_nv12_rgb_p16x<DST, true, 0, SRC>), grid, dim3(256), 0, st, a, c, bpr, nb);
      else VPF_LAUNCH((k_nv12_rgb_p16x<DST, true, 16, SRC>), grid, dim3(256), 0, st, a, c, bpr, nb);  // 40 KiB of LDS -> 4 workgroups / CU
      return hipGetLastError();
    }
  }
  if (variant == 8 || variant == 12 || variant == 30) {
    const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * (h / 2);
    dim3 grid((tasks + 3) / 4, n);
    if (n == 1 && variant != 12) VPF_LAUNCH((k_nv12_rgb_p16_one<DST, true, SRC>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
    else if (n == 1) VPF_LAUNCH((k_nv12_rgb_p16_one<DST, false, SRC>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
    else if (variant == 8) VPF_LAUNCH((k_nv12_rgb_p16<DST, true, 0, SRC>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
    else if (variant == 12) VPF_LAUNCH((k_nv12_rgb_p16<DST, false, 0, SRC>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
    else VPF_LAUNCH((k_nv12_rgb_p16<DST, true, 16, SRC>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);  // 40 KiB of LDS -> 4 workgroups / CU
    return hipGetLastError();
  }
  if (variant != 9) {  // 4, 40 and every fallback
    const uint32_t chunks = ((w + 3) / 4 + 63) / 64, tasks = chunks * ((h + 1) / 2);
    dim3 grid((tasks + 3) / 4, n);
    if (n == 1) VPF_LAUNCH((k_yuv420_rgb_p4_one<SRC, DST>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
    else VPF_LAUNCH((k_yuv420_rgb_p4<SRC, DST>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
    return hipGetLastError();
  }
  dim3 grid(((w + 1) / 2 + 63) / 64, ((h + 1) / 2 + 3) / 4, n);
  VPF_LAUNCH((k_yuv_rgb_generic<SRC, DST>), grid, dim3(256), 0, st, a, c, w, h);
  return hipGetLastError();
}

template <int DST>
static hipError_t launch_444(hipStream_t st, const Yuv2RgbCoef& c, uint32_t w, uint32_t h, uint32_t n,
                             const BatchArgs& a, int variant) {
  const int ndst = (DST == FC_PLANAR) ? 3 : 1;
  if (variant != 9 && variant != 40 && w % 16 == 0 && aligned_all(a, n, 3, ndst, 16, 16, 16)) {
    const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * h;
    if (n == 1) VPF_LAUNCH((k_yuv444_rgb_r16_one<DST>), dim3((tasks + 3) / 4, n), dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), w, h, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c);
    else VPF_LAUNCH((k_yuv444_rgb_r16<DST>), dim3((tasks + 3) / 4, n), dim3(256), 0, st, a, c, w, h, chunks, tasks);
    return hipGetLastError();
  }
  if (variant != 9 && w % 4 == 0 && aligned_all(a, n, 3, ndst, 4, 4, 4) && h <= 65535) {
    dim3 grid((w / 4 + 255) / 256, h, n);
    VPF_LAUNCH((k_yuv444_rgb_p4<DST, 1>), grid, dim3(256), 0, st, a, c, w, h, w / 4);
    return hipGetLastError();
  }
  dim3 grid(((w + 1) / 2 + 63) / 64, ((h + 1) / 2 + 3) / 4, n);
  VPF_LAUNCH((k_yuv_rgb_generic<FC_YUV444, DST>), grid, dim3(256), 0, st, a, c, w, h);
  return hipGetLastError();
}

hipError_t launch_yuv_to_rgb(hipStream_t st, int src_fc, int dst_fc, const Yuv2RgbCoef& c, uint32_t w, uint32_t h,
                             uint32_t n, const BatchArgs& a, int variant, bool dst_reused) {
  // VPF_EXEC_DST_REUSED: single-frame launches keep their output in the Infinity Cache (allocating stores) for the next
  // kernel of the chain: variant 12 = p16 with non-temporal loads only, 44 = planar r16 with plain stores
  if (variant == 0 && dst_reused && n < 4 && (src_fc == FC_NV12 || src_fc == FC_YUV420)) variant = (dst_fc == FC_PLANAR) ? 44 : 12;
#define VPF_DST_SWITCH(FN, ...)                                                   \
  switch (dst_fc) {                                                               \
    case FC_RGB: return FN<__VA_ARGS__ FC_RGB>(st, c, w, h, n, a, variant);       \
    case FC_BGR: return FN<__VA_ARGS__ FC_BGR>(st, c, w, h, n, a, variant);       \
    case FC_PLANAR: return FN<__VA_ARGS__ FC_PLANAR>(st, c, w, h, n, a, variant); \
    default: return hipErrorInvalidValue;                                         \
  }
  switch (src_fc) {
    case FC_NV12: VPF_DST_SWITCH(launch_420, FC_NV12, )
    case FC_YUV420: VPF_DST_SWITCH(launch_420, FC_YUV420, )
    case FC_YUV444: VPF_DST_SWITCH(launch_444, )
    default: return hipErrorInvalidValue;
  }
#undef VPF_DST_SWITCH
}

}  // namespace vpf