// tools/lab/k_lab.hip — MEASUREMENT LAB, not product.  Builds into tools/lab/libvpfhip_lab.so (tools/lab/build_lab.py); nothing in
// videoprocessingframework_amd/ includes, links or loads it, and libvpfhip.so contains none of these kernels.
//
// Round 1 located the headline converter's ceiling by trying ~40 forms of the NV12 -> RGB kernel (profiles/r01_bench_sweep.log,
// r01_write_probe.txt).  The forms that lost, and the bandwidth probes that bracket the ceiling, are kept here so the numbers stay
// reproducible (`python bench.py --sweep`):
//   1,2,3 / 5,6   p4 with 1,2,4 row pairs per task, plain / non-temporal accesses        10   p4 with the explicit (non cvt_pk) pack
//   14,16         p4 RP1 / RP2 with non-temporal stores only                             33-35 p4 with 6 / 5 / 4 workgroups per CU
//   7,11,13       p16: plain accesses / NT stores only / lane-strided stores without the LDS transpose
//   17,18,19      p16r: one LDS row tile per wave, 1 / 2 / 4 row pairs per task          20,21 p16 with 1 / 2 waves per workgroup
//   31,32,36      p16 capped at 3 / 2 / 5 workgroups per CU                              41,42 p16 (4-workgroup cap / plain) + XCD swizzle
//   27,28,29      r4 / b4: one store instruction per wave                                38   r16 packed: one row per wave, 3 stores
//   43            s16: the output row as a byte stream, one 1-KiB store per wave, no LDS
//   15,22-26      BANDWIDTH PROBES — NOT conversions, they write garbage: 15 = p16's loads + LDS transpose + stores with the arithmetic
//                 removed; 22 = its loads only; 23 / 24 = its stores only (NT / plain); 25 / 26 = linear fill (NT / plain)
// Every non-probe variant writes exactly the product's pixels (tests/test_gpu_lab.py).
#include <cstring>

#include "k_yuv2rgb_tasks.h"
#include "vpf_coef.h"

namespace vpf {

// ---------------------------------------------------------------------------------------------
// p16r: p16 for packed outputs with RPW row pairs per wave task (all 3*RPW loads in flight first) and ONE 3 KiB LDS
// row tile per wave reused for every output row (12 KiB per block -> 8 blocks/CU instead of 6).  LDS operations of a
// wave execute in order, so the next row's ds_write cannot overtake the previous row's ds_read.
// ---------------------------------------------------------------------------------------------
template <int DST, int RPW, bool NTS>
__global__ __launch_bounds__(256) void k_nv12_rgb_p16r(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h,
                                                       uint32_t chunks_x, uint32_t n_tasks) {
  __shared__ u32x4 tile[4 * 192];
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = blockIdx.x * 4 + wv;
  if (wt >= n_tasks) return;
  const FrameDesc f = args.f[blockIdx.y];
  const uint32_t rpg = wt / chunks_x, chunk = wt - rpg * chunks_x;
  const uint32_t x = chunk * 1024 + lane * 16;
  const bool act = x < w;
  const uint32_t nrp = h >> 1;
  u32x4 y[RPW][2], uv[RPW];
#pragma unroll
  for (int r = 0; r < RPW; r++) {
    const uint32_t rp = rpg * RPW + r;
    if (act && rp < nrp) {
      y[r][0] = ldg<false, u32x4>(f.s[0] + (size_t)(2 * rp) * f.sp[0] + x);
      y[r][1] = ldg<false, u32x4>(f.s[0] + (size_t)(2 * rp + 1) * f.sp[0] + x);
      uv[r] = ldg<false, u32x4>(f.s[1] + (size_t)rp * f.sp[1] + x);
    }
  }
  u32x4* t = tile + wv * 192;
  const uint32_t row_bytes = 3 * w;
#pragma unroll
  for (int r = 0; r < RPW; r++) {
    const uint32_t rp = rpg * RPW + r;
    if (rp >= nrp) break;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (act) {
        uint32_t o[12];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const Chroma k0 = chroma_terms(c, ubyte<0>(uv[r][j]), ubyte<1>(uv[r][j]));
          const Chroma k1 = chroma_terms(c, ubyte<2>(uv[r][j]), ubyte<3>(uv[r][j]));
          const Quad q = convert4(c, y[r][half][j], k0, k1);
          pack_rgb12<DST, 1>(q, o[3 * j], o[3 * j + 1], o[3 * j + 2]);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) t[lane * 3 + j] = u32x4{o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint8_t* rowp = f.d[0] + (size_t)(2 * rp + half) * f.dp[0];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const uint32_t off = chunk * 3072 + (k * 64 + lane) * 16;
        if (off < row_bytes) stg<NTS, u32x4>(rowp + off, t[k * 64 + lane]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}


// ---------------------------------------------------------------------------------------------
// r4 / b4: ONE STORE INSTRUCTION PER WAVE.  tools/write_probe.hip shows the HBM write rate on gfx950 depends on how
// many store instructions a wave issues before it retires: 1 x 1 KiB per wave 6.8 TB/s, 2 -> 5.9, 3 -> 5.7, 6 -> 5.3
// (256-thread blocks, linear buffer).  Writes are 2/3 of this converter's traffic, so these kernels give every wave a
// single store:
//   r4: a lane owns 4 px of ONE row (Y dword + the UV dword it shares with the lane one row below, which another wave of
//       the same block reads too: the second read is an L1/L2 hit, not HBM) -> one 768-B dwordx3 wave store.
//   b4: same compute, but the block's 2 rows x 512 px (3 KiB) are gathered in LDS and leave as three dense 1-KiB dwordx4
//       wave stores (waves 0-2; wave 3 stores nothing).
// Require w % 4 == 0, h even, 4-B (r4) / 16-B (b4) aligned planes.  NV12 source, packed RGB/BGR destination.
// ---------------------------------------------------------------------------------------------
template <int DST, bool LDS_T, bool NTS>
__global__ __launch_bounds__(256) void k_nv12_rgb_r4(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h,
                                                     uint32_t tiles_x, uint32_t n_tiles) {
  __shared__ uint32_t lds[LDS_T ? 768 : 1];  // [row A: 1536 B][row B: 1536 B]
  const uint32_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  const FrameDesc f = args.f[blockIdx.y];
  const uint32_t rp = tile / tiles_x, tx = tile - rp * tiles_x;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t half = wv >> 1;                          // waves 0,1 -> row 2rp ; waves 2,3 -> row 2rp+1
  const uint32_t px = ((wv & 1) * 64 + lane) * 4;         // pixel offset inside the 512-px tile
  const uint32_t x = tx * 512 + px;
  const bool act = x < w;
  uint32_t d0 = 0, d1 = 0, d2 = 0;
  if (act) {
    const uint32_t yd = ldg<true, uint32_t>(f.s[0] + (size_t)(2 * rp + half) * f.sp[0] + x);
    const uint32_t uv = ldg<false, uint32_t>(f.s[1] + (size_t)rp * f.sp[1] + x);  // read by two waves: keep it cacheable
    const Chroma k0 = chroma_terms(c, ubyte<0>(uv), ubyte<1>(uv)), k1 = chroma_terms(c, ubyte<2>(uv), ubyte<3>(uv));
    pack_rgb12<DST, 1>(convert4(c, yd, k0, k1), d0, d1, d2);
  }
  if constexpr (!LDS_T) {
    if (act) stg3<NTS>(f.d[0] + (size_t)(2 * rp + half) * f.dp[0] + 3 * (size_t)x, d0, d1, d2);
  } else {
    uint32_t* t = lds + half * 384 + (px >> 2) * 3;      // 12 B per lane, lane stride 3 dwords: conflict free
    t[0] = d0; t[1] = d1; t[2] = d2;
    __syncthreads();
    if (wv < 3) {
      const uint32_t o = (wv * 64 + lane) * 16;            // byte offset in the 3 KiB block image
      const u32x4 v = reinterpret_cast<const u32x4*>(lds)[wv * 64 + lane];
      const uint32_t r = o >= 1536, col = tx * 1536 + (o - r * 1536);
      if (col < 3 * w) stg<NTS, u32x4>(f.d[0] + (size_t)(2 * rp + r) * f.dp[0] + col, v);
    }
  }
}


// r16 for packed outputs: one row x 1024 px per wave (Y + the UV line it shares with its neighbour row), the 48 B/lane
// transposed through a wave-private 3 KiB LDS tile -> three dense 1-KiB stores per wave (p16 issues six).
template <int DST, bool NTS, int BALLAST_KB>
__global__ __launch_bounds__(256) void k_nv12_rgb_r16(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h,
                                                      uint32_t chunks_x, uint32_t n_tasks) {
  __shared__ u32x4 tile[4 * 192 + BALLAST_KB * 64];
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = blockIdx.x * 4 + wv;
  if (wt >= n_tasks) return;
  const FrameDesc f = args.f[blockIdx.y];
  const uint32_t pair = wt >> 1, half = wt & 1;
  const uint32_t rp = pair / chunks_x, chunk = pair - rp * chunks_x;
  const uint32_t x = chunk * 1024 + lane * 16;
  const uint32_t y = 2 * rp + half;
  if (y >= h) return;
  const bool act = x < w;
  u32x4* t = tile + wv * 192;
  if (act) {
    const u32x4 yq = ldg<true, u32x4>(f.s[0] + (size_t)y * f.sp[0] + x);
    const u32x4 uv = ldg<false, u32x4>(f.s[1] + (size_t)rp * f.sp[1] + x);
    uint32_t o[12];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const Chroma k0 = chroma_terms(c, ubyte<0>(uv[j]), ubyte<1>(uv[j])), k1 = chroma_terms(c, ubyte<2>(uv[j]), ubyte<3>(uv[j]));
      pack_rgb12<DST, 1>(convert4(c, yq[j], k0, k1), o[3 * j], o[3 * j + 1], o[3 * j + 2]);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) t[lane * 3 + j] = u32x4{o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint8_t* rowp = f.d[0] + (size_t)y * f.dp[0];
  const uint32_t row_bytes = 3 * w;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t off = chunk * 3072 + (k * 64 + lane) * 16;
    if (off < row_bytes) stg<NTS, u32x4>(rowp + off, t[k * 64 + lane]);
  }
}

// s16 ("stream"): ONE 1-KiB store per wave with no LDS and no cross-lane traffic.  The packed output row is treated as a
// byte stream: a wave owns 1 KiB of it, a lane owns 16 B = 5 1/3 pixels.  The lane converts the 8 pixels starting at the even
// pixel that contains its first byte (Y and UV arrive as one 12-B load each from the same 4-B aligned column; v_alignbyte_b32
// drops the 0 or 2 leading bytes), packs 24 bytes and funnels out its 16 with v_alignbyte_b32 by (byte offset mod 3-ish).
// 1.5x the arithmetic of p16 (8 px converted per 5.33 px stored) buys the best store geometry of tools/write_probe.hip.

// Requires w % 16 == 0, h even, 16-B aligned destination rows, 4-B aligned source rows.
template <int DST, bool NTS>
__global__ __launch_bounds__(256) void k_nv12_rgb_s16(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h,
                                                      uint32_t segs, uint32_t n_tasks) {
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = blockIdx.x * 4 + wv;
  if (wt >= n_tasks) return;
  const FrameDesc f = args.f[blockIdx.y];
  const uint32_t pair = wt >> 1, half = wt & 1;  // the two rows that share a UV row sit in neighbouring waves of one block
  const uint32_t rp = pair / segs, seg = pair - rp * segs;
  const uint32_t y = 2 * rp + half;
  const uint32_t B = seg * 1024 + lane * 16;  // first output byte of the lane
  if (B >= 3 * w) return;
  const uint32_t p0 = (uint32_t)(((uint64_t)B * 0xAAAAAAABull) >> 33);  // B / 3
  const uint32_t pe = p0 & ~1u, s = B - 3 * pe;                        // even pixel holding byte B; s in [0, 5]
  const uint32_t col = pe & ~3u, sh = pe & 3u;                         // 4-B aligned source column, 0 or 2 bytes to drop
  const uint8_t* yr = f.s[0] + (size_t)y * f.sp[0];
  const uint8_t* ur = f.s[1] + (size_t)rp * f.sp[1];
  uint32_t yd[3], ud[3];
  if (col + 12 <= w) {
#pragma unroll
    for (int i = 0; i < 3; i++) { yd[i] = ldg<true, uint32_t>(yr + col + 4 * i); ud[i] = ldg<false, uint32_t>(ur + col + 4 * i); }
  } else {  // right edge: a dword past column w belongs to pixels that do not exist; never read it (tight pitch, last row)
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const uint32_t a = (col + 4 * i + 4 <= w) ? col + 4 * i : w - 4;
      yd[i] = ldg<true, uint32_t>(yr + a); ud[i] = ldg<false, uint32_t>(ur + a);
    }
  }
  const uint32_t y_lo = __builtin_amdgcn_alignbyte(yd[1], yd[0], sh), y_hi = __builtin_amdgcn_alignbyte(yd[2], yd[1], sh);
  const uint32_t u_lo = __builtin_amdgcn_alignbyte(ud[1], ud[0], sh), u_hi = __builtin_amdgcn_alignbyte(ud[2], ud[1], sh);
  uint32_t o[6];
  {
    const Chroma k0 = chroma_terms(c, ubyte<0>(u_lo), ubyte<1>(u_lo)), k1 = chroma_terms(c, ubyte<2>(u_lo), ubyte<3>(u_lo));
    pack_rgb12<DST, 1>(convert4(c, y_lo, k0, k1), o[0], o[1], o[2]);
    const Chroma k2 = chroma_terms(c, ubyte<0>(u_hi), ubyte<1>(u_hi)), k3 = chroma_terms(c, ubyte<2>(u_hi), ubyte<3>(u_hi));
    pack_rgb12<DST, 1>(convert4(c, y_hi, k2, k3), o[3], o[4], o[5]);
  }
  const uint32_t s4 = s & 3u;
  uint32_t a[5];
#pragma unroll
  for (int i = 0; i < 5; i++) a[i] = __builtin_amdgcn_alignbyte(o[i + 1], o[i], s4);
  const bool hi = s >= 4;
  const u32x4 v = {hi ? a[1] : a[0], hi ? a[2] : a[1], hi ? a[3] : a[2], hi ? a[4] : a[3]};
  stg<NTS, u32x4>(f.d[0] + (size_t)y * f.dp[0] + B, v);
}


// ---------------------------------------------------------------------------------------------
// bandwidth probes with the p16 geometry (NOT conversions; reachable only through the tuning hook, used by
// bench.py --sweep to locate the ceilings): MODE 0 = the loads only (one dword per wave stored so they are not
// dead), MODE 1 = the stores only.
// ---------------------------------------------------------------------------------------------
// p16 with the frame's 16-px x 2-row blocks numbered straight through the picture: a wave takes 64 consecutive blocks wherever the row
// ends (3840 px = 240 blocks per row pair = 3.75 waves: the product kernel leaves a quarter of every fourth wave idle).  Loads are per
// lane; the LDS-transposed stores compute the destination of every 16-B unit from the lane that produced it.  Requires w % 16 == 0.
template <int DST, int BALLAST_KB>
__global__ __launch_bounds__(256) void k_nv12_rgb_p16x(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h, uint32_t bpr /* blocks per row pair */,
                                                        uint32_t n_blocks) {
  __shared__ u32x4 tile[4 * 2 * 192 + BALLAST_KB * 64];
  const FrameDesc& f = args.f[blockIdx.y];
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t b0 = (blockIdx.x * 4 + wv) * 64;
  if (b0 >= n_blocks) return;
  const uint32_t blk = b0 + lane;
  const bool act = blk < n_blocks;
  const uint32_t rp = blk / bpr, x = (blk - rp * bpr) * 16;
  u32x4 y[2], uv;
  if (act) {
    y[0] = ldg<true, u32x4>(f.s[0] + (size_t)(2 * rp) * f.sp[0] + x);
    y[1] = ldg<true, u32x4>(f.s[0] + (size_t)(2 * rp + 1) * f.sp[0] + x);
    uv = ldg<true, u32x4>(f.s[1] + (size_t)rp * f.sp[1] + x);
  }
  uint32_t o[2][12];
  if (act) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const Chroma k0 = chroma_terms(c, ubyte<0>(uv[j]), ubyte<1>(uv[j]));
      const Chroma k1 = chroma_terms(c, ubyte<2>(uv[j]), ubyte<3>(uv[j]));
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const Quad q = convert4(c, y[half][j], k0, k1);
        pack_rgb12<DST, 1>(q, o[half][3 * j], o[half][3 * j + 1], o[half][3 * j + 2]);
      }
    }
  }
#pragma unroll
  for (int half = 0; half < 2; half++) {
    u32x4* t = tile + (wv * 2 + half) * 192;
    if (act) {
#pragma unroll
      for (int j = 0; j < 3; j++) t[lane * 3 + j] = u32x4{o[half][4 * j], o[half][4 * j + 1], o[half][4 * j + 2], o[half][4 * j + 3]};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const uint32_t idx = k * 64 + lane, src_lane = idx / 3, part = idx - 3 * src_lane;  // 16-B unit idx came from lane idx / 3
      const uint32_t ublk = b0 + src_lane;
      if (ublk < n_blocks) {
        const uint32_t urp = ublk / bpr, ux = (ublk - urp * bpr) * 16;
        stg<true, u32x4>(f.d[0] + (size_t)(2 * urp + half) * f.dp[0] + 3 * (size_t)ux + 16 * part, t[idx]);
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe_p16(const BatchArgs args, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = blockIdx.x * 4 + wv;
  if (wt >= n_tasks) return;
  const FrameDesc f = args.f[blockIdx.y];
  const uint32_t rp = wt / chunks_x, chunk = wt - rp * chunks_x;
  const uint32_t x = chunk * 1024 + lane * 16;
  if constexpr (MODE == 0) {
    if (x >= w) return;
    const u32x4 a = ldg<true, u32x4>(f.s[0] + (size_t)(2 * rp) * f.sp[0] + x);
    const u32x4 b = ldg<true, u32x4>(f.s[0] + (size_t)(2 * rp + 1) * f.sp[0] + x);
    const u32x4 c = ldg<true, u32x4>(f.s[1] + (size_t)rp * f.sp[1] + x);
    const uint32_t r = a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3];
    if (r == 0x12345678u) f.d[0][(size_t)(2 * rp) * f.dp[0] + 3 * x] = 1;  // practically never: keeps the loads alive
  } else if constexpr (MODE == 1 || MODE == 2) {  // the converter's store geometry; NT (1) or plain (2) stores
    const uint32_t row_bytes = 3 * w;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      uint8_t* rowp = f.d[0] + (size_t)(2 * rp + half) * f.dp[0];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const uint32_t off = chunk * 3072 + (k * 64 + lane) * 16;
        if (off < row_bytes) stg<MODE == 1, u32x4>(rowp + off, u32x4{off, rp, lane, (uint32_t)k});
      }
    }
  } else {  // MODE 3 / 4: linear fill of the frame (needs pitch == row bytes): wave t writes 6 KiB at t * 6 KiB; NT (3) / plain (4)
    const size_t frame_bytes = (size_t)3 * w * h;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const size_t off = (size_t)wt * 6144 + (size_t)(k * 64 + lane) * 16;
      if (off < frame_bytes) stg<MODE == 3, u32x4>(f.d[0] + off, u32x4{(uint32_t)off, rp, lane, (uint32_t)k});
    }
  }
}


// the product's task bodies under other memory-policy / occupancy parameters
template <int DST, int RP, int PACK, bool NTL, bool NTS, int BALLAST_KB>
__global__ __launch_bounds__(256) void k_lab_p4(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  yuv420_rgb_p4_task<FC_NV12, DST, RP, PACK, NTL, NTS, BALLAST_KB>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
template <int DST, bool NTL, bool NTS, bool LDS_T, int WPB, int BALLAST_KB, bool XCD_SWZ>
__global__ __launch_bounds__(64 * WPB) void k_lab_p16(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  p16_task<DST, 1, NTL, NTS, LDS_T, WPB, BALLAST_KB, XCD_SWZ, FC_NV12>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
// probe 15: p16's loads, LDS transpose and stores with the arithmetic removed (NOT a conversion)
__global__ __launch_bounds__(256) void k_probe_nomath(const BatchArgs args, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  __shared__ u32x4 tile[4 * 2 * 192];
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = blockIdx.x * 4 + wv;
  if (wt >= n_tasks) return;
  const FrameDesc f = args.f[blockIdx.y];
  const uint32_t rp = wt / chunks_x, chunk = wt - rp * chunks_x, x = chunk * 1024 + lane * 16;
  const bool act = x < w;
  u32x4 y[2] = {}, uv = {};
  if (act) {
    y[0] = ldg<true, u32x4>(f.s[0] + (size_t)(2 * rp) * f.sp[0] + x);
    y[1] = ldg<true, u32x4>(f.s[0] + (size_t)(2 * rp + 1) * f.sp[0] + x);
    uv = ldg<true, u32x4>(f.s[1] + (size_t)rp * f.sp[1] + x);
  }
#pragma unroll
  for (int half = 0; half < 2; half++) {
    u32x4* t = tile + (wv * 2 + half) * 192;
    if (act) {
#pragma unroll
      for (int j = 0; j < 3; j++) t[lane * 3 + j] = u32x4{y[half][0] ^ uv[j], y[half][1] ^ uv[j], y[half][2] ^ uv[j], y[half][3] ^ uv[j]};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint8_t* rowp = f.d[0] + (size_t)(2 * rp + half) * f.dp[0];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const uint32_t off = chunk * 3072 + (k * 64 + lane) * 16;
      if (off < 3 * w) stg<true, u32x4>(rowp + off, t[k * 64 + lane]);
    }
  }
}

static bool lab_aligned(const BatchArgs& a, uint32_t n, int ndst, uint32_t al) {
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < 2; k++) if (((uintptr_t)a.f[i].s[k] | a.f[i].sp[k]) & (al - 1)) return false;
    for (int k = 0; k < ndst; k++) if (((uintptr_t)a.f[i].d[k] | a.f[i].dp[k]) & (al - 1)) return false;
  }
  return true;
}

// -> hipSuccess, or hipErrorInvalidValue when the variant does not exist / does not apply to this frame shape (the lab never falls back)
template <int DST>
static hipError_t lab_launch(hipStream_t st, int variant, const Yuv2RgbCoef& c, uint32_t w, uint32_t h, uint32_t n, const BatchArgs& a) {
  constexpr int ndst = (DST == FC_PLANAR) ? 3 : 1;
  const bool ok16 = (w % 16 == 0) && (h % 2 == 0) && lab_aligned(a, n, ndst, 16);
  const bool ok4 = lab_aligned(a, n, ndst, 4);
  const bool packed = DST != FC_PLANAR;
  (void)hipGetLastError();
  const uint32_t chunks16 = (w + 1023) / 1024, tasks16 = chunks16 * (h / 2);
  const dim3 grid16((tasks16 + 3) / 4, n);
  auto p4 = [&](auto kern, int rp) {
    if (!ok4) return hipErrorInvalidValue;
    const uint32_t chunks = ((w + 3) / 4 + 63) / 64, tasks = chunks * (((h + 1) / 2 + rp - 1) / rp);
    hipLaunchKernelGGL(kern, dim3((tasks + 3) / 4, n), dim3(256), 0, st, a, c, w, h, chunks, tasks);
    return hipGetLastError();
  };
#define LAB_P16(NTL, NTS, LDS, WPB, BAL, SWZ, GRID, BLOCK) \
  do { if (!ok16) return hipErrorInvalidValue; hipLaunchKernelGGL((k_lab_p16<DST, NTL, NTS, LDS, WPB, BAL, SWZ>), GRID, BLOCK, 0, st, a, c, w, h, chunks16, tasks16); return hipGetLastError(); } while (0)
  switch (variant) {
    case 1: return p4(k_lab_p4<DST, 1, 1, false, false, 0>, 1);
    case 2: return p4(k_lab_p4<DST, 2, 1, false, false, 0>, 2);
    case 3: return p4(k_lab_p4<DST, 4, 1, false, false, 0>, 4);
    case 5: return p4(k_lab_p4<DST, 2, 1, true, true, 0>, 2);
    case 6: return p4(k_lab_p4<DST, 4, 1, true, true, 0>, 4);
    case 10: return p4(k_lab_p4<DST, 1, 0, true, true, 0>, 1);
    case 14: return p4(k_lab_p4<DST, 1, 1, false, true, 0>, 1);
    case 16: return p4(k_lab_p4<DST, 2, 1, false, true, 0>, 2);
    case 33: return p4(k_lab_p4<DST, 1, 1, true, true, 26>, 1);
    case 34: return p4(k_lab_p4<DST, 1, 1, true, true, 32>, 1);
    case 35: return p4(k_lab_p4<DST, 1, 1, true, true, 40>, 1);
    case 7: LAB_P16(false, false, true, 4, 0, false, grid16, dim3(256));
    case 11: LAB_P16(false, true, true, 4, 0, false, grid16, dim3(256));
    case 13: LAB_P16(true, true, false, 4, 0, false, grid16, dim3(256));
    case 36: LAB_P16(true, true, true, 4, 8, false, grid16, dim3(256));
    case 31: LAB_P16(true, true, true, 4, 29, false, grid16, dim3(256));
    case 32: LAB_P16(true, true, true, 4, 56, false, grid16, dim3(256));
    case 41: if (grid16.x & 7) return hipErrorInvalidValue; LAB_P16(true, true, true, 4, 16, true, grid16, dim3(256));
    case 42: if (grid16.x & 7) return hipErrorInvalidValue; LAB_P16(true, true, true, 4, 0, true, grid16, dim3(256));
    case 20: LAB_P16(false, true, true, 1, 0, false, dim3(tasks16, n), dim3(64));
    case 21: LAB_P16(false, true, true, 2, 0, false, dim3((tasks16 + 1) / 2, n), dim3(128));
    default: break;
  }
#undef LAB_P16
  if (!ok16) return hipErrorInvalidValue;
  if (variant >= 22 && variant <= 26) {
    if (variant == 22) hipLaunchKernelGGL((k_probe_p16<0>), grid16, dim3(256), 0, st, a, w, h, chunks16, tasks16);
    else if (variant == 23) hipLaunchKernelGGL((k_probe_p16<1>), grid16, dim3(256), 0, st, a, w, h, chunks16, tasks16);
    else if (variant == 24) hipLaunchKernelGGL((k_probe_p16<2>), grid16, dim3(256), 0, st, a, w, h, chunks16, tasks16);
    else if (variant == 25) hipLaunchKernelGGL((k_probe_p16<3>), grid16, dim3(256), 0, st, a, w, h, chunks16, tasks16);
    else hipLaunchKernelGGL((k_probe_p16<4>), grid16, dim3(256), 0, st, a, w, h, chunks16, tasks16);
    return hipGetLastError();
  }
  if (variant == 15) { hipLaunchKernelGGL(k_probe_nomath, grid16, dim3(256), 0, st, a, w, h, chunks16, tasks16); return hipGetLastError(); }
  if constexpr (DST != FC_PLANAR) {
    if (variant >= 17 && variant <= 19) {
      const uint32_t rpw = variant == 17 ? 1 : (variant == 18 ? 2 : 4), tasks = chunks16 * ((h / 2 + rpw - 1) / rpw);
      const dim3 grid((tasks + 3) / 4, n);
      if (rpw == 1) hipLaunchKernelGGL((k_nv12_rgb_p16r<DST, 1, true>), grid, dim3(256), 0, st, a, c, w, h, chunks16, tasks);
      else if (rpw == 2) hipLaunchKernelGGL((k_nv12_rgb_p16r<DST, 2, true>), grid, dim3(256), 0, st, a, c, w, h, chunks16, tasks);
      else hipLaunchKernelGGL((k_nv12_rgb_p16r<DST, 4, true>), grid, dim3(256), 0, st, a, c, w, h, chunks16, tasks);
      return hipGetLastError();
    }
    if (variant >= 27 && variant <= 29) {
      const uint32_t tiles = (w + 511) / 512, nt = tiles * (h / 2);
      if (variant == 27) hipLaunchKernelGGL((k_nv12_rgb_r4<DST, false, true>), dim3(nt, n), dim3(256), 0, st, a, c, w, h, tiles, nt);
      else if (variant == 28) hipLaunchKernelGGL((k_nv12_rgb_r4<DST, true, true>), dim3(nt, n), dim3(256), 0, st, a, c, w, h, tiles, nt);
      else hipLaunchKernelGGL((k_nv12_rgb_r4<DST, true, false>), dim3(nt, n), dim3(256), 0, st, a, c, w, h, tiles, nt);
      return hipGetLastError();
    }
    if (variant == 38) {
      const uint32_t tasks = chunks16 * h;
      hipLaunchKernelGGL((k_nv12_rgb_r16<DST, true, 0>), dim3((tasks + 3) / 4, n), dim3(256), 0, st, a, c, w, h, chunks16, tasks);
      return hipGetLastError();
    }
    if (variant == 45 || variant == 46) {  // row-crossing p16 (46: 16 KiB of LDS ballast like the product's batch kernel)
      if constexpr (DST == FC_PLANAR) { return hipErrorInvalidValue; } else {
        const uint32_t bpr = w / 16, nb = bpr * (h / 2);
        if (variant == 45) hipLaunchKernelGGL((k_nv12_rgb_p16x<DST, 0>), dim3((nb + 255) / 256, n), dim3(256), 0, st, a, c, w, h, bpr, nb);
        else hipLaunchKernelGGL((k_nv12_rgb_p16x<DST, 16>), dim3((nb + 255) / 256, n), dim3(256), 0, st, a, c, w, h, bpr, nb);
        return hipGetLastError();
      }
    }
    if (variant == 43) {
      const uint32_t segs = (3 * w + 1023) / 1024, tasks = segs * h;
      hipLaunchKernelGGL((k_nv12_rgb_s16<DST, true>), dim3((tasks + 3) / 4, n), dim3(256), 0, st, a, c, w, h, segs, tasks);
      return hipGetLastError();
    }
  }
  (void)packed;
  return hipErrorInvalidValue;
}

}  // namespace vpf

using namespace vpf;

extern "C" {
// 1 if `variant` exists and writes a correct conversion; 0 if it is a bandwidth probe (garbage output); -1 if unknown
__attribute__((visibility("default"))) int vpf_lab_is_conversion(int variant) {
  switch (variant) {
    case 15: case 22: case 23: case 24: case 25: case 26: return 0;
    case 1: case 2: case 3: case 5: case 6: case 7: case 10: case 11: case 13: case 14: case 16: case 17: case 18: case 19: case 20: case 21:
    case 27: case 28: case 29: case 31: case 32: case 33: case 34: case 35: case 36: case 38: case 41: case 42: case 43: case 45: case 46: return 1;
    default: return -1;
  }
}
// NV12 -> RGB / BGR / RGB_PLANAR over n <= 32 frames with lab kernel `variant`.  0 ok, 1 unknown variant / not applicable to this frame
// shape (the lab never falls back to another kernel), 2 bad argument, 3 launch error.
__attribute__((visibility("default"))) int vpf_lab_nv12_rgb(const vpf_exec* exec, int variant, int dst_fmt, int cs, int cr, vpf_size size, uint32_t n,
                                                           const vpf_frame_io* frames) {
  Yuv2RgbCoef c;
  if (!exec || !frames || !n || n > (uint32_t)kSmallBatch || !size.width || !size.height || !coef_yuv2rgb(cs, cr, &c)) return 2;
  BatchArgs a;
  std::memset(&a, 0, sizeof(a));
  const int nd = dst_fmt == VPF_FMT_RGB_PLANAR ? 3 : 1;
  for (uint32_t i = 0; i < (uint32_t)kSmallBatch; i++) {
    const vpf_frame_io& f = frames[i < n ? i : 0];
    for (int k = 0; k < 2; k++) { if (!f.src[k].ptr) return 2; a.f[i].s[k] = (const uint8_t*)f.src[k].ptr; a.f[i].sp[k] = f.src[k].pitch; }
    for (int k = 0; k < nd; k++) { if (!f.dst[k].ptr) return 2; a.f[i].d[k] = (uint8_t*)f.dst[k].ptr; a.f[i].dp[k] = f.dst[k].pitch; }
  }
  hipStream_t st = (hipStream_t)exec->stream;
  hipError_t e;
  switch (dst_fmt) {
    case VPF_FMT_RGB: e = lab_launch<FC_RGB>(st, variant, c, size.width, size.height, n, a); break;
    case VPF_FMT_BGR: e = lab_launch<FC_BGR>(st, variant, c, size.width, size.height, n, a); break;
    case VPF_FMT_RGB_PLANAR: e = lab_launch<FC_PLANAR>(st, variant, c, size.width, size.height, n, a); break;
    default: return 2;
  }
  return e == hipSuccess ? 0 : (e == hipErrorInvalidValue ? 1 : 3);
}
}
