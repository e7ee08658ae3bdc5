// k_yuv2rgb_tasks.h — device-side task bodies of the NV12 / YUV420 -> RGB kernels (gfx950), shared by the product
// kernels (k_yuv2rgb.hip) and by the measurement lab (tools/lab/k_lab.hip, which instantiates them with other memory-policy /
// occupancy parameters).  Every template parameter here is correctness-neutral: all instantiations write identical pixels.
#pragma once
#include "vpf_device.h"

namespace vpf {

// ---------------------------------------------------------------------------------------------
// pixel math
// ---------------------------------------------------------------------------------------------
// 4 px -> 12 packed bytes (3 dwords) in R,G,B or B,G,R order
template <int DST, int PACK>
VPF_DEV void pack_rgb12(const Quad& q, uint32_t& d0, uint32_t& d1, uint32_t& d2) {
  const float* a = (DST == FC_BGR) ? q.b : q.r;
  const float* c = (DST == FC_BGR) ? q.r : q.b;
  d0 = pack4<PACK>(a[0], q.g[0], c[0], a[1]);
  d1 = pack4<PACK>(q.g[1], c[1], a[2], q.g[2]);
  d2 = pack4<PACK>(c[2], a[3], q.g[3], c[3]);
}

// ---------------------------------------------------------------------------------------------
// p4: 4 px per lane, RP row pairs per wave task.  Requires every plane pointer/pitch 4-byte aligned (2-byte for
// YUV420 chroma) — which also guarantees a whole dword can be LOADED at the ragged end of a row (pitch >= round_up(w,4)).
// Any width / height: the last pixel group of a row stores only its valid bytes, an odd last row is a pair of one.
// SRC in {FC_NV12, FC_YUV420}.
// ---------------------------------------------------------------------------------------------
VPF_DEV void store_bytes(uint8_t* p, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t nbytes) {
  const uint32_t d[3] = {d0, d1, d2};
  for (uint32_t i = 0; i < nbytes; i++) p[i] = (uint8_t)(d[i >> 2] >> (8 * (i & 3)));
}

template <int SRC, int DST, int RP, int PACK, bool NTL, bool NTS, int BALLAST_KB = 0>
VPF_DEV void yuv420_rgb_p4_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  if constexpr (BALLAST_KB > 0) {  // occupancy experiment: an LDS footprint that caps resident workgroups per CU
    __shared__ uint32_t ballast[BALLAST_KB * 256];
    if (n_tasks == 0xffffffffu) ballast[threadIdx.x] = w;  // never true; keeps the allocation
  }
  // wave-uniform by construction; readfirstlane tells the compiler so (scalar branches, SGPR addressing)
  const uint32_t wt = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wt >= n_tasks) return;
  const uint32_t rpt = wt / chunks_x, chunk = wt - rpt * chunks_x;
  const uint32_t x = (chunk * 64 + (threadIdx.x & 63)) * 4;
  if (x >= w) return;
  const uint32_t nrp = (h + 1) >> 1, rp0 = rpt * RP;
  const uint32_t npx = (w - x < 4) ? w - x : 4;  // valid pixels of this lane's group (4 except at a ragged row end)

  uint32_t ya[RP], yb[RP], uv[RP];
#pragma unroll
  for (int r = 0; r < RP; r++) {
    const uint32_t rp = rp0 + r;
    if (rp < nrp) {
      ya[r] = ldg<NTL, uint32_t>(f.s[0] + (size_t)(2 * rp) * f.sp[0] + x);
      yb[r] = (2 * rp + 1 < h) ? ldg<NTL, uint32_t>(f.s[0] + (size_t)(2 * rp + 1) * f.sp[0] + x) : 0u;
      if constexpr (SRC == FC_NV12) {
        uv[r] = ldg<NTL, uint32_t>(f.s[1] + (size_t)rp * f.sp[1] + x);
      } else {  // two U bytes and two V bytes -> same (U0 V0 U1 V1) byte order as NV12
        uint32_t u2 = ldg<NTL, uint16_t>(f.s[1] + (size_t)rp * f.sp[1] + (x >> 1));
        uint32_t v2 = ldg<NTL, uint16_t>(f.s[2] + (size_t)rp * f.sp[2] + (x >> 1));
        uv[r] = __builtin_amdgcn_perm(v2, u2, 0x05010400u);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RP; r++) {
    const uint32_t rp = rp0 + r;
    if (rp < nrp) {
      const Chroma k0 = chroma_terms(c, ubyte<0>(uv[r]), ubyte<1>(uv[r]));
      const Chroma k1 = chroma_terms(c, ubyte<2>(uv[r]), ubyte<3>(uv[r]));
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const size_t row = (size_t)(2 * rp + half);
        if (row >= h) break;
        const Quad q = convert4(c, half ? yb[r] : ya[r], k0, k1);
        if constexpr (DST == FC_PLANAR) {
          const uint32_t pr = pack4<PACK>(q.r[0], q.r[1], q.r[2], q.r[3]), pg = pack4<PACK>(q.g[0], q.g[1], q.g[2], q.g[3]),
                         pb = pack4<PACK>(q.b[0], q.b[1], q.b[2], q.b[3]);
          if (npx == 4) {
            stg<NTS, uint32_t>(f.d[0] + row * f.dp[0] + x, pr);
            stg<NTS, uint32_t>(f.d[1] + row * f.dp[1] + x, pg);
            stg<NTS, uint32_t>(f.d[2] + row * f.dp[2] + x, pb);
          } else {
            store_bytes(f.d[0] + row * f.dp[0] + x, pr, 0, 0, npx);
            store_bytes(f.d[1] + row * f.dp[1] + x, pg, 0, 0, npx);
            store_bytes(f.d[2] + row * f.dp[2] + x, pb, 0, 0, npx);
          }
        } else {
          uint32_t d0, d1, d2;
          pack_rgb12<DST, PACK>(q, d0, d1, d2);
          if (npx == 4) stg3<NTS>(f.d[0] + row * f.dp[0] + 3 * (size_t)x, d0, d1, d2);
          else store_bytes(f.d[0] + row * f.dp[0] + 3 * (size_t)x, d0, d1, d2, 3 * npx);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// p16: 16 px per lane (dwordx4 loads), one row pair x 1024 px per wave task.  Packed outputs are
// transposed through a wave-private LDS tile (LDS_T) so each global store instruction writes a
// dense 1 KiB; LDS_T = false keeps the lane-strided 48 B stores for comparison.
// Requires w % 16 == 0, h even, 16-byte aligned planes and pitches.  SRC = FC_NV12 only.
//
// LDS banking: ds_write_b128 is serviced in groups of 8 consecutive lanes; lane l writes at byte
// 48*l + 16*j -> dword banks {12l+4j .. +3} mod 32, which tile all 32 banks exactly once per
// group: conflict free.  The ds_read_b128 side reads 16*l: contiguous, conflict free.
// ---------------------------------------------------------------------------------------------
// XCD_SWZ: workgroups are dealt round-robin to the 8 XCDs (workgroup b -> XCD b % 8 when gridDim.x % 8 == 0); the
// swizzle hands each XCD ONE contiguous eighth of every frame instead of every eighth row pair, so each XCD's L2
// write-back stream is sequential (tools/write_probe.hip X0/X1: +6 % on pure writes).
template <int DST, int PACK, bool NTL, bool NTS, bool LDS_T, int WPB = 4, int BALLAST_KB = 0, bool XCD_SWZ = false, int SRC = FC_NV12>
VPF_DEV void p16_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  // BALLAST_KB > 0 pads the LDS footprint to cap the number of resident workgroups per CU (occupancy experiment)
  __shared__ u32x4 tile[((LDS_T && DST != FC_PLANAR) ? WPB * 2 * 192 : 1) + BALLAST_KB * 64];
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint32_t bx = blockIdx.x;
  if constexpr (XCD_SWZ) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);  // host guarantees gridDim.x % 8 == 0
  const uint32_t wt = bx * WPB + wv;
  if (wt >= n_tasks) return;
  const uint32_t rp = wt / chunks_x, chunk = wt - rp * chunks_x;
  const uint32_t x = chunk * 1024 + lane * 16;
  const bool act = x < w;

  u32x4 y[2], uv;
  if (act) {
    y[0] = ldg<NTL, u32x4>(f.s[0] + (size_t)(2 * rp) * f.sp[0] + x);
    y[1] = ldg<NTL, u32x4>(f.s[0] + (size_t)(2 * rp + 1) * f.sp[0] + x);
    uv = load_uv16<SRC, NTL>(f, rp, x);
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
        if constexpr (DST == FC_PLANAR) {
          o[half][j] = pack4<PACK>(q.r[0], q.r[1], q.r[2], q.r[3]);
          o[half][4 + j] = pack4<PACK>(q.g[0], q.g[1], q.g[2], q.g[3]);
          o[half][8 + j] = pack4<PACK>(q.b[0], q.b[1], q.b[2], q.b[3]);
        } else {
          pack_rgb12<DST, PACK>(q, o[half][3 * j], o[half][3 * j + 1], o[half][3 * j + 2]);
        }
      }
    }
  }
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const size_t row = (size_t)(2 * rp + half);
    if constexpr (DST == FC_PLANAR) {
      if (act) {
#pragma unroll
        for (int p = 0; p < 3; p++)
          stg<NTS, u32x4>(f.d[p] + row * f.dp[p] + x,
                         u32x4{o[half][4 * p], o[half][4 * p + 1], o[half][4 * p + 2], o[half][4 * p + 3]});
      }
    } else if constexpr (LDS_T) {
      u32x4* t = tile + (wv * 2 + half) * 192;
      if (act) {
#pragma unroll
        for (int j = 0; j < 3; j++)
          t[lane * 3 + j] = u32x4{o[half][4 * j], o[half][4 * j + 1], o[half][4 * j + 2], o[half][4 * j + 3]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint8_t* rowp = f.d[0] + row * f.dp[0];
      const uint32_t row_bytes = 3 * w;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const uint32_t off = chunk * 3072 + (k * 64 + lane) * 16;
        if (off < row_bytes) stg<NTS, u32x4>(rowp + off, t[k * 64 + lane]);
      }
    } else {
      if (act) {
        uint8_t* p = f.d[0] + row * f.dp[0] + 3 * (size_t)x;
#pragma unroll
        for (int j = 0; j < 3; j++)
          stg<NTS, u32x4>(p + 16 * j, u32x4{o[half][4 * j], o[half][4 * j + 1], o[half][4 * j + 2], o[half][4 * j + 3]});
      }
    }
  }
}

// p16 with the frame's 16-px x 2-row blocks numbered STRAIGHT THROUGH the picture ("p16x"): a wave takes 64 consecutive blocks wherever
// the row ends.  With p16's chunk-per-row tasks a 3840-px row pair is 3.75 waves — every fourth wave works with 48 of its 64 lanes, 6 % fewer
// bytes in flight chip-wide — and the straight numbering is worth exactly that on the HBM-bound batch: 0.78 -> 0.82 of 8 TB/s
// (profiles/r02_bench_sweep.log, lab 45 / 46 before it moved here).  Loads are per lane (own row pair, own column); the LDS-transposed
// stores compute the destination of every 16-B unit from the lane that produced it.  A wave crosses at most one row boundary:
// requires bpr = w / 16 >= 64 (w >= 1024), w % 16 == 0, h even, 16-B aligned planes / pitches.  Packed outputs only.
template <int DST, bool NTS, int BALLAST_KB, int SRC>
VPF_DEV void p16x_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t bpr /* blocks per row pair */, uint32_t n_blocks) {
  __shared__ u32x4 tile[4 * 2 * 192 + BALLAST_KB * 64];
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t b0 = (blockIdx.x * 4 + wv) * 64;
  if (b0 >= n_blocks) return;
  const uint32_t rp0 = __builtin_amdgcn_readfirstlane(b0 / bpr), c0 = b0 - rp0 * bpr;  // the wave's first block: row pair, column (in blocks)
  auto place = [&](uint32_t l, uint32_t& rp, uint32_t& x) {  // block b0 + l: at most one row boundary inside a wave (bpr >= 64)
    const uint32_t col = c0 + l, over = col >= bpr ? 1u : 0u;
    rp = rp0 + over; x = (col - over * bpr) * 16;
  };
  const bool act = b0 + lane < n_blocks;
  uint32_t rp, x;
  place(lane, rp, x);
  u32x4 y[2], uv;
  if (act) {
    y[0] = ldg<true, u32x4>(f.s[0] + (size_t)(2 * rp) * f.sp[0] + x);
    y[1] = ldg<true, u32x4>(f.s[0] + (size_t)(2 * rp + 1) * f.sp[0] + x);
    uv = load_uv16<SRC, true>(f, rp, x);
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
    wave_sync();
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const uint32_t idx = k * 64 + lane, src_lane = idx / 3, part = idx - 3 * src_lane;  // 16-B unit idx was produced by lane idx / 3
      if (b0 + src_lane < n_blocks) {
        uint32_t urp, ux;
        place(src_lane, urp, ux);
        stg<NTS, u32x4>(f.d[0] + (size_t)(2 * urp + half) * f.dp[0] + 3 * (size_t)ux + 16 * part, t[idx]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// r16 (planar outputs): a wave owns ONE row x 1024 px: lane = 16 px, Y dwordx4 + UV dwordx4 (the row below re-reads the same
// UV line from L2, not HBM), three dense 1-KiB dwordx4 stores (R, G, B planes) — 3 stores per wave instead of the 6 a
// row-pair wave needs for three planes (write-rate law, tools/write_probe.hip).  Requires w % 16 == 0, 16-B aligned planes.
// ---------------------------------------------------------------------------------------------
template <bool NTS, int SRC>
VPF_DEV void planar_r16_task(const FrameDesc& f, const Yuv2RgbCoef& c, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  const uint32_t wt = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wt >= n_tasks) return;
  // consecutive waves of a block take rows 2rp, 2rp+1 of the same chunk, so the shared UV line is hot in L1/L2
  const uint32_t pair = wt >> 1, half = wt & 1;
  const uint32_t rp = pair / chunks_x, chunk = pair - rp * chunks_x;
  const uint32_t x = chunk * 1024 + (threadIdx.x & 63) * 16;
  const uint32_t y = 2 * rp + half;
  if (x >= w || y >= h) return;
  const u32x4 yq = ldg<true, u32x4>(f.s[0] + (size_t)y * f.sp[0] + x);
  const u32x4 uv = load_uv16<SRC, false>(f, rp, x);
  u32x4 r, g, b;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const Chroma k0 = chroma_terms(c, ubyte<0>(uv[j]), ubyte<1>(uv[j])), k1 = chroma_terms(c, ubyte<2>(uv[j]), ubyte<3>(uv[j]));
    const Quad q = convert4(c, yq[j], k0, k1);
    r[j] = pack4<1>(q.r[0], q.r[1], q.r[2], q.r[3]);
    g[j] = pack4<1>(q.g[0], q.g[1], q.g[2], q.g[3]);
    b[j] = pack4<1>(q.b[0], q.b[1], q.b[2], q.b[3]);
  }
  stg<NTS, u32x4>(f.d[0] + (size_t)y * f.dp[0] + x, r);
  stg<NTS, u32x4>(f.d[1] + (size_t)y * f.dp[1] + x, g);
  stg<NTS, u32x4>(f.d[2] + (size_t)y * f.dp[2] + x, b);
}

}  // namespace vpf
