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
template <int SRC, int DST, int RP, int PACK, bool NTL, bool NTS, int BALLAST_KB = 0>
__global__ __launch_bounds__(256) void k_yuv420_rgb_p4(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w,
                                                       uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  yuv420_rgb_p4_task<SRC, DST, RP, PACK, NTL, NTS, BALLAST_KB>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
template <int SRC, int DST>  // single-frame entry of the default p4 form (irregular sizes / alignments): scalar arguments
__global__ __launch_bounds__(256) void k_yuv420_rgb_p4_one(VPF_ONE_SRC_PARAMS, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks,
                                                           VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  yuv420_rgb_p4_task<SRC, DST, 1, 1, true, true, 0>(VPF_ONE_FRAME, c, w, h, chunks_x, n_tasks);
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
template <int DST, int PACK, bool NTL, bool NTS, bool LDS_T, bool NOMATH, int WPB = 4, int BALLAST_KB = 0, bool XCD_SWZ = false, int SRC = FC_NV12>
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
  if constexpr (NOMATH) {  // bandwidth-ceiling probe: same loads / LDS transpose / stores, no arithmetic (NOT a conversion)
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
      for (int j = 0; j < 12; j++) o[half][j] = y[half][j & 3] ^ uv[(j >> 2) & 3];
  } else if (act) {
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

// batched entry: up to 32 frame descriptors by value in the kernarg segment, blockIdx.y = frame
template <int DST, int PACK, bool NTL, bool NTS, bool LDS_T, bool NOMATH, int WPB = 4, int BALLAST_KB = 0, bool XCD_SWZ = false, int SRC = FC_NV12>
__global__ __launch_bounds__(64 * WPB) void k_nv12_rgb_p16(const BatchArgs args, const Yuv2RgbCoef c, uint32_t w,
                                                           uint32_t h, uint32_t chunks_x, uint32_t n_tasks) {
  p16_task<DST, PACK, NTL, NTS, LDS_T, NOMATH, WPB, BALLAST_KB, XCD_SWZ, SRC>(args.f[blockIdx.y], c, w, h, chunks_x, n_tasks);
}
// single-frame entry (one Execute() = one launch): the frame arrives as scalar kernel arguments, source side first, which
// the dispatcher preloads into SGPRs (-amdgpu-kernarg-preload-count): the wave's first loads no longer wait for a
// scalar-cache round trip to the kernarg segment — worth 0.5-0.8 us on a kernel that lasts 6-7 us
template <int DST, bool NTS, int SRC>
__global__ __launch_bounds__(256) void k_nv12_rgb_p16_one(VPF_ONE_SRC_PARAMS, uint32_t w, uint32_t h, uint32_t chunks_x, uint32_t n_tasks,
                                                          VPF_ONE_DST_PARAMS, const Yuv2RgbCoef c) {
  p16_task<DST, 1, true, NTS, true, false, 4, 0, false, SRC>(VPF_ONE_FRAME, c, w, h, chunks_x, n_tasks);
}

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

template <int SRC, int DST>
static hipError_t launch_420(hipStream_t st, const Yuv2RgbCoef& c, uint32_t w, uint32_t h, uint32_t n,
                             const BatchArgs& a, int variant) {
  const int nsrc = (SRC == FC_NV12) ? 2 : 3, ndst = (DST == FC_PLANAR) ? 3 : 1;
  const bool even = (w % 4 == 0) && (h % 2 == 0);
  // variant (VPF_TUNE_NV12_RGB_VARIANT): 0 = default policy
  //   1/2/3   p4, 1/2/4 row pairs per wave task          4/5/6   the same with non-temporal loads+stores
  //   7       p16 + LDS transpose                        8       p16 + LDS transpose, non-temporal loads+stores
  //   9       generic byte kernel                        10      p4 RP1 NT with the explicit (non cvt_pk) pack
  //   11/12   p16 LDS with NT stores only / NT loads only
  //   13      p16 lane-strided stores (no LDS)           15      p16 LDS NT, arithmetic removed (ceiling probe, wrong pixels)
  //   22-26   bandwidth probes (wrong pixels): loads only / stores only (NT, plain, linear NT, linear plain)
  //   37      r16 (planar outputs): one row x 1024 px per wave, 3 stores      38  r16 packed (LDS transpose; ties with 30)
  //   27      r4: one 768-B store per wave (lane = 4 px of one row)    28/29  b4: r4 + block LDS gather -> 1-KiB stores (NT / plain)
  //   14/16   p4 RP1 / RP2 with NT stores only          17/18/19 p16r (one LDS row tile per wave), 1/2/4 row pairs per task, NT stores
  if constexpr (SRC == FC_YUV420) {
    // I420 (what software decoders hand over): the same 16-px kernels with the chroma re-interleaved in registers
    // (load_uv16).  Default policy as for NV12; tuning values other than these take the p4 / generic kernels below.
    const bool ok420 = even && (w % 16 == 0) && aligned_all(a, n, 3, ndst, 16, 16, 8);
    const bool mine = variant == 0 || variant == 8 || variant == 12 || variant == 30 || variant == 37 || variant == 44;
    if (ok420 && mine) {
      const int v = variant ? variant : (DST == FC_PLANAR ? 37 : (n >= 4 ? 30 : 8));
      if constexpr (DST == FC_PLANAR) {
        if (v == 37 || v == 44) {
          const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * h;
          dim3 grid((tasks + 3) / 4, n);
          if (n == 1 && v == 37) VPF_LAUNCH((k_nv12_planar_r16_one<true, FC_YUV420>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), w, h, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c);
          else if (n == 1) VPF_LAUNCH((k_nv12_planar_r16_one<false, FC_YUV420>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), w, h, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c);
          else if (v == 37) VPF_LAUNCH((k_nv12_planar_r16<true, FC_YUV420>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
          else VPF_LAUNCH((k_nv12_planar_r16<false, FC_YUV420>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
          return hipGetLastError();
        }
      } else {
        if (v == 8 || v == 12 || v == 30) {
          const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * (h / 2);
          dim3 grid((tasks + 3) / 4, n);
          const FrameDesc& f0 = a.f[0];
          if (n == 1 && v == 8) VPF_LAUNCH((k_nv12_rgb_p16_one<DST, true, FC_YUV420>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
          else if (n == 1 && v == 12) VPF_LAUNCH((k_nv12_rgb_p16_one<DST, false, FC_YUV420>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
          else if (v == 8) VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, true, true, false, 4, 0, false, FC_YUV420>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
          else if (v == 12) VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, false, true, false, 4, 0, false, FC_YUV420>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
          else VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, true, true, false, 4, 16, false, FC_YUV420>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
          return hipGetLastError();
        }
      }
    }
  }
  const bool p16_ok = (SRC == FC_NV12) && even && (w % 16 == 0) && aligned_all(a, n, nsrc, ndst, 16, 16, 16);
  const bool p4_ok = aligned_all(a, n, nsrc, ndst, 4, 4, SRC == FC_NV12 ? 4 : 2);
  // default policy (profiles/r01_bench_sweep.log): packed outputs -> p16 + LDS transpose with non-temporal loads and
  // stores (ties with the other top variants when batched, +3 % when a launch is a single frame); planar outputs
  // (no transpose needed) and anything not 16-B aligned -> p4 non-temporal
  // batched launches run long enough that 4 resident workgroups per CU (variant 30: LDS-capped) beat 6 by 1-2 %
  // (a narrower chip-wide write frontier; tools/write_probe.hip); short single-frame launches want all the waves they can get
  // planar outputs: r16 (one row per wave, three 1-KiB plane stores) beats p4's 256-B stores by ~9 % when batched
  // a lone big planar frame: the row-pair kernel (half as many waves, each bringing in 2.5 KiB) beats r16 — kernel durations
  // 7.3 vs 7.9 us at 4K, but 4.3 vs 3.6 at 1080p and 3.7 vs 2.8 at 720p (tools/gpu_planar_single.sh)
  const bool big_single = n < 4 && (size_t)w * h >= (size_t)3 << 20;
  if (variant == 0) variant = p16_ok ? (DST == FC_PLANAR ? (big_single ? 8 : 37) : (n >= 4 ? 30 : 8)) : 4;
  const bool want_p16 = (variant == 7 || variant == 8 || (variant >= 11 && variant <= 15) || (variant >= 17 && variant <= 21) || (variant >= 30 && variant <= 32) || variant == 36 || variant == 41 || variant == 42);
  const bool packed_only = (variant >= 17 && variant <= 19) || (variant >= 22 && variant <= 29) || variant == 38 || variant == 43;
  if ((want_p16 || packed_only || variant == 37 || variant == 44) && !p16_ok) variant = 4;
  if ((packed_only && DST == FC_PLANAR) || ((variant == 37 || variant == 44) && DST != FC_PLANAR)) variant = 4;
  if (variant != 9 && !p4_ok) variant = 9;  // p16_ok implies p4_ok
  if constexpr (SRC == FC_NV12) {
    if (variant >= 17 && variant <= 19) {  // p16r: RPW = 1, 2, 4
      const uint32_t rpw = variant == 17 ? 1 : (variant == 18 ? 2 : 4);
      const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * ((h / 2 + rpw - 1) / rpw);
      dim3 grid((tasks + 3) / 4, n);
      if constexpr (DST != FC_PLANAR) {
        if (rpw == 1) VPF_LAUNCH((k_nv12_rgb_p16r<DST, 1, true>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
        else if (rpw == 2) VPF_LAUNCH((k_nv12_rgb_p16r<DST, 2, true>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
        else VPF_LAUNCH((k_nv12_rgb_p16r<DST, 4, true>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
      }
      return hipGetLastError();
    }
    if (variant == 38) {
      const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * h;
      dim3 grid((tasks + 3) / 4, n);
      if constexpr (DST != FC_PLANAR) {
        VPF_LAUNCH((k_nv12_rgb_r16<DST, true, 0>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
      }
      return hipGetLastError();
    }
    if (variant == 43) {
      const uint32_t segs = (3 * w + 1023) / 1024, tasks = segs * h;
      if constexpr (DST != FC_PLANAR) VPF_LAUNCH((k_nv12_rgb_s16<DST, true>), dim3((tasks + 3) / 4, n), dim3(256), 0, st, a, c, w, h, segs, tasks);
      return hipGetLastError();
    }
    if (variant == 37 || variant == 44) {
      const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * h;  // one task per row per chunk (h even)
      dim3 grid((tasks + 3) / 4, n);
      if (n == 1 && variant == 37) VPF_LAUNCH((k_nv12_planar_r16_one<true, FC_NV12>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), w, h, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c);
      else if (n == 1) VPF_LAUNCH((k_nv12_planar_r16_one<false, FC_NV12>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), w, h, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c);
      else if (variant == 37) VPF_LAUNCH((k_nv12_planar_r16<true>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
      else VPF_LAUNCH((k_nv12_planar_r16<false>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);  // allocating stores
      return hipGetLastError();
    }
    if (variant >= 27 && variant <= 29) {
      const uint32_t tiles = (w + 511) / 512, nt = tiles * (h / 2);
      dim3 grid(nt, n);
      if constexpr (DST != FC_PLANAR) {
        if (variant == 27) VPF_LAUNCH((k_nv12_rgb_r4<DST, false, true>), grid, dim3(256), 0, st, a, c, w, h, tiles, nt);
        else if (variant == 28) VPF_LAUNCH((k_nv12_rgb_r4<DST, true, true>), grid, dim3(256), 0, st, a, c, w, h, tiles, nt);
        else VPF_LAUNCH((k_nv12_rgb_r4<DST, true, false>), grid, dim3(256), 0, st, a, c, w, h, tiles, nt);
      }
      return hipGetLastError();
    }
    if (variant >= 22 && variant <= 26) {
      const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * (h / 2);
      dim3 grid((tasks + 3) / 4, n);
      if (variant == 22) VPF_LAUNCH((k_probe_p16<0>), grid, dim3(256), 0, st, a, w, h, chunks, tasks);
      else if (variant == 23) VPF_LAUNCH((k_probe_p16<1>), grid, dim3(256), 0, st, a, w, h, chunks, tasks);
      else if (variant == 24) VPF_LAUNCH((k_probe_p16<2>), grid, dim3(256), 0, st, a, w, h, chunks, tasks);
      else if (variant == 25) VPF_LAUNCH((k_probe_p16<3>), grid, dim3(256), 0, st, a, w, h, chunks, tasks);
      else VPF_LAUNCH((k_probe_p16<4>), grid, dim3(256), 0, st, a, w, h, chunks, tasks);
      return hipGetLastError();
    }
    if (want_p16 && p16_ok) {
      const uint32_t chunks = (w + 1023) / 1024, tasks = chunks * (h / 2);
      dim3 grid((tasks + 3) / 4, n);
      if ((variant == 41 || variant == 42) && (grid.x & 7)) variant = (variant == 41) ? 30 : 8;  // swizzle needs gridDim.x % 8 == 0
#define VPF_P16(NTL, NTS, LDS, NOMATH) \
  VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, NTL, NTS, LDS, NOMATH>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks)
      if (n == 1 && (variant == 8 || variant == 12)) {  // one frame per launch: scalar-argument entry
        const FrameDesc& f0 = a.f[0];
        if (variant == 8) VPF_LAUNCH((k_nv12_rgb_p16_one<DST, true, FC_NV12>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
        else VPF_LAUNCH((k_nv12_rgb_p16_one<DST, false, FC_NV12>), grid, dim3(256), 0, st, VPF_ONE_SRC_ARGS(f0), w, h, chunks, tasks, VPF_ONE_DST_ARGS(f0), c);
        return hipGetLastError();
      }
      switch (variant) {
        case 7: VPF_P16(false, false, true, false); break;
        case 11: VPF_P16(false, true, true, false); break;
        case 12: VPF_P16(true, false, true, false); break;
        case 13: VPF_P16(true, true, false, false); break;
        case 15: VPF_P16(true, true, true, true); break;
        case 36: VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, true, true, false, 4, 8>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks); break;  // 32 KiB -> 5 blocks/CU
        case 30: VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, true, true, false, 4, 16>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks); break;  // 40 KiB -> 4 blocks/CU
        case 41: VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, true, true, false, 4, 16, true>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks); break;  // 30 + XCD swizzle
        case 42: VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, true, true, false, 4, 0, true>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks); break;   // 8 + XCD swizzle
        case 31: VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, true, true, false, 4, 29>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks); break;  // 53 KiB -> 3 blocks/CU
        case 32: VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, true, true, true, false, 4, 56>), grid, dim3(256), 0, st, a, c, w, h, chunks, tasks); break;  // 80 KiB -> 2 blocks/CU
        case 20: {  // one wave per workgroup: 4x more, smaller workgroups -> finer balance when a launch is only one frame
          dim3 g1(tasks, n);
          VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, false, true, true, false, 1>), g1, dim3(64), 0, st, a, c, w, h, chunks, tasks);
        } break;
        case 21: {  // two waves per workgroup
          dim3 g2((tasks + 1) / 2, n);
          VPF_LAUNCH((k_nv12_rgb_p16<DST, 1, false, true, true, false, 2>), g2, dim3(128), 0, st, a, c, w, h, chunks, tasks);
        } break;
        default: VPF_P16(true, true, true, false);
      }
#undef VPF_P16
      return hipGetLastError();
    }
  }
  if (variant != 9) {
    const uint32_t chunks = ((w + 3) / 4 + 63) / 64;
    auto go = [&](auto kern, int rp) {
      const uint32_t tasks = chunks * (((h + 1) / 2 + rp - 1) / rp);
      dim3 grid((tasks + 3) / 4, n);
      VPF_LAUNCH(kern, grid, dim3(256), 0, st, a, c, w, h, chunks, tasks);
      return hipGetLastError();
    };
    switch (variant) {
      case 1: return go(k_yuv420_rgb_p4<SRC, DST, 1, 1, false, false>, 1);
      case 2: return go(k_yuv420_rgb_p4<SRC, DST, 2, 1, false, false>, 2);
      case 3: return go(k_yuv420_rgb_p4<SRC, DST, 4, 1, false, false>, 4);
      case 5: return go(k_yuv420_rgb_p4<SRC, DST, 2, 1, true, true>, 2);
      case 6: return go(k_yuv420_rgb_p4<SRC, DST, 4, 1, true, true>, 4);
      case 10: return go(k_yuv420_rgb_p4<SRC, DST, 1, 0, true, true>, 1);
      case 33: return go(k_yuv420_rgb_p4<SRC, DST, 1, 1, true, true, 26>, 1);  // 6 blocks/CU
      case 34: return go(k_yuv420_rgb_p4<SRC, DST, 1, 1, true, true, 32>, 1);  // 5 blocks/CU
      case 35: return go(k_yuv420_rgb_p4<SRC, DST, 1, 1, true, true, 40>, 1);  // 4 blocks/CU
      case 14: return go(k_yuv420_rgb_p4<SRC, DST, 1, 1, false, true>, 1);
      case 16: return go(k_yuv420_rgb_p4<SRC, DST, 2, 1, false, true>, 2);
      default:
        if (n == 1) {
          const uint32_t tasks = chunks * ((h + 1) / 2);
          VPF_LAUNCH((k_yuv420_rgb_p4_one<SRC, DST>), dim3((tasks + 3) / 4, 1), dim3(256), 0, st, VPF_ONE_SRC_ARGS(a.f[0]), w, h, chunks, tasks, VPF_ONE_DST_ARGS(a.f[0]), c);
          return hipGetLastError();
        }
        return go(k_yuv420_rgb_p4<SRC, DST, 1, 1, true, true>, 1);
    }
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
