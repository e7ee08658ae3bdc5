// k_resize.hip — nearest / bilinear / Lanczos-3 resize of 8-bit and float surfaces (gfx950): vpf_resize, vpf_resize_batch.
// (Remap lives in k_remap.hip, the fused convert + resize kernels in k_convert_resize.hip; the three share k_bilinear_blend.h.)
//
// Replaces nppiResize_8u_C3R / _C1R (reference: NppResizeSurfacePacked3C_Impl::Run and NppResizeSurfacePlanar_Impl::Run,
// src/TC/src/Tasks.cpp:1162-1203,1217-1261) and nppiResize_32f_C3R / _C1R (:1334-1445).  The reference resizer asks NPP for
// Lanczos (:1190), which is the Task layer's default here too (north_star benchmarks bilinear: VPF_INTERP_LINEAR).
//
// Kernel families; within a filter all are bit-identical to one another (launch_resize / launch_resize_jobs pick):
//   GatherTask, LanczosGatherTask, FloatGatherTask   gather forms: any size / alignment
//   RowPairTask (k_resize_lds)    a wave stages the two source rows it needs in wave-private LDS strips (dynamic LDS sized per scale
//                                 factor) and picks taps from LDS; taps with weight exactly 0 are skipped
//   RowBandTask                   batches: R destination rows per wave, column taps once, each source row's horizontal lerp once per band
//   TileTask (k_resize_tile), TileTaskF32   tiled + separable: horizontal pass once per (source row, column) into LDS, then the vertical pass
//                                 (8-bit: bilinear up-scales of a single frame; float surfaces: Lanczos-3, bilinear with tuning 43)
//   8-bit Lanczos-3               k_lanczos_mfma.hip (the i8 matrix cores) wherever its tap windows fit (down-scales up to ~6 x, 1- / 2-channel
//                                 planes ~10 x across); LanczosTileTask below beyond them and for one small frame per dispatch;
//                                 LanczosGatherTask for what is left
//   HalfTask, Half3R16Task        exact 2x: quad-structured streaming kernels (no taps, no gathers; integer blend)
//   odd integer factors on both axes   every filter returns the centre sample -> nearest kernel (Lanczos) / RowPairTask's byte-move path
//   packed RGB taps               both taps of a row = 6 contiguous bytes: fetched as ONE 12-B window from the aligned address below
//                                 (global or LDS) and cut out with v_alignbyte_b32
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "k_resize_common.h"
#ifdef VPF_LAB_FORMS
#include "vpf_persist.h"
#endif

#include <atomic>

namespace vpf {

// CH interleaved channels per pixel (1, 2 or 3); 4 destination pixels per lane
template <int CH, int INTERP>
struct GatherTask {
  static constexpr int kThreads = 256;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by);
};
template <int CH, int INTERP>
VPF_DEV void GatherTask<CH, INTERP>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G,
                                         uint32_t bx, uint32_t by) {
  const uint32_t sw = G.sw, sh = G.sh, dw = G.dw, dh = G.dh;
  const float scx = G.scx, scy = G.scy;
  const int vec_ok = G.vec_ok;
  const uint32_t gx = bx * 64 + (threadIdx.x & 63);
  const uint32_t y = by * 4 + (threadIdx.x >> 6);
  const uint32_t x0 = gx * 4;
  if (x0 >= dw || y >= dh) return;
  const Tap ty = make_tap<INTERP>(y, scy, sh);
  const uint8_t* r0 = src + (size_t)ty.i0 * sp;
  const uint8_t* r1 = src + (size_t)ty.i1 * sp;
  float o[4 * CH];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t x = (x0 + k < dw) ? x0 + k : dw - 1;
    const Tap tx = make_tap<INTERP>(x, scx, sw);
#pragma unroll
    for (int c = 0; c < CH; c++)
      o[k * CH + c] = bilerp(r0[CH * tx.i0 + c], r0[CH * tx.i1 + c], r1[CH * tx.i0 + c], r1[CH * tx.i1 + c], tx.f, ty.f);
  }
  uint8_t* out = dst + (size_t)y * dp + (size_t)CH * x0;
  if (vec_ok && x0 + 4 <= dw) {
    if constexpr (CH == 3) {
      { uint32_t d0, d1, d2; pack12_trunc(o, d0, d1, d2); stg3<true>(out, d0, d1, d2); }
    } else if constexpr (CH == 2) {
      stg<true, u32x2>(out, u32x2{pack4_trunc(o[0], o[1], o[2], o[3]), pack4_trunc(o[4], o[5], o[6], o[7])});
    } else {
      stg<true, uint32_t>(out, pack4_trunc(o[0], o[1], o[2], o[3]));
    }
  } else {
    const uint32_t nv = (dw - x0 < 4 ? dw - x0 : 4) * CH;
    for (uint32_t i = 0; i < nv; i++) out[i] = (uint8_t)sat_trunc(o[i]);
  }
}

template <int CH, int INTERP>
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh,
                                                uint8_t* __restrict__ dst, uint32_t dp, uint32_t dw, uint32_t dh,
                                                float scx, float scy, int vec_ok) {
  GatherTask<CH, INTERP>::run(src, sp, dst, dp, PlaneGeom{sw, sh, dw, dh, scx, scy, vec_ok, 0, 0, 0, 0}, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------
// Lanczos-3 (VPF_INTERP_LANCZOS3): what the reference's resizer asks NPP for (Tasks.cpp:1190).  Textbook separable
// 6 x 6 taps, weights normalised, indices clamped, no widening when minifying.  sin / cos come from fixed fma
// polynomials and the six taps from angle-addition identities so the host oracle reproduces the weights bit for bit
// (the test oracle restates the same sequence).  Gather kernel: lane = one destination pixel.
// ------------------------------------------------------------------------------------------
VPF_DEV uint32_t pack_i16(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
typedef short s16x2 __attribute__((ext_vector_type(2)));
VPF_DEV int32_t dot2(uint32_t a, uint32_t b, int32_t c) {  // a.lo * b.lo + a.hi * b.hi + c on int16 halves
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
// the first dot of a chain: the VOP3P encoding takes the inline constant 0 as its addend (the compiler only emits the accumulate-in-place
// VOP2 form v_dot2c_i32_i16, which wants a zeroed register per chain: three v_mov_b32 per source row of the tiled kernel)
VPF_DEV int32_t dot2z(uint32_t a, uint32_t b) {
  int32_t r;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int CH>
struct LanczosGatherTask {
  static constexpr int kThreads = 256;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by);
};
template <int CH>
__global__ __launch_bounds__(256) void k_resize_lanczos(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh,
                                                        uint8_t* __restrict__ dst, uint32_t dp, uint32_t dw, uint32_t dh,
                                                        float scx, float scy) {
  LanczosGatherTask<CH>::run(src, sp, dst, dp, PlaneGeom{sw, sh, dw, dh, scx, scy, 0, 0, 0, 0, 0}, blockIdx.x, blockIdx.y);
}
template <int CH>
VPF_DEV void LanczosGatherTask<CH>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G,
                                        uint32_t bx, uint32_t by) {
  const uint32_t sw = G.sw, sh = G.sh, dw = G.dw, dh = G.dh;
  const float scx = G.scx, scy = G.scy;
  const uint32_t x = bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const QTap tx = quantize_ltap(make_ltap(x, scx)), ty = quantize_ltap(make_ltap(y, scy));
  uint32_t xi[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int32_t i = tx.i0 + k - 2;
    xi[k] = (uint32_t)(i < 0 ? 0 : (i > (int32_t)sw - 1 ? (int32_t)sw - 1 : i)) * CH;
  }
  // vertical pass on byte-wide partial products (the arithmetic of the matrix-core kernel, restated by the oracle): z = Hr - 8192 and the
  // row's weight q as two signed bytes each, q z without its lowest partial product ql zl; taps the edge clamp puts on the same source row
  // act as one tap with the summed weight
  int32_t acc[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) acc[c] = (1 << 19) + (1 << 11);
  // Horizontal sums of one source row.  The six taps of a pixel are 6 CH CONTIGUOUS bytes (unless the edge clamp folds some of them onto
  // the border pixel), so the fast path fetches them as ONE window of dwords from the 4-B aligned address below the first tap — 1 / 1 / 2
  // load instructions per row for 1 / 2 / 3 channels instead of 6 / 12 / 18 byte loads: with byte loads the kernel was bound by the rate
  // at which the memory pipeline takes instructions, nothing else —, shifts the window down to the first tap with v_alignbyte_b32, spreads
  // two taps of a channel into int16 halves with one v_perm_b32 and takes them two at a time with v_dot2_i32_i16 against the packed Q14
  // weights.  Same exact integer sums as the byte-wise path, which the lanes next to the left / right image edge (clamped taps), windows
  // that would reach past the row's last dword, and rows that are not 4-B aligned keep.
  constexpr int NW = (6 * CH + 3) / 4;        // dwords of tap bytes: 2 / 3 / 5
  constexpr int ND = NW + 1;                  // dwords loaded: the window may start up to 3 bytes below the first tap
  const uint32_t b0 = (uint32_t)CH * (uint32_t)(tx.i0 - 2), a0 = b0 & ~3u;
  const bool lane_fast = (((uintptr_t)src | sp) & 3u) == 0 && tx.i0 - 2 >= 0 && tx.i0 + 3 <= (int32_t)sw - 1 && a0 + 4u * ND <= (((uint32_t)CH * sw + 3u) & ~3u);
  const bool fast = __builtin_amdgcn_ballot_w64(!lane_fast) == 0;  // decided per WAVE: the two waves at a row's ends take the byte-wise path whole, none runs both
  const uint32_t q01 = pack_i16(tx.q[0], tx.q[1]), q23 = pack_i16(tx.q[2], tx.q[3]), q45 = pack_i16(tx.q[4], tx.q[5]);
  auto row_sums = [&](const uint8_t* r, int32_t (&h)[CH]) {
    if (fast) {
      uint32_t d[ND];
      const uint32_t* p = reinterpret_cast<const uint32_t*>(r + a0);
#pragma unroll
      for (int i = 0; i < ND; i++) d[i] = p[i];
      uint32_t w[NW + 1];
#pragma unroll
      for (int i = 0; i < NW; i++) w[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], b0 & 3u);
      w[NW] = w[NW - 1];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        uint32_t pr[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int B0 = 2 * k * CH + c, B1 = (2 * k + 1) * CH + c, W0 = B0 >> 2;                 // tap 2 k in the low half, tap 2 k + 1 in the high half
          const uint32_t sel = (uint32_t)(B0 & 3) | 0x0c00u | ((uint32_t)((B1 >> 2) == W0 ? (B1 & 3) : 4 + (B1 & 3)) << 16) | 0x0c000000u;
          pr[k] = __builtin_amdgcn_perm(w[W0 + 1], w[W0], sel);
        }
        h[c] = dot2(pr[2], q45, dot2(pr[1], q23, dot2z(pr[0], q01)));
      }
    } else {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        int32_t t = 0;  // exact: |t| <= 255 * sum |q| < 2^24
#pragma unroll
        for (int kx = 0; kx < 6; kx++) t += __mul24(tx.q[kx], (int32_t)r[xi[kx] + c]);  // 16-bit x 8-bit: v_mad_i32_i24 (a 32-bit multiply is quarter rate)
        h[c] = t;
      }
    }
  };
  int32_t qrun = 0;
#pragma unroll
  for (int ky = 0; ky < 6; ky++) {
    const int32_t j = ty.i0 + ky - 2, hi = (int32_t)sh - 1;
    const int32_t row = j < 0 ? 0 : (j > hi ? hi : j), nxt = j + 1 < 0 ? 0 : (j + 1 > hi ? hi : j + 1);
    qrun += ty.q[ky];
    if (ky < 5 && nxt == row) continue;  // the run goes on: its last tap carries the sum
    int32_t h[CH];
    row_sums(src + (size_t)row * sp, h);
    const int32_t ql = ((qrun + 128) & 0xff) - 128;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int32_t z = ((h[c] + 128) >> 8) - 8192, zl = ((z + 128) & 0xff) - 128;
      acc[c] += (__mul24(qrun, z) - __mul24(ql, zl)) >> 8;  // a multiple of 256: exact (|qrun| < 2^15, |z| < 2^15)
    }
    qrun = 0;
  }
  uint8_t* o = dst + (size_t)y * dp + (size_t)CH * x;
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int32_t v = acc[c] >> 12;  // (V / 256 + 2^19 + 2^11) >> 12: round half up
    o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

// ------------------------------------------------------------------------------------------
// LDS-staged variant (the default when it applies): a wave produces 256 consecutive destination pixels of one
// row.  It first copies the two source rows' byte span it needs into a wave-private LDS strip with fully coalesced
// dword loads (each source byte crosses the memory pipeline once, as part of a dense 256-B wave load, instead of
// 24 scattered byte loads per lane), then every lane picks its taps out of LDS with ds_read_u8 (cheap: ~0.6 us of
// LDS issue per 720p frame chip-wide).  Same make_tap / bilerp arithmetic => bit-identical to k_resize.
// Requires: 16-B aligned source rows (pointer and pitch); span of a wave <= kResizeRowBytes (host checks the scale).
// ------------------------------------------------------------------------------------------
template <int CH, int IT /* 1-KiB loads per strip */>
struct RowPairTask {
  static constexpr int kThreads = 256;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by);
};
template <int CH, int IT>
__global__ __launch_bounds__(256) void k_resize_lds(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh,
                                                    uint8_t* __restrict__ dst, uint32_t dp, uint32_t dw, uint32_t dh,
                                                    float scx, float scy, int vec_ok, uint32_t rowq /* strip size in 16-B units */) {
  RowPairTask<CH, IT>::run(src, sp, dst, dp, PlaneGeom{sw, sh, dw, dh, scx, scy, vec_ok, rowq, 0, 0, 0}, blockIdx.x, blockIdx.y);
}
template <int CH, int IT>
VPF_DEV void RowPairTask<CH, IT>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G,
                                      uint32_t bx, uint32_t by) {
  const uint32_t sw = G.sw, sh = G.sh, dw = G.dw, dh = G.dh, rowq = G.a0;
  const float scx = G.scx, scy = G.scy;
  const int vec_ok = G.vec_ok;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t y = by * 4 + wv;
  if (y >= dh || bx * 256 >= dw) return;
  const uint32_t xs = bx * 256, xe = (xs + 255 < dw - 1) ? xs + 255 : dw - 1;  // this wave's dst columns [xs, xe]
  const Tap ty = make_tap<VPF_INTERP_LINEAR>(y, scy, sh);
  const uint32_t first = make_tap<VPF_INTERP_LINEAR>(xs, scx, sw).i0, last = make_tap<VPF_INTERP_LINEAR>(xe, scx, sw).i1;
  const uint32_t base = (CH * first) & ~15u, nq = (CH * (last + 1) - base + 15) / 16;
  // exact-alignment shortcuts (bit-identical: fma(0, finite, t) == t): fy == 0 for the whole row (odd integer vertical
  // scale, e.g. 4K -> 720p) -> the second source row is neither loaded nor read; fx == 0 in every lane -> one tap per row
  const bool row1 = __builtin_amdgcn_readfirstlane(__float_as_uint(ty.f)) != 0u;
  Span<IT> s0, s1;
  s0.load(src + (size_t)ty.i0 * sp, base, nq, lane);
  if (row1) s1.load(src + (size_t)ty.i1 * sp, base, nq, lane);
  u32x4* st0 = dyn_strip + (wv * 2) * rowq;
  u32x4* st1 = st0 + rowq;
  s0.store(st0, nq, lane);
  if (row1) s1.store(st1, nq, lane);
  wave_lds_sync();
  const uint32_t x0 = xs + lane * 4;
  if (x0 >= dw) return;
  const uint8_t* r0 = reinterpret_cast<const uint8_t*>(st0);
  const uint8_t* r1 = reinterpret_cast<const uint8_t*>(st1);
  if constexpr (CH == 3) {
    // Odd integer scale factors (kernel-uniform on x, row-uniform on y; 4K -> 720p is 3x): every destination pixel IS the
    // source pixel k x + (k - 1) / 2 — exactly, in float as well (all values < 2^24) — and the general path's float round
    // trip (+ 0.5, truncate) returns its bytes unchanged.  So the bytes are moved as bytes: one 8-B LDS window +
    // v_alignbyte_b32 per pixel, three v_perm_b32 per four pixels.
    const uint32_t kx = (uint32_t)scx;
    if (!row1 && (float)kx == scx && (kx & 1u) && sw == kx * dw && vec_ok && x0 + 4 <= dw) {
      uint32_t e[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t a = 3 * (kx * (x0 + k) + (kx >> 1)) - base;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(r0 + (a & ~3u));
        e[k] = __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);  // R G B of the pixel in bytes 0..2
      }
      stg3<true>(dst + (size_t)y * dp + 3 * (size_t)x0, __builtin_amdgcn_perm(e[1], e[0], 0x04020100u),
                  __builtin_amdgcn_perm(e[2], e[1], 0x05040201u), __builtin_amdgcn_perm(e[3], e[2], 0x06050402u));
      return;
    }
  }
  float o[4 * CH];
  const ColTaps<CH> T = make_col_taps<CH>(base, x0, dw, sw, scx);
  rowpair_blend4<CH>(r0, r1, row1, ty.f, T, o);
  store_blend4<CH>(dst + (size_t)y * dp + (size_t)CH * x0, o, vec_ok && x0 + 4 <= dw, dw - x0 < 4 ? dw - x0 : 4);
}

// ------------------------------------------------------------------------------------------
// Row band: the row-pair blend over R consecutive destination rows per wave, for vertical scale factors up to 2 (down) and any up-scale.
// What a row-pair wave spends outside the blend proper — the column taps (53 of its ~360 VALU instructions at CH = 3), the strip
// geometry, a memory round trip — depends on the columns only, and below 2x neighbouring destination rows share source rows.  A band
// wave stages the contiguous source rows [i0(first row), i1(last row)] once (all loads in flight together; the launcher sizes LDS for
// the at most floor((R - 1) scy) + 3 rows of a band), computes the column taps once, and walks down its destination rows keeping the
// HORIZONTAL lerps of the two current source rows in registers: a source row's lerp fma(fx, p1 - p0, p0) is evaluated once per band and
// reused by every destination row that blends it (1.5 evaluations per destination row at a 1.5x down-scale instead of 2, 0.5 at a 2x
// up-scale), leaving fma(fy, bot - top, top) + 0.5 per row.  Same operations on the same operands as bilerp() -> the bytes of
// RowPairTask and of the oracle (zero weights need no special case: fma(0, finite, t) == t and every staged byte is finite).
// Used where the launch still has plenty of workgroups (batches); a single small frame keeps one row per wave.
// ------------------------------------------------------------------------------------------
// destination pixels per lane: 4, or 8 = 512 columns per wave on 1-channel planes in the `P1 = 8` instantiations (a wave's per-row fixed
// work is the same whatever the channel count; the launcher picks them when 512-column chunks fill the planes' rows well)
constexpr int band_px(int ch, int p1) { return ch == 1 ? p1 : 4; }
template <int CH, int R, int IT = 2 /* 1-KiB staging passes per strip */, int P1 = 4 /* pixels per lane on 1-channel planes */, bool MULTI = false /* several bands per wave */>
struct RowBandTask {
  static constexpr int kThreads = 256;
  static constexpr int kSlots = kBandSlots * 2 / IT < 2 * R + 1 ? kBandSlots * 2 / IT : 2 * R + 1;  // a band touches at most floor((R - 1) scy) + 3 source rows, scy <= 2
  static constexpr int kPx = band_px(CH, P1);
  static constexpr bool kPx4 = CH == 3 && R == 16 && IT == 1;  // strips hold four bytes per pixel (band_rows sizes them: vpf_bound_strip_bytes_px4)
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by) {
    run_w(src, sp, dst, dp, G, bx, by * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
  }
  // one wave's share: column chunk bx, wave row wrow
  static VPF_DEV void run_w(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t wrow);
#ifdef VPF_LAB_FORMS
  // the persistent launch (k_planes_mp_persist, k_resize_common.h): chunk after chunk while the stream hands out chunks of CH-channel planes
  template <class Stream>
  static VPF_DEV void run_chunks(BandChunk& c, Stream& ts, const PlaneTable& PT);
#endif
};
template <int CH, int R, int IT, int P1, bool MULTI>
VPF_DEV void RowBandTask<CH, R, IT, P1, MULTI>::run_w(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G,
                                           uint32_t bx, uint32_t wrow) {
  constexpr int PX = kPx;
  constexpr uint32_t W = 64 * PX;  // destination columns per wave
  const uint32_t sw = G.sw, sh = G.sh, dw = G.dw, dh = G.dh, rowq = G.a0, slots = G.a1;
  const uint32_t nb = (MULTI && G.a2) ? G.a2 : 1u;  // bands per wave (launcher): the wave walks down nb consecutive bands of its columns
  const float scx = G.scx, scy = G.scy;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint32_t ya = wrow * (R * nb);
  if (ya >= dh || bx * W >= dw) return;
  const uint32_t xs = bx * W, xe = (xs + W - 1 < dw - 1) ? xs + W - 1 : dw - 1;
  const uint32_t first = make_tap<VPF_INTERP_LINEAR>(xs, scx, sw).i0, last = make_tap<VPF_INTERP_LINEAR>(xe, scx, sw).i1;
  // packed RGB in 16-row bands (up-scales: a source pixel is tapped ~4 times per row it is staged): "px4" strips (k_bilinear_blend.h) — pixels
  // [first & ~3, last + 1] widened to R G B x in units of four; else the packed bytes in 16-B units.  (Down-scales tap a staged pixel ~1.3
  // times and pay a third more LDS per row — 1080p -> 720p: four workgroups per CU instead of five, 1.71 -> 2.33 us; profiles/r06_c_*.)
  constexpr bool X4 = kPx4;
  const uint32_t base_px = first & ~3u;
  const uint32_t base = X4 ? 3u * base_px : (CH * first) & ~15u, nq = X4 ? (last + 2 - base_px + 3) / 4 : (CH * (last + 1) - base + 15) / 16;
  const uint32_t lim = (CH * sw + 15u) & ~15u;  // bytes of a source row that may be read
  u32x4* const strips = dyn_strip + (size_t)wv * slots * rowq;
  // the source rows [i0(first row), i1(last row)] of one band (the launcher guarantees r_hi - r_lo < slots <= kSlots) ...
  uint32_t yb, r_lo, r_hi;
  auto rows_of = [&](uint32_t y0) {
    yb = (y0 + R - 1 < dh - 1) ? y0 + R - 1 : dh - 1;
    r_lo = __builtin_amdgcn_readfirstlane(make_tap<VPF_INTERP_LINEAR>(y0, scy, sh).i0);
    r_hi = __builtin_amdgcn_readfirstlane(make_tap<VPF_INTERP_LINEAR>(yb, scy, sh).i1);
  };
  // ... requested (all loads in flight together) ...
  typename std::conditional<X4, Span4<IT>, Span<IT>>::type rows[kSlots];
  auto request = [&]() {
#pragma unroll
    for (int k = 0; k < kSlots; k++)
      if (r_lo + k <= r_hi) {
        if constexpr (X4) rows[k].load(src + (size_t)(r_lo + k) * sp, base, nq, lane, lim);
        else rows[k].load(src + (size_t)(r_lo + k) * sp, base, nq, lane);
      }
  };
  // ... and written to the wave's strips
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < kSlots; k++)
      if (r_lo + k <= r_hi) rows[k].store(strips + (size_t)k * rowq, nq, lane);
    wave_lds_sync();
  };
  const uint32_t x0 = xs + lane * PX;
  const bool draws = x0 < dw;  // lanes past the picture's right edge stage, but blend nothing
  const bool vec4 = G.vec_ok && x0 + PX <= dw;
  const uint32_t nv = !draws ? 0u : dw - x0 < (uint32_t)PX ? dw - x0 : (uint32_t)PX;
  rows_of(ya);
  request();
  // once for all rows of all bands (evaluated while the first band's rows are in flight)
  typename std::conditional<X4, ColTapsX, ColTaps<CH, PX>>::type T;
  if constexpr (X4) T = make_col_taps_x(base_px, x0, dw, sw, scx);
  else T = make_col_taps<CH, PX>(base, x0, dw, sw, scx);
  VPF_WAVE_MARK(0);  // (lab builds: setup done)
  // With nb > 1 the source rows of band k + 1 are requested right after band k's have been written to the strips — the registers are free
  // again — and arrive while band k is blended: the wave hides its own memory latency, and its fixed part (task decode, column taps) is
  // paid once per nb bands.  Measured on 1- and 2-channel planes, whose waves are short (profiles/r04_bilinear_ablate.txt: staging, blend
  // and the fixed part are nearly additive there).
  BandWalk<CH, PX> walk;
  for (uint32_t k = 0; k < nb; k++) {
    commit();
    VPF_WAVE_MARK(1);  // (lab builds: the first band's rows have arrived and sit in LDS)
    const Tap row_taps = band_row_taps(ya, yb, scy, sh);  // every lane active here
    const uint32_t ya_k = ya, yb_k = yb, r_lo_k = r_lo;
    ya += R;
    const bool more = k + 1 < nb && ya < dh;
    if (more) { rows_of(ya); request(); }
    if (draws) {
      band_blend_rows<CH, R, PX>(reinterpret_cast<const uint8_t*>(strips), rowq * 16, r_lo_k, ya_k, yb_k, row_taps, T, walk, [&](uint32_t y, const float* o) {
        store_blend4<CH, PX>(dst + (size_t)y * dp + (size_t)CH * x0, o, vec4, nv);
      });
    }
    VPF_WAVE_MARK(2);  // (lab builds: the first band is blended and its stores are issued)
    if (!more) return;
    wave_lds_sync();  // the blend's LDS reads are done before the next band's rows overwrite the strips
  }
}

#ifdef VPF_LAB_FORMS
// The same walk for the persistent launch: a wave works through CHUNKS (runs of bands of one column chunk) that a ChunkStream hands out.  Inside
// a chunk it is the march form; at a chunk's last band the NEXT chunk — any frame, plane (of the same channel count) or column — is fetched from
// the stream and ITS first rows are requested before this band is blended and stored; the column taps and the walk are rebuilt after the blend.
template <int CH, int R, int IT, int P1, bool MULTI>
template <class Stream>
VPF_DEV void RowBandTask<CH, R, IT, P1, MULTI>::run_chunks(BandChunk& c, Stream& ts, const PlaneTable& PT) {
  constexpr int PX = kPx;
  constexpr uint32_t W = 64 * PX;
  constexpr bool X4 = kPx4;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t rowq = PT.g[c.pi].a0, slots = PT.g[c.pi].a1;  // one strip size for the launch
  u32x4* const strips = dyn_strip + (size_t)wv * slots * rowq;
  // the staging side: the chunk whose source rows are being requested ...
  const uint8_t* s_src = nullptr;
  uint32_t s_sp = 0, s_sh = 0, s_dh = 0, s_base = 0, s_nq = 0, s_lim = 0, s_bpx = 0;
  float s_scy = 0.f;
  uint32_t ya = 0, y_end = 0, yb = 0, r_lo = 0, r_hi = 0;
  auto stage_of = [&](const BandChunk& k) {
    const PlaneGeom& g = PT.g[k.pi];
    const uint32_t xs = k.bx * W, xe = (xs + W - 1 < g.dw - 1) ? xs + W - 1 : g.dw - 1;
    const uint32_t first = __builtin_amdgcn_readfirstlane(make_tap<VPF_INTERP_LINEAR>(xs, g.scx, g.sw).i0), last = __builtin_amdgcn_readfirstlane(make_tap<VPF_INTERP_LINEAR>(xe, g.scx, g.sw).i1);
    s_bpx = first & ~3u;
    s_base = X4 ? 3u * s_bpx : (CH * first) & ~15u;
    s_nq = X4 ? (last + 2 - s_bpx + 3) / 4 : (CH * (last + 1) - s_base + 15) / 16;
    s_lim = (CH * g.sw + 15u) & ~15u;
    s_src = k.src; s_sp = k.sp; s_sh = g.sh; s_dh = g.dh; s_scy = g.scy;
    ya = k.band0 * R;
    y_end = (k.band0 + k.nb) * R < g.dh ? (k.band0 + k.nb) * R : g.dh;
  };
  auto rows_of = [&](uint32_t y0) {
    yb = (y0 + R - 1 < s_dh - 1) ? y0 + R - 1 : s_dh - 1;
    r_lo = __builtin_amdgcn_readfirstlane(make_tap<VPF_INTERP_LINEAR>(y0, s_scy, s_sh).i0);
    r_hi = __builtin_amdgcn_readfirstlane(make_tap<VPF_INTERP_LINEAR>(yb, s_scy, s_sh).i1);
  };
  typename std::conditional<X4, Span4<IT>, Span<IT>>::type rows[kSlots];
  auto request = [&]() {
#pragma unroll
    for (int k = 0; k < kSlots; k++)
      if (r_lo + k <= r_hi) {
        if constexpr (X4) rows[k].load(s_src + (size_t)(r_lo + k) * s_sp, s_base, s_nq, lane, s_lim);
        else rows[k].load(s_src + (size_t)(r_lo + k) * s_sp, s_base, s_nq, lane);
      }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < kSlots; k++)
      if (r_lo + k <= r_hi) rows[k].store(strips + (size_t)k * rowq, s_nq, lane);
    wave_lds_sync();
  };
  // ... and the drawing side: the chunk whose bands are being blended
  uint8_t* d_dst = nullptr;
  uint32_t d_dp = 0, d_x0 = 0, d_nv = 0;
  bool d_draws = false, d_vec4 = false;
  typename std::conditional<X4, ColTapsX, ColTaps<CH, PX>>::type T;
  auto draw_of = [&](const BandChunk& k) {  // (after stage_of(k): s_base / s_bpx are k's)
    const PlaneGeom& g = PT.g[k.pi];
    d_x0 = k.bx * W + lane * PX;
    d_draws = d_x0 < g.dw;
    d_vec4 = g.vec_ok && d_x0 + PX <= g.dw;
    d_nv = !d_draws ? 0u : g.dw - d_x0 < (uint32_t)PX ? g.dw - d_x0 : (uint32_t)PX;
    d_dst = k.dst; d_dp = k.dp;
    if constexpr (X4) T = make_col_taps_x(s_bpx, d_x0, g.dw, g.sw, g.scx);
    else T = make_col_taps<CH, PX>(s_base, d_x0, g.dw, g.sw, g.scx);
  };
  stage_of(c);
  rows_of(ya);
  request();
  draw_of(c);
  BandWalk<CH, PX> walk;
  for (;;) {
    commit();
    const Tap row_taps = band_row_taps(ya, yb, s_scy, s_sh);  // every lane active here
    const uint32_t ya_k = ya, yb_k = yb, r_lo_k = r_lo;
    ya += R;
    const bool same = ya < y_end;
    BandChunk nx{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
    bool cross = false;
    if (same) {
      rows_of(ya);
      request();
    } else {
      nx = ts.next();
      cross = nx.nb != 0 && PT.ch[nx.pi] == (uint32_t)CH;
      if (cross) { stage_of(nx); rows_of(ya); request(); }
    }
    if (d_draws) {
      uint8_t* const dst = d_dst;
      const uint32_t dp = d_dp, x0 = d_x0, nv = d_nv;
      const bool vec4 = d_vec4;
      band_blend_rows<CH, R, PX>(reinterpret_cast<const uint8_t*>(strips), rowq * 16, r_lo_k, ya_k, yb_k, row_taps, T, walk, [&](uint32_t y, const float* o) {
        store_blend4<CH, PX>(dst + (size_t)y * dp + (size_t)CH * x0, o, vec4, nv);
      });
    }
    if (!same) {
      if (!cross) { c = nx; return; }  // (the caller's wave_lds_sync stands between this blend and the next task's strips)
      draw_of(nx);
      walk = BandWalk<CH, PX>();
    }
    wave_lds_sync();  // the blend's LDS reads are done before the next band's rows overwrite the strips
  }
}
#endif  // VPF_LAB_FORMS

// ------------------------------------------------------------------------------------------
// Tiled, separable BILINEAR resize for up-scales (8-bit Lanczos-3 lives in k_lanczos_mfma.hip; k_resize_lanczos / k_resize remain the
// any-input gather forms).  The horizontal lerp of (source row r, destination column x) does not depend on the destination row, and
// in an up-scale several destination rows sit between the same two source rows, so a workgroup that owns a tile of 64 columns x TY
// rows computes each of them ONCE:
//   phase 0  the tile's whole source window goes to LDS in one sweep of dense 16-B loads;
//   phase 1  wave w takes source rows w, w + WPB, ...: H[row][channel][column] = fma(fx, p1 - p0, p0) (k_resize's bilerp) as floats in LDS;
//   phase 2  every lane blends two H rows per destination pixel: fma(fy, bottom - top, top) + 0.5, truncating pack.
// Same arithmetic as the gather forms -> bit-identical to them and to the oracle.  2x up-scale: 0.7 horizontal lerps per
// destination pixel instead of 2.
// WPB = waves per workgroup (4 or 8): 8 halves the number of source rows a wave walks through one after the other, which is what
// the duration of a single-frame launch (one round of workgroups, each a chain of dependent steps) is made of.
// TileTaskF32 below is the same tiling for float surfaces (bilinear and Lanczos-3).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kLzStripQ = 128;  // at most 2 KiB of source bytes per row of a tile

constexpr int kTileStagePasses = 6;  // staging loads a thread may have in flight (the launcher sizes tiles accordingly)

template <int CH, int WPB>
struct TileTask {
  static constexpr int kThreads = 64 * WPB;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& P, uint32_t bx, uint32_t by);
};
template <int CH, int WPB>
VPF_DEV void TileTask<CH, WPB>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& P,
                                    uint32_t bx, uint32_t by) {
  // dynamic LDS: RAW[nr_cap][rowq x 16 B] source bytes | H[nr_cap][CH][64] floats | WY[tile_rows][8] floats (fy, -, ..., top H row, bottom H row)
  constexpr uint32_t T = 64 * WPB;
  const uint32_t sw = P.sw, sh = P.sh, dw = P.dw, dh = P.dh, tile_rows = P.a0, nr_cap = P.a1, rowq = P.a2, lshift = P.a3;
  const float scx = P.scx, scy = P.scy;
  u32x4* const RAW = dyn_strip;
  float* const H = reinterpret_cast<float*>(dyn_strip + (size_t)nr_cap * rowq);
  float* const WY = H + (size_t)nr_cap * CH * 64;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t xf = bx * 64, xl = (xf + 63 < dw - 1) ? xf + 63 : dw - 1;
  if (xf >= dw) return;  // a narrower plane of a multi-plane launch (workgroup-uniform)
  const uint32_t x = xf + lane, xc = x < dw ? x : dw - 1;  // lanes past the right edge compute a duplicate, never stored
  const uint32_t y0 = by * tile_rows, yl = (y0 + tile_rows - 1 < dh - 1) ? y0 + tile_rows - 1 : dh - 1;
  auto clampi = [](int32_t i, int32_t hi) { return (uint32_t)(i < 0 ? 0 : (i > hi ? hi : i)); };
  const int32_t R0 = (int32_t)make_tap<VPF_INTERP_LINEAR>(y0, scy, sh).i0, R1 = (int32_t)make_tap<VPF_INTERP_LINEAR>(yl, scy, sh).i1;
  const uint32_t first = make_tap<VPF_INTERP_LINEAR>(xf, scx, sw).i0, last = make_tap<VPF_INTERP_LINEAR>(xl, scx, sw).i1;
  const uint32_t nrows = (uint32_t)(R1 - R0 + 1);
  const uint32_t base = (CH * first) & ~15u, nq = (CH * (last + 1) - base + 15) / 16;
  // Phase 0: the tile's whole source window goes to LDS in ONE sweep — every 16-B unit of every source row is requested before the
  // first one is waited for, so the workgroup pays one memory latency, not one per source row.  The weights are computed while the
  // loads are in flight.
  const uint32_t scol = threadIdx.x & ((1u << lshift) - 1u), srow0 = threadIdx.x >> lshift, srows = T >> lshift;
  u32x4 stage[kTileStagePasses];
#pragma unroll
  for (int k = 0; k < kTileStagePasses; k++) {
    const uint32_t r = srow0 + k * srows;
    if (r < nrows && scol < nq) stage[k] = ldg<false, u32x4>(src + (size_t)clampi(R0 + (int32_t)r, (int32_t)sh - 1) * sp + base + 16 * scol);
  }
  if (threadIdx.x < tile_rows) {  // vertical weights: one lane per destination row
    const uint32_t y = y0 + threadIdx.x, yc = y < dh ? y : dh - 1;
    const Tap t = make_tap<VPF_INTERP_LINEAR>(yc, scy, sh);
    WY[threadIdx.x * 8] = t.f;
    WY[threadIdx.x * 8 + 6] = __int_as_float((int32_t)t.i0 - R0);
    WY[threadIdx.x * 8 + 7] = __int_as_float((int32_t)t.i1 - R0);
  }
#pragma unroll
  for (int k = 0; k < kTileStagePasses; k++) {
    const uint32_t r = srow0 + k * srows;
    if (r < nrows && scol < nq) RAW[r * rowq + scol] = stage[k];
  }
  __syncthreads();
  const Tap tx = make_tap<VPF_INTERP_LINEAR>(xc, scx, sw);
  const uint32_t xo0 = tx.i0 * CH - base, xo1 = tx.i1 * CH - base;
  const float wx = tx.f;
  // Phase 1, horizontal: wave w takes source rows w, w + WPB, ...; rows are independent of one another (no barrier inside the loop)
  for (uint32_t r = wv; r < nrows; r += WPB) {
    const uint8_t* b = reinterpret_cast<const uint8_t*>(RAW + (size_t)r * rowq);
    if constexpr (CH == 3) {
      float t0[3], t1[3];  // at the right image edge i1 == i0 and the window's second tap is junk with weight exactly 0
      strip_window_taps(b, xo0, t0, t1);
#pragma unroll
      for (int c = 0; c < 3; c++) H[(r * 3 + c) * 64 + lane] = __builtin_fmaf(wx, t1[c] - t0[c], t0[c]);
    } else {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const float p0 = (float)b[xo0 + c], p1 = (float)b[xo1 + c];
        H[(r * CH + c) * 64 + lane] = __builtin_fmaf(wx, p1 - p0, p0);
      }
    }
  }
  __syncthreads();
  // phase 2: a lane owns 4 consecutive columns of one destination row (a wave = 4 rows x 64 columns): H comes out of LDS
  // as one ds_read_b128 per (tap, channel) and the pixels leave as one 4-12 B vector store per lane
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const uint32_t cg = lane & 15, rsub = lane >> 4, x0 = xf + 4 * cg;
  for (uint32_t yb = 0; yb < tile_rows; yb += 4 * WPB) {
    const uint32_t yy = yb + wv * 4 + rsub, y = y0 + yy;
    if (yy >= tile_rows || y >= dh || x0 >= dw) continue;
    const uint32_t r0 = (uint32_t)__float_as_int(WY[yy * 8 + 6]), r1 = (uint32_t)__float_as_int(WY[yy * 8 + 7]);
    // packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two independent IEEE operations per instruction at the issue cost of one —
    // tools/probe_valu_rate.hip, profiles/r02_probe_valu_rate.txt); every component goes through exactly the scalar form's
    // operations, so results are bit-identical to it
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[CH][2];
    const float fy = WY[yy * 8];
    const f32x2 fy2 = {fy, fy};
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const f32x4 top = *reinterpret_cast<const f32x4*>(&H[(r0 * CH + c) * 64 + 4 * cg]);
      const f32x4 bot = *reinterpret_cast<const f32x4*>(&H[(r1 * CH + c) * 64 + 4 * cg]);
      const f32x2 t0 = {top[0], top[1]}, t1 = {top[2], top[3]}, b0 = {bot[0], bot[1]}, b1 = {bot[2], bot[3]};
      acc[c][0] = __builtin_elementwise_fma(fy2, b0 - t0, t0);
      acc[c][1] = __builtin_elementwise_fma(fy2, b1 - t1, t1);
    }
    float o[4 * CH];  // pixel-major; + 0.5, then the truncating pack of every bilinear kernel
#pragma unroll
    for (int c = 0; c < CH; c++)
#pragma unroll
      for (int hlf = 0; hlf < 2; hlf++) {
        const f32x2 v = acc[c][hlf] + f32x2{0.5f, 0.5f};
        o[(2 * hlf) * CH + c] = v[0]; o[(2 * hlf + 1) * CH + c] = v[1];
      }
    uint8_t* out = dst + (size_t)y * dp + (size_t)CH * x0;
    if (P.vec_ok && x0 + 4 <= dw) {
      if constexpr (CH == 3) {
        { uint32_t d0, d1, d2; pack12_trunc(o, d0, d1, d2); stg3<true>(out, d0, d1, d2); }
      } else if constexpr (CH == 2) {
        stg<true, u32x2>(out, u32x2{pack4_trunc(o[0], o[1], o[2], o[3]), pack4_trunc(o[4], o[5], o[6], o[7])});
      } else {
        stg<true, uint32_t>(out, pack4_trunc(o[0], o[1], o[2], o[3]));
      }
    } else {
      const uint32_t nv = (dw - x0 < 4 ? dw - x0 : 4) * CH;
      for (uint32_t i = 0; i < nv; i++) out[i] = (uint8_t)sat_trunc(o[i]);
    }
  }
}

// The same tiling for 32-bit float surfaces (RGB_32F: CH = 3 interleaved, RGB_32F_PLANAR: CH = 1 per plane; reference
// NppResizeSurfacePacked32F3C_Impl / NppResizeSurface32FPlanar_Impl, Tasks.cpp:1334-1445).  Samples are floats, so both passes are
// fp32 fma chains in FloatGatherTask's order (tap 0 first, row 0 first, accumulators starting at 0; bilinear: its two lerps): results
// are bit-identical to the gather form; nothing is rounded or clamped.  Geometry fields as for TileTask, with 4-byte samples; the
// window is staged in rounds of kTileStagePasses units per thread (a float row is four times as long as a byte row).
template <int CH, bool LZ, int WPB>
struct TileTaskF32 {
  static constexpr int kThreads = 64 * WPB;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& P, uint32_t bx, uint32_t by);
};
template <int CH, bool LZ, int WPB>
VPF_DEV void TileTaskF32<CH, LZ, WPB>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& P,
                                           uint32_t bx, uint32_t by) {
  // dynamic LDS: RAW[nr_cap][rowq x 16 B] source floats | H[nr_cap][CH][64] floats | WY[tile_rows][8] floats | WX[7][64] floats (Lanczos)
  constexpr int NT = LZ ? 6 : 2;
  constexpr uint32_t T = 64 * WPB;
  const uint32_t sw = P.sw, sh = P.sh, dw = P.dw, dh = P.dh, tile_rows = P.a0, nr_cap = P.a1, rowq = P.a2, lshift = P.a3;
  const float scx = P.scx, scy = P.scy;
  u32x4* const RAW = dyn_strip;
  float* const H = reinterpret_cast<float*>(dyn_strip + (size_t)nr_cap * rowq);
  float* const WY = H + (size_t)nr_cap * CH * 64;
  float* const WX = WY + (size_t)tile_rows * 8;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t xf = bx * 64, xl = (xf + 63 < dw - 1) ? xf + 63 : dw - 1;
  if (xf >= dw) return;
  const uint32_t x = xf + lane, xc = x < dw ? x : dw - 1;
  const uint32_t y0 = by * tile_rows, yl = (y0 + tile_rows - 1 < dh - 1) ? y0 + tile_rows - 1 : dh - 1;
  auto clampi = [](int32_t i, int32_t hi) { return (uint32_t)(i < 0 ? 0 : (i > hi ? hi : i)); };
  int32_t R0, R1;
  uint32_t first, last;
  if constexpr (LZ) {
    R0 = ltap_i0(y0, scy) - 2; R1 = ltap_i0(yl, scy) + 3;
    first = clampi(ltap_i0(xf, scx) - 2, (int32_t)sw - 1); last = clampi(ltap_i0(xl, scx) + 3, (int32_t)sw - 1);
  } else {
    R0 = (int32_t)make_tap<VPF_INTERP_LINEAR>(y0, scy, sh).i0; R1 = (int32_t)make_tap<VPF_INTERP_LINEAR>(yl, scy, sh).i1;
    first = make_tap<VPF_INTERP_LINEAR>(xf, scx, sw).i0; last = make_tap<VPF_INTERP_LINEAR>(xl, scx, sw).i1;
  }
  const uint32_t nrows = (uint32_t)(R1 - R0 + 1);
  const uint32_t base = (4 * CH * first) & ~15u, nq = (4 * CH * (last + 1) - base + 15) / 16;
  const uint32_t scol = threadIdx.x & ((1u << lshift) - 1u), srow0 = threadIdx.x >> lshift, srows = T >> lshift;
  for (uint32_t rb = 0; rb < nrows; rb += kTileStagePasses * srows) {  // usually one round; the units of a round are all in flight together
    u32x4 stage[kTileStagePasses];
#pragma unroll
    for (int k = 0; k < kTileStagePasses; k++) {
      const uint32_t r = rb + srow0 + k * srows;
      if (r < nrows && scol < nq) stage[k] = ldg<false, u32x4>(src + (size_t)clampi(R0 + (int32_t)r, (int32_t)sh - 1) * sp + base + 16 * scol);
    }
#pragma unroll
    for (int k = 0; k < kTileStagePasses; k++) {
      const uint32_t r = rb + srow0 + k * srows;
      if (r < nrows && scol < nq) RAW[r * rowq + scol] = stage[k];
    }
  }
  const uint32_t vt = LZ ? threadIdx.x - 64 : threadIdx.x;
  if (vt < tile_rows) {
    const uint32_t y = y0 + vt, yc = y < dh ? y : dh - 1;
    if constexpr (LZ) {
      const LTap t = make_ltap(yc, scy);
#pragma unroll
      for (int k = 0; k < 6; k++) WY[vt * 8 + k] = t.w[k];
      WY[vt * 8 + 6] = __int_as_float(t.i0 - 2 - R0);
    } else {
      const Tap t = make_tap<VPF_INTERP_LINEAR>(yc, scy, sh);
      WY[vt * 8] = t.f;
      WY[vt * 8 + 6] = __int_as_float((int32_t)t.i0 - R0);
      WY[vt * 8 + 7] = __int_as_float((int32_t)t.i1 - R0);
    }
  }
  if constexpr (LZ) {
    if (wv == 0) {
      const LTap tx = make_ltap(xc, scx);
#pragma unroll
      for (int k = 0; k < 6; k++) WX[k * 64 + lane] = tx.w[k];
      WX[6 * 64 + lane] = __int_as_float(tx.i0);
    }
  }
  __syncthreads();
  uint32_t xo[NT];  // float index of tap k's first channel inside the staged row
  float wx[NT];
  if constexpr (LZ) {
    const int32_t i0 = __float_as_int(WX[6 * 64 + lane]);
#pragma unroll
    for (int k = 0; k < 6; k++) { xo[k] = (clampi(i0 + k - 2, (int32_t)sw - 1) * 4 * CH - base) >> 2; wx[k] = WX[k * 64 + lane]; }
  } else {
    const Tap tx = make_tap<VPF_INTERP_LINEAR>(xc, scx, sw);
    xo[0] = (tx.i0 * 4 * CH - base) >> 2; xo[1] = (tx.i1 * 4 * CH - base) >> 2; wx[0] = tx.f; wx[1] = 0.f;
  }
  for (uint32_t r = wv; r < nrows; r += WPB) {
    const float* b = reinterpret_cast<const float*>(RAW + (size_t)r * rowq);
    float v[NT][CH];  // all taps requested before the first is used
#pragma unroll
    for (int k = 0; k < NT; k++)
#pragma unroll
      for (int c = 0; c < CH; c++) v[k][c] = b[xo[k] + c];
#pragma unroll
    for (int c = 0; c < CH; c++) {
      float ra;
      if constexpr (LZ) {
        ra = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) ra = __builtin_fmaf(wx[k], v[k][c], ra);
      } else {
        ra = __builtin_fmaf(wx[0], v[1][c] - v[0][c], v[0][c]);
      }
      H[(r * CH + c) * 64 + lane] = ra;
    }
  }
  __syncthreads();
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const uint32_t cg = lane & 15, rsub = lane >> 4, x0 = xf + 4 * cg;
  for (uint32_t yb = 0; yb < tile_rows; yb += 4 * WPB) {
    const uint32_t yy = yb + wv * 4 + rsub, y = y0 + yy;
    if (yy >= tile_rows || y >= dh || x0 >= dw) continue;
    const uint32_t r0 = (uint32_t)__float_as_int(WY[yy * 8 + 6]);
    f32x2 acc[CH][2];
    if constexpr (LZ) {
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c][0] = acc[c][1] = f32x2{0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 6; ky++) {
        const float wy = WY[yy * 8 + ky];
        const f32x2 wy2 = {wy, wy};
#pragma unroll
        for (int c = 0; c < CH; c++) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(&H[((r0 + ky) * CH + c) * 64 + 4 * cg]);
          acc[c][0] = __builtin_elementwise_fma(wy2, f32x2{hv[0], hv[1]}, acc[c][0]);
          acc[c][1] = __builtin_elementwise_fma(wy2, f32x2{hv[2], hv[3]}, acc[c][1]);
        }
      }
    } else {
      const uint32_t r1 = (uint32_t)__float_as_int(WY[yy * 8 + 7]);
      const float fy = WY[yy * 8];
      const f32x2 fy2 = {fy, fy};
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const f32x4 top = *reinterpret_cast<const f32x4*>(&H[(r0 * CH + c) * 64 + 4 * cg]);
        const f32x4 bot = *reinterpret_cast<const f32x4*>(&H[(r1 * CH + c) * 64 + 4 * cg]);
        const f32x2 t0 = {top[0], top[1]}, t1 = {top[2], top[3]}, b0 = {bot[0], bot[1]}, b1 = {bot[2], bot[3]};
        acc[c][0] = __builtin_elementwise_fma(fy2, b0 - t0, t0);
        acc[c][1] = __builtin_elementwise_fma(fy2, b1 - t1, t1);
      }
    }
    float* out = reinterpret_cast<float*>(dst + (size_t)y * dp) + (size_t)CH * x0;
    float o[4 * CH];  // pixel-major
#pragma unroll
    for (int c = 0; c < CH; c++) { o[c] = acc[c][0][0]; o[CH + c] = acc[c][0][1]; o[2 * CH + c] = acc[c][1][0]; o[3 * CH + c] = acc[c][1][1]; }
    if (P.vec_ok && x0 + 4 <= dw) {  // 16-B aligned rows: CH 16-B stores per lane
#pragma unroll
      for (int q = 0; q < CH; q++) stg<true, f32x4>(out + 4 * q, f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]});
    } else {
      const uint32_t nv = (dw - x0 < 4 ? dw - x0 : 4) * CH;
      for (uint32_t i = 0; i < nv; i++) out[i] = o[i];
    }
  }
}
// single plane, single frame: scalar arguments, source side first (kernarg preload: see VPF_ONE_SRC_PARAMS in vpf_internal.h)
template <int CH, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_resize_tile(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh,
                                                          uint8_t* __restrict__ dst, uint32_t dp, uint32_t dw, uint32_t dh,
                                                          float scx, float scy, uint32_t tile_rows, uint32_t nr_cap, uint32_t rowq, uint32_t lshift,
                                                          int vec_ok) {
  TileTask<CH, WPB>::run(src, sp, dst, dp, PlaneGeom{sw, sh, dw, dh, scx, scy, vec_ok, tile_rows, nr_cap, rowq, lshift}, blockIdx.x, blockIdx.y);
}
// ------------------------------------------------------------------------------------------
// Tiled, separable 8-bit LANCZOS-3 for what the matrix-core kernel (k_lanczos_mfma.hip) does not take: packed RGB beyond ~6 x and any plane
// beyond ~10 x across / ~6 x down (its tap windows and its four-tile ring end there), and ONE small frame per dispatch, where this kernel's
// short waves finish sooner than that kernel's latency chain (lanczos_single_prefers_tile, launch_resize_jobs).
// Same integer filter definition as that kernel, the gather kernel and the oracle: exact Q14 horizontal sums,
// Hr = (H + 128) >> 8, vertical taps on byte-wide partial products with clamped taps merged per source row.  A workgroup of WPB waves owns
// a tile of 64 columns x TY rows:
//   phase 0  the tile's whole source window goes to LDS in one sweep of dense 16-B loads (edge tiles: the pixels clamped taps fall on are
//            replicated into the rows' margins, so every lane's six taps are contiguous bytes in every tile); wave 0 evaluates the 64
//            column weight sets meanwhile, wave 1 the row sets (merged where the clamp folds taps onto one row, split into signed bytes);
//   phase 1  wave w takes source rows w, w + WPB, ...: aligned dword reads, v_alignbyte_b32, one v_perm_b32 per tap pair, two taps per
//            v_dot2_i32_i16; z = Hr - 8192 goes to LDS as the int16 pair (z, zh) with z = 256 zh + zl;
//   phase 2  a lane owns 4 consecutive columns of one destination row: per vertical tap and channel ONE v_dot2_i32_i16 of (z, zh)
//            against (qh, ql) — qh z + ql zh = (q z - ql zl) >> 8 exactly —, then (acc + 2^19 + 2^11) >> 12, clamped.
// (The gather kernel — a lane per destination pixel, six taps times six rows from global memory — remains for rows that are not 16-B
// aligned and tiles that do not fit LDS.)
// ------------------------------------------------------------------------------------------
template <int CH, int WPB>
struct LanczosTileTask {
  static constexpr int kThreads = 64 * WPB;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& P, uint32_t bx, uint32_t by);
};
template <int CH, int WPB>
VPF_DEV void LanczosTileTask<CH, WPB>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& P,
                                           uint32_t bx, uint32_t by) {
  // dynamic LDS: RAW[nr_cap][rowq x 16 B] source bytes | H[nr_cap][CH][64] (z | zh << 16) | WY[tile_rows][8] (six (qh | ql << 16), first H row) | WX[4][64]
  constexpr uint32_t T = 64 * WPB;
  const uint32_t sw = P.sw, sh = P.sh, dw = P.dw, dh = P.dh, tile_rows = P.a0, nr_cap = P.a1, rowq = P.a2, lshift = P.a3;
  const float scx = P.scx, scy = P.scy;
  u32x4* const RAW = dyn_strip;
  uint32_t* const H = reinterpret_cast<uint32_t*>(dyn_strip + (size_t)nr_cap * rowq);
  uint32_t* const WY = H + (size_t)nr_cap * CH * 64;
  uint32_t* const WX = WY + (size_t)tile_rows * 8;  // [4][64]: three Q14 weight pairs and the first tap index of every column
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t xf = bx * 64, xl = (xf + 63 < dw - 1) ? xf + 63 : dw - 1;
  if (xf >= dw) return;  // a narrower plane of a multi-plane launch (workgroup-uniform)
  const uint32_t x = xf + lane, xc = x < dw ? x : dw - 1;  // lanes past the right edge compute a duplicate, never stored
  const uint32_t y0 = by * tile_rows, yl = (y0 + tile_rows - 1 < dh - 1) ? y0 + tile_rows - 1 : dh - 1;
  auto clampi = [](int32_t i, int32_t hi) { return (uint32_t)(i < 0 ? 0 : (i > hi ? hi : i)); };
  const int32_t R0 = ltap_i0(y0, scy) - 2, R1 = ltap_i0(yl, scy) + 3;
  const int32_t fv = ltap_i0(xf, scx) - 2, lv = ltap_i0(xl, scx) + 3;  // the tile's VIRTUAL source columns: before 0 / after sw - 1 they are copies of the edge pixel
  const uint32_t first = clampi(fv, (int32_t)sw - 1), last = clampi(lv, (int32_t)sw - 1);
  const uint32_t nrows = (uint32_t)(R1 - R0 + 1);
  const uint32_t base = (CH * first) & ~15u, nq = (CH * (last + 1) - base + 15) / 16;
  const uint32_t scol = threadIdx.x & ((1u << lshift) - 1u), srow0 = threadIdx.x >> lshift, srows = T >> lshift;
  u32x4 stage[kTileStagePasses];
#pragma unroll
  for (int k = 0; k < kTileStagePasses; k++) {
    const uint32_t r = srow0 + k * srows;
    if (r < nrows && scol < nq) stage[k] = ldg<false, u32x4>(src + (size_t)clampi(R0 + (int32_t)r, (int32_t)sh - 1) * sp + base + 16 * scol);
  }
  const uint32_t vt = threadIdx.x - 64;  // lane of wave 1 (and later waves) that owns destination row y0 + vt
  if (vt < tile_rows) {
    const uint32_t y = y0 + vt, yc = y < dh ? y : dh - 1;
    const QTap t = quantize_ltap(make_ltap(yc, scy));
    int32_t q[6];
#pragma unroll
    for (int k = 0; k < 6; k++) q[k] = t.q[k];
#pragma unroll
    for (int k = 0; k < 5; k++)  // taps the clamp folds onto one source row act as ONE tap with the summed weight, carried by the last of them
      if (clampi(t.i0 + k - 2, (int32_t)sh - 1) == clampi(t.i0 + k - 1, (int32_t)sh - 1)) { q[k + 1] += q[k]; q[k] = 0; }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int32_t ql = ((q[k] + 128) & 0xff) - 128, qh = (q[k] - ql) >> 8;
      WY[vt * 8 + k] = pack_i16(qh, ql);
    }
    WY[vt * 8 + 6] = (uint32_t)(t.i0 - 2 - R0);
  }
  if (wv == 0) {
    const QTap tx = quantize_ltap(make_ltap(xc, scx));
#pragma unroll
    for (int k = 0; k < 3; k++) WX[k * 64 + lane] = pack_i16(tx.q[2 * k], tx.q[2 * k + 1]);
    WX[3 * 64 + lane] = (uint32_t)tx.i0;
  }
#pragma unroll
  for (int k = 0; k < kTileStagePasses; k++) {
    const uint32_t r = srow0 + k * srows;
    if (r < nrows && scol < nq) RAW[r * rowq + scol + 1] = stage[k];  // rows start one 16-B unit into their LDS row: room for up to 3 replicated pixels on the left
  }
  if (fv < 0 || lv > (int32_t)sw - 1) {  // workgroup-uniform: a tile on the left / right image edge
    __syncthreads();
    const int32_t nl = fv < 0 ? -fv * CH : 0, nr = lv > (int32_t)sw - 1 ? (lv - ((int32_t)sw - 1)) * CH : 0;  // bytes to add on each side (<= 3 px)
    for (uint32_t t = threadIdx.x; t < nrows * 16; t += T) {
      uint8_t* b = reinterpret_cast<uint8_t*>(RAW + (size_t)(t >> 4) * rowq) + 16;
      const int32_t i = (int32_t)(t & 15);
      if (i < nl) b[CH * fv + i] = b[i % CH];                                                    // base == 0 on the left edge
      if (i < nr) b[CH * sw - base + (uint32_t)i] = b[CH * (sw - 1) - base + (uint32_t)i % CH];
    }
  }
  __syncthreads();
  uint32_t qx[3];
  const int32_t i0 = (int32_t)WX[3 * 64 + lane];
  const uint32_t xo0 = (uint32_t)((i0 - 2) * CH - (int32_t)base + 16);  // the six taps are contiguous bytes from here
#pragma unroll
  for (int k = 0; k < 3; k++) qx[k] = WX[k * 64 + lane];
  // Phase 1, horizontal, two source rows per iteration (both rows' LDS reads are in flight before the first is used)
  {
    constexpr int NE = (6 * CH + 3) / 4;  // dwords of the lead-free run
    const uint32_t lead = xo0 & 3u, qoff = xo0 & ~3u;
    auto fetch = [&](uint32_t r, uint32_t* d) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(RAW + (size_t)r * rowq) + qoff);
#pragma unroll
      for (int i = 0; i <= NE; i++) d[i] = q[i];
    };
    auto hdots = [&](uint32_t r, const uint32_t* d) {
      uint32_t e[NE];
#pragma unroll
      for (int i = 0; i < NE; i++) e[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], lead);
#pragma unroll
      for (int c = 0; c < CH; c++) {
        uint32_t p0, p1, p2;  // (tap 0 | tap 1 << 16), (tap 2 | tap 3 << 16), (tap 4 | tap 5 << 16) of channel c
        if constexpr (CH == 3) {  // bytes c, 3 + c | 6 + c, 9 + c | 12 + c, 15 + c of the run
          p0 = __builtin_amdgcn_perm(e[1], e[0], 0x0c000c00u | ((3u + c) << 16) | (uint32_t)c);
          p1 = __builtin_amdgcn_perm(e[2], e[1], 0x0c000c00u | ((5u + c) << 16) | (2u + c));
          p2 = __builtin_amdgcn_perm(e[4], e[3], 0x0c000c00u | ((3u + c) << 16) | (uint32_t)c);
        } else if constexpr (CH == 2) {  // bytes c, 2 + c of dwords 0, 1, 2
          const uint32_t sel = 0x0c000c00u | ((2u + c) << 16) | (uint32_t)c;
          p0 = __builtin_amdgcn_perm(e[0], e[0], sel); p1 = __builtin_amdgcn_perm(e[1], e[1], sel); p2 = __builtin_amdgcn_perm(e[2], e[2], sel);
        } else {  // bytes 0, 1 | 2, 3 | 4, 5
          p0 = __builtin_amdgcn_perm(e[0], e[0], 0x0c010c00u); p1 = __builtin_amdgcn_perm(e[0], e[0], 0x0c030c02u); p2 = __builtin_amdgcn_perm(e[1], e[1], 0x0c010c00u);
        }
        const int32_t h = dot2(p2, qx[2], dot2(p1, qx[1], dot2z(p0, qx[0])));  // exact: |h| < 2^24
        const int32_t z = ((h + 128) >> 8) - 8192, zh = (z + 128) >> 8;
        H[(r * CH + c) * 64 + lane] = pack_i16(z, zh);
      }
    };
    for (uint32_t r = wv; r < nrows; r += 2 * WPB) {
      uint32_t da[NE + 1], db[NE + 1];
      const bool two = r + WPB < nrows;  // wave-uniform
      fetch(r, da);
      if (two) fetch(r + WPB, db);
      hdots(r, da);
      if (two) hdots(r + WPB, db);
    }
  }
  __syncthreads();
  // phase 2: a lane owns 4 consecutive columns of one destination row (a wave = 4 rows x 64 columns)
  const uint32_t cg = lane & 15, rsub = lane >> 4, x0 = xf + 4 * cg;
  for (uint32_t yb = 0; yb < tile_rows; yb += 4 * WPB) {
    const uint32_t yy = yb + wv * 4 + rsub, y = y0 + yy;
    if (yy >= tile_rows || y >= dh || x0 >= dw) continue;
    const uint32_t r0 = WY[yy * 8 + 6];
    int32_t acc[CH][4];
#pragma unroll
    for (int c = 0; c < CH; c++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[c][i] = (1 << 19) + (1 << 11);
#pragma unroll
    for (int ky = 0; ky < 6; ky++) {
      const uint32_t wq = WY[yy * 8 + ky];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const u32x4 hv = *reinterpret_cast<const u32x4*>(&H[((r0 + ky) * CH + c) * 64 + 4 * cg]);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[c][i] = dot2(hv[i], wq, acc[c][i]);  // qh z + ql zh
      }
    }
    // pixel-major bytes, four per dword: shift, clamp and pack in six instructions per dword (shift12_sat_pack4, k_resize_common.h)
    auto val = [&](int b) { return (uint32_t)acc[b % CH][b / CH]; };  // byte b of the lane's 4 CH output bytes
    uint32_t ow[CH];
#pragma unroll
    for (int q = 0; q < CH; q++) ow[q] = shift12_sat_pack4(val(4 * q), val(4 * q + 1), val(4 * q + 2), val(4 * q + 3));
    uint8_t* out = dst + (size_t)y * dp + (size_t)CH * x0;
    if (P.vec_ok && x0 + 4 <= dw) {
      if constexpr (CH == 3) {
        stg3<true>(out, ow[0], ow[1], ow[2]);
      } else if constexpr (CH == 2) {
        stg<true, u32x2>(out, u32x2{ow[0], ow[1]});
      } else {
        stg<true, uint32_t>(out, ow[0]);
      }
    } else {
      const uint32_t nv = (dw - x0 < 4 ? dw - x0 : 4) * CH;
      for (uint32_t i = 0; i < nv; i++) out[i] = (uint8_t)(ow[i >> 2] >> (8 * (i & 3)));
    }
  }
}
template <int CH, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_resize_lztile(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh,
                                                            uint8_t* __restrict__ dst, uint32_t dp, uint32_t dw, uint32_t dh,
                                                            float scx, float scy, uint32_t tile_rows, uint32_t nr_cap, uint32_t rowq, uint32_t lshift,
                                                            int vec_ok) {
  LanczosTileTask<CH, WPB>::run(src, sp, dst, dp, PlaneGeom{sw, sh, dw, dh, scx, scy, vec_ok, tile_rows, nr_cap, rowq, lshift}, blockIdx.x, blockIdx.y);
}
template <int CH> struct TileLz4 : LanczosTileTask<CH, 4> {};
template <int CH> struct TileLz8 : LanczosTileTask<CH, 8> {};

// the template-template forms k_planes_mp wants
template <int CH> struct TileBl4 : TileTask<CH, 4> {};
template <int CH> struct TileBl8 : TileTask<CH, 8> {};
template <int CH> struct RowPair1 : RowPairTask<CH, 1> {};
template <int CH> struct RowPair2 : RowPairTask<CH, 2> {};
template <int CH> struct RowPair3 : RowPairTask<CH, 3> {};
template <int CH> struct RowPair4 : RowPairTask<CH, 4> {};
template <int CH> struct RowBand2 : RowBandTask<CH, 2> {};
template <int CH> struct RowBand4 : RowBandTask<CH, 4> {};
template <int CH> struct RowBand8 : RowBandTask<CH, 8> {};
template <int CH> struct RowBand8n : RowBandTask<CH, 8, 1> {};    // narrow strips (<= 1 KiB): 16 slots
template <int CH> struct RowBand16n : RowBandTask<CH, 16, 1> {};
template <int CH> struct RowBand8w : RowBandTask<CH, 8, 1, 8> {};   // ... and 8 pixels per lane on 1-channel planes
template <int CH> struct RowBand16w : RowBandTask<CH, 16, 1, 8> {};
template <int CH> struct RowBand4wm : RowBandTask<CH, 4, 1, 8, true> {};  // the march form: 4-row bands, several per wave (plan_band)

// ------------------------------------------------------------------------------------------
// Exact 2x bilinear down-scale (4K -> 1080p ...): s = 2 d + 0.5 exactly, so every destination pixel is the fx = fy = 0.5
// bilerp of one 2 x 2 block: no tap arithmetic, no LDS, pure streaming.  Lane = 4 destination pixels = 8 source pixels x 2
// rows (CH x 8 contiguous bytes per row, loaded as 8-B units), one CH x 4-B store.  Bit-identical to the general kernels.
// Requires dw % 4 == 0, 8-B aligned source rows, 4-B (8-B for CH == 2) aligned destination rows.
// ------------------------------------------------------------------------------------------
template <int CH>
struct HalfTask {
  static constexpr int kThreads = 256;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by);
};
template <int CH>
__global__ __launch_bounds__(256) void k_resize_half(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp,
                                                     uint32_t dw, uint32_t dh) {
  HalfTask<CH>::run(src, sp, dst, dp, PlaneGeom{2 * dw, 2 * dh, dw, dh, 2.f, 2.f, 1, 0, 0, 0, 0}, blockIdx.x, blockIdx.y);
}
template <int CH>
VPF_DEV void HalfTask<CH>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx,
                               uint32_t by) {
  const uint32_t dw = G.dw, dh = G.dh;
  const uint32_t gx = bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
  const uint32_t x0 = gx * 4;
  if (x0 >= dw || y >= dh) return;
  uint32_t ra[2 * CH], rb[2 * CH];  // 8 source pixels of rows 2y and 2y + 1
  const uint8_t* pa = src + (size_t)(2 * y) * sp + (size_t)CH * 2 * x0;
  const uint8_t* pb = pa + sp;
#pragma unroll
  for (int j = 0; j < CH; j++) {
    const u32x2 a = ldg<true, u32x2>(pa + 8 * j), b = ldg<true, u32x2>(pb + 8 * j);
    ra[2 * j] = a[0]; ra[2 * j + 1] = a[1]; rb[2 * j] = b[0]; rb[2 * j + 1] = b[1];
  }
  float o[4 * CH];
#pragma unroll
  for (int e = 0; e < 4; e++)
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int i0 = CH * 2 * e + c, i1 = i0 + CH;  // byte index of the two horizontal taps inside the lane's run
      auto byte = [](const uint32_t* d, int i) { return (float)((d[i >> 2] >> (8 * (i & 3))) & 0xffu); };
      o[e * CH + c] = bilerp(byte(ra, i0), byte(ra, i1), byte(rb, i0), byte(rb, i1), 0.5f, 0.5f);
    }
  uint8_t* out = dst + (size_t)y * dp + (size_t)CH * x0;
  if constexpr (CH == 3) {
    { uint32_t d0, d1, d2; pack12_trunc(o, d0, d1, d2); stg3<true>(out, d0, d1, d2); }
  } else if constexpr (CH == 2) {
    stg<true, u32x2>(out, u32x2{pack4_trunc(o[0], o[1], o[2], o[3]), pack4_trunc(o[4], o[5], o[6], o[7])});
  } else {
    stg<true, uint32_t>(out, pack4_trunc(o[0], o[1], o[2], o[3]));
  }
}

// tiled separable launch (Lanczos always; bilinear when up-scaling, where the horizontal lerp is shared by several
// destination rows): needs 16-B aligned source rows and a 64-column span that fits the 2-KiB strip.  Returns false when
// it does not apply.
// Exact 2x down-scale of packed RGB / BGR in the r16 form: a wave turns two source rows x 1024 px into 512 destination
// pixels.  Every global access is a dense 1-KiB wave access (load_run48 on the way in, an LDS-transposed 1 KiB + 512 B on
// the way out; k_resize_half's 8-B lane loads reach ~0.6 of that rate), and the blend is integer: with fx = fy = 0.5 the
// bilinear blend of four 8-bit taps is exactly (p00 + p01 + p10 + p11 + 2) >> 2 (see k_convert_half), taken per channel
// from the de-interleaved dwords with two v_dot4_u32_u8 whose weights (64 on two bytes) pick a horizontal pair of each row.
// Requires sw % 32 == 0 and 16-B aligned source / destination rows.
struct Half3R16Task {
  static constexpr int kThreads = 256;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by);
};
__global__ __launch_bounds__(256) void k_resize_half3_r16(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp,
                                                          uint32_t sw, uint32_t chunks_x, uint32_t n_tasks) {
  Half3R16Task::run(src, sp, dst, dp, PlaneGeom{sw, 0, sw >> 1, 0, 2.f, 2.f, 1, chunks_x, n_tasks, 0, 0}, blockIdx.x, 0);
}
VPF_DEV void Half3R16Task::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx,
                               uint32_t /*by*/) {
  __shared__ u32x4 tile[4 * 192];  // 3 KiB per wave
  const uint32_t sw = G.sw, chunks_x = G.a0, n_tasks = G.a1;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wt = bx * 4 + wv;
  if (wt >= n_tasks) return;
  const uint32_t y = wt / chunks_x, chunk = wt - y * chunks_x;
  u32x4* t = tile + wv * 192;
  uint32_t top[12], bot[12];
  u32x4 qt[3], qb[3];  // both rows' loads are in flight before the first transpose
  run48_fetch(src + (size_t)(2 * y) * sp, chunk * 3072, 3 * sw, lane, qt);
  run48_fetch(src + (size_t)(2 * y + 1) * sp, chunk * 3072, 3 * sw, lane, qb);
  run48_transpose(t, lane, qt, top);
  wave_sync();  // every lane has read its top bytes before the tile is rewritten
  run48_transpose(t, lane, qb, bot);
  uint32_t v[3][8];  // channel value of destination pixel i in byte 1
#pragma unroll
  for (int g = 0; g < 4; g++) {
    uint32_t tc[3], bc[3];
    deint4(top[3 * g], top[3 * g + 1], top[3 * g + 2], tc[0], tc[1], tc[2]);
    deint4(bot[3 * g], bot[3 * g + 1], bot[3 * g + 2], bc[0], bc[1], bc[2]);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      v[c][2 * g] = __builtin_amdgcn_udot4(tc[c], 0x00004040u, __builtin_amdgcn_udot4(bc[c], 0x00004040u, 128u, false), false);
      v[c][2 * g + 1] = __builtin_amdgcn_udot4(tc[c], 0x40400000u, __builtin_amdgcn_udot4(bc[c], 0x40400000u, 128u, false), false);
    }
  }
  auto gather4 = [](uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {  // byte 1 of four registers -> one dword
    return __builtin_amdgcn_perm(__builtin_amdgcn_perm(v3, v2, 0x0c0c0501u), __builtin_amdgcn_perm(v1, v0, 0x0c0c0501u), 0x05040100u);
  };
  uint32_t* tw = reinterpret_cast<uint32_t*>(t);
  wave_sync();
#pragma unroll
  for (int g = 0; g < 2; g++) {  // 4 px -> 3 dwords, twice
    const int q = 4 * g;
    tw[lane * 6 + 3 * g] = gather4(v[0][q], v[1][q], v[2][q], v[0][q + 1]);
    tw[lane * 6 + 3 * g + 1] = gather4(v[1][q + 1], v[2][q + 1], v[0][q + 2], v[1][q + 2]);
    tw[lane * 6 + 3 * g + 2] = gather4(v[2][q + 2], v[0][q + 3], v[1][q + 3], v[2][q + 3]);
  }
  wave_sync();
  uint8_t* row = dst + (size_t)y * dp;
  const uint32_t row_bytes = 3 * (sw >> 1);
#pragma unroll
  for (int k = 0; k < 2; k++) {  // the wave's 1536 B leave as one dense 1-KiB store and one 512-B store
    const uint32_t idx = k * 64 + lane, off = chunk * 1536 + idx * 16;
    if (idx < 96 && off < row_bytes) stg<true, u32x4>(row + off, t[idx]);
  }
}

// Shape of a tiled launch: destination rows per tile (ty) and waves per workgroup (wpb).  A taller tile computes fewer horizontal dots per
// destination pixel ((ty - 1) scy + taps + 2 source rows for ty destination rows) but there are fewer of them to fill the chip with.
// Rule read off rocprofv3 kernel durations over (ty, wpb) for six size pairs (profiles/r02_tile_shape_sweep.txt): 8 waves per workgroup
// always (each wave walks fewer source rows one after the other), and the TALLEST tile that fits LDS and still leaves >= ~900 workgroups
// (about one full round of the chip at 4 workgroups per CU) — e.g. 1080p -> 720p: ty 16 (900 workgroups, 7.5 us; ty 32: 7.7, ty 8: 9.7),
// 720p -> 1080p: ty 32 (7.2 us; ty 64: 9.4), 1080p -> 4K: ty 64 (15.5 us; ty 32: 18.9), batches: the tallest that fits.
// VPF_TUNE_RESIZE_TILE overrides the choice for such sweeps (ty | wpb << 8) — every shape writes the same pixels.
struct TileShape { bool ok; uint32_t ty, nr, lds, rowq, lshift; int wpb; };
static TileShape plan_tile(bool lz, int np, const int* ch, const uint32_t* dw, const uint32_t* dh, const float* scxs, const float* scys, uint32_t frames,
                           int elem = 1 /* bytes per sample: 1, or 4 for float surfaces */) {
  TileShape best{false, 0, 0, 0, 0, 0, 4};
  int ch_max = 0;
  uint32_t rowq = 0;  // 16-B units a tile row can span (+ alignment slack), the widest plane decides
  float scy = 0.f;
  for (int p = 0; p < np; p++) {
    ch_max = ch[p] > ch_max ? ch[p] : ch_max;
    const uint32_t q = vpf_bound_tile_rowq(scxs[p], lz ? 6 : 2, ch[p], elem);  // (8-bit Lanczos: + pad unit + right margin; vpf_plan_bounds.h)
    rowq = q > rowq ? q : rowq;
    scy = scys[p] > scy ? scys[p] : scy;
  }
  if (rowq > kLzStripQ * (uint32_t)elem || scy > 48.0f) return best;
  uint32_t lshift = 0;
  while ((1u << lshift) < rowq) lshift++;
  const int forced = tuning(VPF_TUNE_RESIZE_TILE);
  uint32_t best_wgs = 0;
  for (int wpb = forced ? 4 : 8; wpb <= 8; wpb += 4)
    for (uint32_t ty = forced ? 4 : 8; ty <= 64; ty += 4) {
      if (forced && ((uint32_t)(forced & 0xff) != ty || (forced >> 8) != wpb)) continue;
      const uint32_t nr = vpf_bound_tile_rows(ty, scy, lz ? 6 : 2);
      const uint32_t lds = nr * rowq * 16 + nr * ch_max * 64 * 4 + ty * 8 * 4 + (lz ? (elem == 4 ? 7 : 4) * 64 * 4 : 0);  // RAW | H | WY | WX (Lanczos)
      if (lds > 64u * 1024u) continue;
      const uint32_t srows = (64u * wpb) >> lshift;  // source rows staged per pass
      if (!srows || (elem == 1 && (nr + srows - 1) / srows > (uint32_t)kTileStagePasses)) continue;  // (the float task stages in rounds)
      double wgs = 0;
      for (int p = 0; p < np; p++) wgs += (double)((dw[p] + 63) / 64) * ((dh[p] + ty - 1) / ty);
      wgs *= frames;
      const uint32_t w = wgs > 4e9 ? 0xffffffffu : (uint32_t)wgs;
      // candidates come in order of growing ty: take a taller tile while it still leaves >= 900 workgroups; below that, the most workgroups win
      if (!best.ok || w >= 900u || w > best_wgs) { best = TileShape{true, ty, nr, lds, rowq, lshift, wpb}; best_wgs = w; }
    }
  return best;
}

// tiled separable launch of ONE plane of ONE frame (bilinear up-scales, where the horizontal lerp is shared by several destination
// rows): needs 16-B aligned source rows and a 64-column span that fits the 2-KiB strip.  Returns false when it does not apply.
static bool launch_resize_tile(hipStream_t st, int ch, uint32_t sw, uint32_t sh, const uint8_t* src, uint32_t sp,
                               uint32_t dw, uint32_t dh, uint8_t* dst, uint32_t dp, float scx, float scy) {
  if (tuning(VPF_TUNE_NV12_RGB_VARIANT) == 9 || (((uintptr_t)src | sp) & 15)) return false;
  const TileShape t = plan_tile(false, 1, &ch, &dw, &dh, &scx, &scy, 1);
  if (!t.ok) return false;
  dim3 tgrid((dw + 63) / 64, (dh + t.ty - 1) / t.ty);
  const int vec_ok = ((((uintptr_t)dst | dp) & 3) == 0) && (ch != 2 || (((uintptr_t)dst | dp) & 7) == 0);
#define VPF_TILE3(C, W) VPF_LAUNCH((k_resize_tile<C, W>), tgrid, dim3(64 * W), t.lds, st, src, sp, sw, sh, dst, dp, dw, dh, scx, scy, t.ty, t.nr, t.rowq, t.lshift, vec_ok)
#define VPF_TILE(C) do { if (t.wpb == 8) VPF_TILE3(C, 8); else VPF_TILE3(C, 4); } while (0)
  if (ch == 1) VPF_TILE(1); else if (ch == 2) VPF_TILE(2); else VPF_TILE(3);
#undef VPF_TILE
#undef VPF_TILE3
  return true;
}

// ONE plane of ONE frame per dispatch (what an unmodified PySurfaceResizer.Execute loop issues): the matrix-core Lanczos kernel's waves are
// few and long — operand tables, a ring to fill, two passes — and a lone launch of them is one latency chain of 5-6 us whatever the picture;
// the tile kernel's waves are short and many, so a small single frame leaves it sooner although its throughput is 2.5x lower.  The tile
// kernel's time grows with the DESTINATION (taps per output sample), the matrix-core kernel's with the source: over 63 measured shapes
// (tools/lanczos_single_sweep.py, profiles/r04_lanczos_single_sweep.txt: RGB 1080p -> 720p 6.9 against 8.9 us, Y 4.7 against 6.9) the rule
// "source bytes / 2 + 3 x destination bytes below 25.5 MB (3 channels) / 14 MB (1 channel)" picks the faster kernel but for 1 us summed over
// all of them.  Same bytes either way (both kernels are the integer definition of the filter).  VPF_TUNE_RESIZE_MFMA | 0x40000, or a forced
// launch shape: always the matrix-core kernel.
static bool lanczos_single_prefers_tile(int ch, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh) {
  if (tuning(VPF_TUNE_RESIZE_MFMA) & 0xfffff) return false;
  const uint64_t src_b = (uint64_t)sw * sh * (uint32_t)ch, dst_b = (uint64_t)dw * dh * (uint32_t)ch;
  return src_b + 6u * dst_b <= (ch == 1 ? 28000000ull : ch == 2 ? 40000000ull : 51000000ull);
}

hipError_t launch_resize(hipStream_t st, int ch, int interp, uint32_t sw, uint32_t sh, const uint8_t* src,
                         uint32_t sp, uint32_t dw, uint32_t dh, uint8_t* dst, uint32_t dp) {
  const float scx = (float)sw / (float)dw, scy = (float)sh / (float)dh;
  const int vec_ok = ((((uintptr_t)dst | dp) & 3) == 0) && (ch != 2 || (((uintptr_t)dst | dp) & 7) == 0);
  // Odd integer scale factors on both axes: s = (d + 0.5) k - 0.5 = k d + (k - 1) / 2 exactly, every destination centre is
  // a source pixel centre, so the bilinear fractions are 0 and the Lanczos weights are (0, 0, 1, 0, 0, 0): both filters return
  // that source pixel unchanged, which is also what NEAREST picks (floor((d + 0.5) k) = k d + (k - 1) / 2).  Identical
  // bytes from the cheapest kernel (the tiled kernels were tried with per-tap zero-weight skipping instead: 10 us for
  // Lanczos 4K -> 720p but 30-45 % slower on every other ratio).  Bilinear keeps its row-pair LDS kernel, whose own
  // zero-weight shortcuts make it as fast there (kernel durations 5.1 vs 5.5 us at 4K -> 720p).
  if (interp == VPF_INTERP_LANCZOS3 && sw % dw == 0 && sh % dh == 0 && ((sw / dw) & 1) && ((sh / dh) & 1) && sw < (1u << 22) && sh < (1u << 22) &&
      tuning(VPF_TUNE_NV12_RGB_VARIANT) != 40 && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 9)
    interp = VPF_INTERP_NEAREST;
  // exact 2x bilinear: the quad-structured streaming kernel
  if (interp == VPF_INTERP_LINEAR && sw == 2 * dw && sh == 2 * dh && dw % 4 == 0 && !(((uintptr_t)src | sp) & 7) &&
      !(((uintptr_t)dst | dp) & (ch == 2 ? 7 : 3)) && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 40 && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 9) {
    if (ch == 3 && sw % 32 == 0 && !(((uintptr_t)src | sp | (uintptr_t)dst | dp) & 15)) {
      const uint32_t chunks = (sw + 1023) / 1024, tasks = chunks * dh;
      VPF_LAUNCH(k_resize_half3_r16, dim3((tasks + 3) / 4), dim3(256), 0, st, src, sp, dst, dp, sw, chunks, tasks);
      return hipGetLastError();
    }
    dim3 hgrid((dw / 4 + 63) / 64, (dh + 3) / 4);
    if (ch == 1) VPF_LAUNCH((k_resize_half<1>), hgrid, dim3(256), 0, st, src, sp, dst, dp, dw, dh);
    else if (ch == 2) VPF_LAUNCH((k_resize_half<2>), hgrid, dim3(256), 0, st, src, sp, dst, dp, dw, dh);
    else VPF_LAUNCH((k_resize_half<3>), hgrid, dim3(256), 0, st, src, sp, dst, dp, dw, dh);
    return hipGetLastError();
  }
  if (interp == VPF_INTERP_LANCZOS3) {
    // the matrix-core kernel (k_lanczos_mfma.hip) wherever its windows fit; the tiled separable form for strong down-scales (beyond those
    // windows) — and, tried FIRST, for one small plane per dispatch (lanczos_single_prefers_tile below); the gather form for what is left
    auto mfma = [&]() -> bool {
      BatchArgsL a;  // (host side only: one frame is launched with the small table)
      std::memset(&a.f[0], 0, sizeof(a.f[0]));
      a.f[0].s[0] = src; a.f[0].sp[0] = sp; a.f[0].d[0] = dst; a.f[0].dp[0] = dp;
      const ResizeJob j{ch, 0, sw, sh, dw, dh};
      return launch_lanczos_mfma(st, 1, &j, 1, a);
    };
    auto tile = [&]() -> bool {
      if (tuning(VPF_TUNE_NV12_RGB_VARIANT) == 9 || tuning(VPF_TUNE_NV12_RGB_VARIANT) == 40 || (((uintptr_t)src | sp) & 15)) return false;
      const TileShape t = plan_tile(true, 1, &ch, &dw, &dh, &scx, &scy, 1);
      if (!t.ok) return false;
      const dim3 tgrid((dw + 63) / 64, (dh + t.ty - 1) / t.ty);
#define VPF_LZT3(C, W) VPF_LAUNCH((k_resize_lztile<C, W>), tgrid, dim3(64 * W), t.lds, st, src, sp, sw, sh, dst, dp, dw, dh, scx, scy, t.ty, t.nr, t.rowq, t.lshift, vec_ok)
#define VPF_LZT(C) do { if (t.wpb == 8) VPF_LZT3(C, 8); else VPF_LZT3(C, 4); } while (0)
      if (ch == 1) VPF_LZT(1); else if (ch == 2) VPF_LZT(2); else VPF_LZT(3);
#undef VPF_LZT
#undef VPF_LZT3
      return true;
    };
    if (lanczos_single_prefers_tile(ch, sw, sh, dw, dh) ? (tile() || mfma()) : (mfma() || tile())) return hipGetLastError();
    dim3 lgrid((dw + 63) / 64, (dh + 3) / 4);
    if (ch == 1) VPF_LAUNCH((k_resize_lanczos<1>), lgrid, dim3(256), 0, st, src, sp, sw, sh, dst, dp, dw, dh, scx, scy);
    else if (ch == 2) VPF_LAUNCH((k_resize_lanczos<2>), lgrid, dim3(256), 0, st, src, sp, sw, sh, dst, dp, dw, dh, scx, scy);
    else VPF_LAUNCH((k_resize_lanczos<3>), lgrid, dim3(256), 0, st, src, sp, sw, sh, dst, dp, dw, dh, scx, scy);
    return hipGetLastError();
  }
  dim3 grid(((dw + 3) / 4 + 63) / 64, (dh + 3) / 4);
  // up-scaling: several destination rows sit between the same two source rows -> tiled kernel (horizontal lerp once per
  // source row).  Kernel durations (rocprofv3), tiled vs row-pair: 1080p->4K 16.6 vs 20.6 us, 720p->1080p 8.2 vs 9.0;
  // for mild down-scales the row-pair kernel is as fast or faster (1080p->720p 5.9 vs 5.8, 4K->1440p 16.2 vs 12.4,
  // 4K->3000x1688 21.2 vs 14.9), so the tiled kernel is used below 1.0 only
  if (interp == VPF_INTERP_LINEAR && (scy < 1.0f || tuning(VPF_TUNE_NV12_RGB_VARIANT) == 43) && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 40 &&
      launch_resize_tile(st, ch, sw, sh, src, sp, dw, dh, dst, dp, scx, scy))
    return hipGetLastError();
  const uint32_t rowb = (interp == VPF_INTERP_LINEAR) ? lds_strip_bytes(ch, sw, dw, src, sp, kResizeRowBytes) : 0;
  if (rowb) {
    // (capping residency at 4 / 2 / 1 workgroups per CU to stagger the waves was measured: 8.5 / 8.7 / 11.3 us vs 8.4)
    const uint32_t it = (rowb + 1023) / 1024, lds = 4 * 2 * rowb + 16;  // + 16: a tap window may start in a strip's last dword
#define VPF_RL(C, I) VPF_LAUNCH((k_resize_lds<C, I>), grid, dim3(256), lds, st, src, sp, sw, sh, dst, dp, dw, dh, scx, scy, vec_ok, rowb / 16)
#define VPF_RLI(C) do { if (it == 1) VPF_RL(C, 1); else if (it == 2) VPF_RL(C, 2); else if (it == 3) VPF_RL(C, 3); else VPF_RL(C, 4); } while (0)
    if (ch == 1) VPF_RLI(1); else if (ch == 2) VPF_RLI(2); else VPF_RLI(3);
#undef VPF_RLI
#undef VPF_RL
    return hipGetLastError();
  }
#define VPF_GO(C, I) VPF_LAUNCH((k_resize<C, I>), grid, dim3(256), 0, st, src, sp, sw, sh, dst, dp, dw, dh, scx, scy, vec_ok)
  if (interp == VPF_INTERP_LINEAR) {
    if (ch == 1) VPF_GO(1, VPF_INTERP_LINEAR); else if (ch == 2) VPF_GO(2, VPF_INTERP_LINEAR); else VPF_GO(3, VPF_INTERP_LINEAR);
  } else {
    if (ch == 1) VPF_GO(1, VPF_INTERP_NEAREST); else if (ch == 2) VPF_GO(2, VPF_INTERP_NEAREST); else VPF_GO(3, VPF_INTERP_NEAREST);
  }
#undef VPF_GO
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// 32-bit float surfaces (RGB_32F: CH = 3 interleaved, RGB_32F_PLANAR: CH = 1 per plane; reference
// NppResizeSurfacePacked32F3C_Impl / NppResizeSurface32FPlanar_Impl, Tasks.cpp:1334-1445): the same taps and operation
// order as the 8-bit kernels on float samples; the result is neither rounded nor clamped.  Lane = one destination pixel
// (CH x 4 B per lane: a wave stores 256-768 contiguous bytes).
// ------------------------------------------------------------------------------------------
template <int CH>
struct FloatPx { float c[CH]; };  // one pixel of an RGB_32F (CH = 3) / planar float (CH = 1) surface: 4-B aligned, loaded as one 4 * CH-byte access
template <int CH, int INTERP>
struct FloatGatherTask {
  static constexpr int kThreads = 256;
  static VPF_DEV void run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G, uint32_t bx, uint32_t by);
};
template <int CH, int INTERP>
__global__ __launch_bounds__(256) void k_resize_f32(const uint8_t* __restrict__ src, uint32_t sp, uint32_t sw, uint32_t sh,
                                                    uint8_t* __restrict__ dst, uint32_t dp, uint32_t dw, uint32_t dh,
                                                    float scx, float scy) {
  FloatGatherTask<CH, INTERP>::run(src, sp, dst, dp, PlaneGeom{sw, sh, dw, dh, scx, scy, 0, 0, 0, 0, 0}, blockIdx.x, blockIdx.y);
}
template <int CH, int INTERP>
VPF_DEV void FloatGatherTask<CH, INTERP>::run(const uint8_t* __restrict__ src, uint32_t sp, uint8_t* __restrict__ dst, uint32_t dp, const PlaneGeom& G,
                                              uint32_t bx, uint32_t by) {
  const uint32_t sw = G.sw, sh = G.sh, dw = G.dw, dh = G.dh;
  const float scx = G.scx, scy = G.scy;
  const uint32_t x = bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  float* o = reinterpret_cast<float*>(dst + (size_t)y * dp) + (size_t)CH * x;
  if constexpr (INTERP == VPF_INTERP_LANCZOS3) {
    const LTap tx = make_ltap(x, scx), ty = make_ltap(y, scy);
    uint32_t xi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int32_t i = tx.i0 + k - 2;
      xi[k] = (uint32_t)(i < 0 ? 0 : (i > (int32_t)sw - 1 ? (int32_t)sw - 1 : i)) * CH;
    }
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 6; ky++) {
      const int32_t j = ty.i0 + ky - 2;
      const float* r = reinterpret_cast<const float*>(src + (size_t)(j < 0 ? 0 : (j > (int32_t)sh - 1 ? (int32_t)sh - 1 : j)) * sp);
      // a row's taps are requested together and only then accumulated: left alone, the compiler sinks each load to its
      // fma and the lane waits out one memory round trip per tap (36 * CH of them)
      // and a pixel's CH floats travel as one 4 * CH-byte load (the kernel is bound by the number of load instructions)
      float v[6][CH];
#pragma unroll
      for (int kx = 0; kx < 6; kx++) {
        const FloatPx<CH> px = *reinterpret_cast<const FloatPx<CH>*>(r + xi[kx]);
#pragma unroll
        for (int c = 0; c < CH; c++) v[kx][c] = px.c[c];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CH; c++) {
        float ra = 0.f;
#pragma unroll
        for (int kx = 0; kx < 6; kx++) ra = __builtin_fmaf(tx.w[kx], v[kx][c], ra);
        acc[c] = __builtin_fmaf(ty.w[ky], ra, acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < CH; c++) o[c] = acc[c];
  } else {
    const Tap tx = make_tap<INTERP>(x, scx, sw), ty = make_tap<INTERP>(y, scy, sh);
    const float* r0 = reinterpret_cast<const float*>(src + (size_t)ty.i0 * sp);
    const float* r1 = reinterpret_cast<const float*>(src + (size_t)ty.i1 * sp);
    FloatPx<CH> q00 = *reinterpret_cast<const FloatPx<CH>*>(r0 + CH * tx.i0), q01 = {}, q10 = {}, q11 = {};
    if constexpr (INTERP != VPF_INTERP_NEAREST) {
      q01 = *reinterpret_cast<const FloatPx<CH>*>(r0 + CH * tx.i1);
      q10 = *reinterpret_cast<const FloatPx<CH>*>(r1 + CH * tx.i0); q11 = *reinterpret_cast<const FloatPx<CH>*>(r1 + CH * tx.i1);
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
      if constexpr (INTERP == VPF_INTERP_NEAREST) {
        o[c] = q00.c[c];
      } else {
        const float p00 = q00.c[c], p01 = q01.c[c], p10 = q10.c[c], p11 = q11.c[c];
        const float top = __builtin_fmaf(tx.f, p01 - p00, p00), bot = __builtin_fmaf(tx.f, p11 - p10, p10);
        o[c] = __builtin_fmaf(ty.f, bot - top, top);
      }
    }
  }
}

hipError_t launch_resize_f32(hipStream_t st, int ch, int interp, uint32_t sw, uint32_t sh, const uint8_t* src, uint32_t sp,
                             uint32_t dw, uint32_t dh, uint8_t* dst, uint32_t dp) {
  const float scx = (float)sw / (float)dw, scy = (float)sh / (float)dh;
  dim3 grid((dw + 63) / 64, (dh + 3) / 4);
  if (interp != VPF_INTERP_NEAREST && sw % dw == 0 && sh % dh == 0 && ((sw / dw) & 1) && ((sh / dh) & 1) && sw < (1u << 22) && sh < (1u << 22) &&
      tuning(VPF_TUNE_NV12_RGB_VARIANT) != 40 && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 9)
    interp = VPF_INTERP_NEAREST;  // odd integer factors: every filter returns the centre sample (see launch_resize)
  // Lanczos: the tiled separable form (16-B aligned source rows; 1080p -> 720p 22.7 -> 15.7 us, 720p -> 1080p 42.7 -> 19.5 us); the
  // gather form below otherwise.  Bilinear stays on the gather form: tiled, a 720p -> 1080p up-scale took 17.0 us against 10.1 (four
  // times the bytes of an 8-bit surface go through LDS for little reuse); the bilinear branch of TileTaskF32 is reachable with tuning 43.
  if ((interp == VPF_INTERP_LANCZOS3 || (interp == VPF_INTERP_LINEAR && scy < 1.0f && tuning(VPF_TUNE_NV12_RGB_VARIANT) == 43)) && tuning(VPF_TUNE_NV12_RGB_VARIANT) != 9 &&
      tuning(VPF_TUNE_NV12_RGB_VARIANT) != 40 && !(((uintptr_t)src | sp) & 15)) {
    const bool lz = interp == VPF_INTERP_LANCZOS3;
    const TileShape t = plan_tile(lz, 1, &ch, &dw, &dh, &scx, &scy, 1, 4);
    if (t.ok) {
      const PlaneGeom g{sw, sh, dw, dh, scx, scy, (int)((((uintptr_t)dst | dp) & 15) == 0), t.ty, t.nr, t.rowq, t.lshift};
      const dim3 tgrid((dw + 63) / 64, (dh + t.ty - 1) / t.ty);
      BatchArgs a;
      std::memset(&a, 0, sizeof(a));
      a.f[0].s[0] = src; a.f[0].sp[0] = sp; a.f[0].d[0] = dst; a.f[0].dp[0] = dp;
#define VPF_TF(C, L, W) VPF_LAUNCH((k_plane_batch<TileTaskF32<C, L, W>>), tgrid, dim3(64 * W), t.lds, st, a, 0, g)
#define VPF_TFW(C, L) do { if (t.wpb == 8) VPF_TF(C, L, 8); else VPF_TF(C, L, 4); } while (0)
      if (ch == 3) { if (lz) VPF_TFW(3, true); else VPF_TFW(3, false); } else { if (lz) VPF_TFW(1, true); else VPF_TFW(1, false); }
#undef VPF_TFW
#undef VPF_TF
      return hipGetLastError();
    }
  }
#define VPF_F32(C, I) VPF_LAUNCH((k_resize_f32<C, I>), grid, dim3(256), 0, st, src, sp, sw, sh, dst, dp, dw, dh, scx, scy)
  if (ch == 3) {
    if (interp == VPF_INTERP_LANCZOS3) VPF_F32(3, VPF_INTERP_LANCZOS3); else if (interp == VPF_INTERP_LINEAR) VPF_F32(3, VPF_INTERP_LINEAR); else VPF_F32(3, VPF_INTERP_NEAREST);
  } else {
    if (interp == VPF_INTERP_LANCZOS3) VPF_F32(1, VPF_INTERP_LANCZOS3); else if (interp == VPF_INTERP_LINEAR) VPF_F32(1, VPF_INTERP_LINEAR); else VPF_F32(1, VPF_INTERP_NEAREST);
  }
#undef VPF_F32
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// vpf_resize_batch: every plane of a frame and up to 32 same-shape frames in as few dispatches as possible.  A 720p plane is 2-3 us
// of work, the same order as a kernel boundary, and an NV12 / YUV420 frame is two or three such planes: one launch per plane per
// frame leaves the chip idle most of the time.  The plane geometry decides the kernel family exactly as in launch_resize; when every
// plane of the format lands in the same (tiled or row-pair) family they all go into ONE launch (k_planes_mp), otherwise each plane
// gets one launch over all frames (k_plane_batch).  Same task bodies as the single-frame kernels -> identical pixels.
// ------------------------------------------------------------------------------------------
static bool planes_aligned(const BatchArgsL& a, uint32_t n, int k, uintptr_t src_mask, uintptr_t dst_mask) {
  for (uint32_t i = 0; i < n; i++)
    if ((((uintptr_t)a.f[i].s[k] | a.f[i].sp[k]) & src_mask) || (((uintptr_t)a.f[i].d[k] | a.f[i].dp[k]) & dst_mask)) return false;
  return true;
}
// (the selection log / roctx mark carries __PRETTY_FUNCTION__: it names the task the generic entry was instantiated with)
// (grid.z = the frames of the launch: up to 32 take the small frame table, more the large one — two instantiations of the same kernel)
template <class Task>
static void launch_plane_batch(hipStream_t st, dim3 grid, uint32_t lds, const BatchArgsL& a, int k, const PlaneGeom& g) {
  if (log_level() >= 2 || trace_on()) note_kernel(__PRETTY_FUNCTION__);
  (void)hipGetLastError();
  if (grid.z <= (uint32_t)kSmallBatch) hipLaunchKernelGGL((k_plane_batch<Task, BatchArgs>), grid, dim3(Task::kThreads), lds, st, small_batch(a, grid.z), k, g);
  else hipLaunchKernelGGL((k_plane_batch<Task, BatchArgsL>), grid, dim3(Task::kThreads), lds, st, a, k, g);
}
template <template <int> class TaskCH>
static void launch_planes_mp(hipStream_t st, dim3 grid, uint32_t lds, const BatchArgsL& a, const PlaneTable& t) {
  if (log_level() >= 2 || trace_on()) note_kernel(__PRETTY_FUNCTION__);
  (void)hipGetLastError();
  if (grid.z <= (uint32_t)kSmallBatch) hipLaunchKernelGGL((k_planes_mp<TaskCH, BatchArgs>), grid, dim3(TaskCH<3>::kThreads), lds, st, small_batch(a, grid.z), t);
  else hipLaunchKernelGGL((k_planes_mp<TaskCH, BatchArgsL>), grid, dim3(TaskCH<3>::kThreads), lds, st, a, t);
}
#ifdef VPF_LAB_FORMS
// ---- the persistent form of the same launch (k_resize_common.h: k_planes_mp_persist; vpf_persist.h): the resident set of workgroups pulls
// wave items (frame, plane, wave row, column chunk) from the stream's work counters.  -> false when it does not apply (captured stream, no
// counter slot free, too few items per wave to be worth it): the caller launches the plain grid.
__device__ uint32_t g_persist_ctr[kPersistSlots * 16];  // two sets of eight per slot (vpf_persist.h)
static PersistSlotTable& persist_slots() { static PersistSlotTable t; return t; }
static bool persist_stream_idle(int dev, const void* stream) {
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != dev) { (void)hipGetLastError(); return false; }  // (another device's stream: not asked from here)
  const hipError_t e = hipStreamQuery((hipStream_t)stream);
  if (e != hipSuccess) (void)hipGetLastError();
  return e != hipErrorNotReady;  // drained — or no longer a stream
}
static uint32_t* persist_ctr_base(int dev) {
  static std::atomic<uint32_t*> base[64];
  uint32_t* p = base[dev].load(std::memory_order_acquire);
  if (!p) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_persist_ctr)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    p = static_cast<uint32_t*>(q);
    base[dev].store(p, std::memory_order_release);
  }
  return p;
}
template <template <int> class TaskCH>
static uint32_t persist_resident_groups(int dev, uint32_t lds) {  // workgroups of this kernel the whole chip holds at `lds` bytes each
  struct Seen { int dev; uint32_t lds, groups; };
  static thread_local Seen seen[4];
  static thread_local unsigned next = 0;
  for (const Seen& e : seen) if (e.groups && e.dev == dev && e.lds == lds) return e.groups;
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_planes_mp_persist<TaskCH, BatchArgs>), (int)TaskCH<3>::kThreads, lds) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1) { (void)hipGetLastError(); return 0; }
  const uint32_t groups = (uint32_t)per_cu * (uint32_t)cus;
  seen[next++ & 3] = Seen{dev, lds, groups};
  return groups;
}
template <template <int> class TaskCH>
static bool launch_planes_mp_persist(hipStream_t st, uint32_t lds, const BatchArgsL& a, const PlaneTable& t, uint32_t n, const uint32_t* nbx, const uint32_t* nbands, uint32_t chunk, uint32_t hops) {
  PersistArgs P{};
  uint32_t per_frame = 0;
  if (!chunk) return false;
  for (uint32_t p = 0; p < t.np; p++) { P.p0[p] = per_frame; P.nbx[p] = nbx[p]; P.nbands[p] = nbands[p]; per_frame += nbx[p] * ((nbands[p] + chunk - 1) / chunk); }
  const uint64_t total = (uint64_t)per_frame * n;
  if (!per_frame || total >= (1u << 22)) return false;
  if (st == hipStreamPerThread) return false;  // one handle, a different stream in every thread: not a key for a slot of counters
  int dev = 0;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) { (void)hipGetLastError(); return false; }
  const uint32_t resident = persist_resident_groups<TaskCH>(dev, lds), wpg = TaskCH<3>::kThreads / 64u;
  if (!resident) return false;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (cs != hipStreamCaptureStatusNone) return false;
  uint32_t* const ctr = persist_ctr_base(dev);
  int set = 0;
  const int slot = ctr ? persist_slots().take(dev, (const void*)st, persist_stream_idle, &set) : -1;
  if (slot < 0) return false;
  P.ctr = ctr + 16 * slot + 8 * set;
  P.ctr_other = ctr + 16 * slot + 8 * (set ^ 1);
  P.per_frame = per_frame;
  P.chunk = chunk;
  P.hops = hops > 8 ? 8 : hops;
  persist_shares((uint32_t)total, P.lo);
  if (log_level() >= 2 || trace_on()) note_kernel(__PRETTY_FUNCTION__);
  (void)hipGetLastError();
  const uint32_t groups = (uint32_t)std::max<uint64_t>(8, std::min<uint64_t>(resident, (total + wpg - 1) / wpg));  // >= 8: every XCD's share has a wave that draws from it
  if (n <= (uint32_t)kSmallBatch) hipLaunchKernelGGL((k_planes_mp_persist<TaskCH, BatchArgs>), dim3(groups), dim3(TaskCH<3>::kThreads), lds, st, small_batch(a, n), t, P);
  else hipLaunchKernelGGL((k_planes_mp_persist<TaskCH, BatchArgsL>), dim3(groups), dim3(TaskCH<3>::kThreads), lds, st, a, t, P);
  return true;
}
#endif  // VPF_LAB_FORMS
template <template <int, int> class T2, int I>
static void launch_gather_ch(hipStream_t st, dim3 grid, const BatchArgsL& a, const ResizeJob& j, const PlaneGeom& g) {
  if (j.ch == 1) launch_plane_batch<T2<1, I>>(st, grid, 0, a, j.k, g);
  else if (j.ch == 2) launch_plane_batch<T2<2, I>>(st, grid, 0, a, j.k, g);
  else launch_plane_batch<T2<3, I>>(st, grid, 0, a, j.k, g);
}

// Destination rows per wave for a row-pair launch (RowBandTask) and the strips per wave its LDS is sized for: the largest of 16 / 8 / 4 / 2
// whose bands fit the strip slots (8; 16 when a strip is at most 1 KiB = one staging pass, the 1- and 2-channel planes and the up-scales)
// and 64 KB of LDS (`rb` bytes per strip, at most two 1-KiB staging passes) while the launch keeps at least kBandMinGroups workgroups (a
// single small frame stays at one row per wave: it needs the parallelism more than the shared work).  Vertical factors above 2 skip
// source rows (a contiguous band would read rows nobody blends) and odd integer factors on both axes move bytes (RowPairTask's
// centre-sample shortcut): both keep one row per wave.  VPF_TUNE_RESIZE_BAND forces a value where it applies.
constexpr uint32_t kBandMinGroups = 2048;
struct BandShape { int rows; uint32_t slots; bool narrow; uint32_t rb; };  // rb: strip bytes of the chosen form (16-row bands of packed RGB: four bytes per pixel)
static uint32_t band_slots(int r, float scy) { return vpf_bound_band_slots(r, scy); }  // (vpf_plan_bounds.h: checked on the CPU against the tap arithmetic)
// The strips a launch ALLOCATES: the rows its bands really touch (vpf_band_rows_exact: a walk over the bands with make_tap's arithmetic),
// never more than the closed-form bound that chose the band height.  1080p -> 720p with 4-row bands: 6 strips instead of 8, five
// workgroups per CU instead of four.  The last shapes are remembered per thread (a launch per frame would walk dh / r taps every time).
static uint32_t band_slots_exact(int r, int njobs, const ResizeJob* jobs, uint32_t bound) {
  struct Seen { int r; uint32_t sh, dh, rows; };
  static thread_local Seen seen[8];
  static thread_local unsigned next = 0;
  uint32_t most = 1;
  for (int p = 0; p < njobs; p++) {
    const ResizeJob& j = jobs[p];
    uint32_t rows = 0;
    for (const Seen& e : seen)
      if (e.r == r && e.sh == j.sh && e.dh == j.dh && e.rows) rows = e.rows;
    if (!rows) {
      rows = vpf_band_rows_exact(r, j.sh, j.dh, (float)j.sh / (float)j.dh);
      seen[next++ & 7] = Seen{r, j.sh, j.dh, rows};
    }
    most = rows > most ? rows : most;
  }
  return most < bound ? most : bound;
}
// pixels per lane for the 1-channel planes of a band launch: 8 (512 columns per wave) when such chunks fill every 1-channel row to >= 80 %
// (1280 px: 3 chunks, 83 %; the 640-px chroma planes of a 720p YUV420 frame: 2 chunks, 62 % -> 4)
static int band_p1(int njobs, const ResizeJob* jobs) {
  bool any = false;
  const bool force8 = (tuning(VPF_TUNE_RESIZE_BAND) >> 17) & 1;  // | 0x20000: 8 pixels per lane on 1-channel planes however well 512-column chunks fill them (measurement knob)
  for (int p = 0; p < njobs; p++) {
    if (jobs[p].ch != 1) continue;
    any = true;
    if (!force8 && (double)jobs[p].dw < 0.8 * 512.0 * ((jobs[p].dw + 511) / 512)) return 4;
  }
  return any ? 8 : 4;
}
// strip bytes of the widest plane for a band launch (RowBandTask: 64 * band_px(ch, p1) columns per wave); 0 when some plane has no LDS path
static uint32_t band_strip_bytes(int njobs, const ResizeJob* jobs, uint32_t n, const BatchArgsL& a, int p1) {
  uint32_t rbmax = 0;
  for (int p = 0; p < njobs; p++) {
    const ResizeJob& j = jobs[p];
    bool al = true;
    for (uint32_t i = 0; i < n; i++) al = al && !(((uintptr_t)a.f[i].s[j.k] | a.f[i].sp[j.k]) & 15);
    const uint32_t rb = al ? lds_strip_bytes(j.ch, j.sw, j.dw, a.f[0].s[j.k], a.f[0].sp[j.k], kResizeRowBytes, 64u * band_px(j.ch, p1)) : 0;
    if (!rb) return 0;
    rbmax = rb > rbmax ? rb : rbmax;
  }
  return rbmax;
}
static BandShape band_rows(int njobs, const ResizeJob* jobs, uint32_t rb, uint32_t n, int p1 = 4) {
  const int forced = tuning(VPF_TUNE_RESIZE_BAND) & 0xff;  // (bits 8..: bands per wave of the march form, plan_band)
  if (forced == 1 || rb == 0 || rb > 2048) return {1, 0, false, rb};
  const uint32_t rb_packed = rb;
  float scy = 0.f;
  for (int p = 0; p < njobs; p++) {
    const ResizeJob& j = jobs[p];
    if (j.sw % j.dw == 0 && j.sh % j.dh == 0 && ((j.sw / j.dw) & 1) && ((j.sh / j.dh) & 1)) return {1, 0, false, rb};
    const float s = (float)j.sh / (float)j.dh;
    scy = s > scy ? s : scy;
  }
  if (scy > 2.0f) return {1, 0, false, rb};
  for (int r = 16; r >= 2; r >>= 1) {
    if (forced && forced != r) continue;
    rb = rb_packed;
    if (r == 16)  // RowBandTask<3, 16, 1>::kPx4: its strips hold packed RGB four bytes per pixel
      for (int p = 0; p < njobs; p++)
        if (jobs[p].ch == 3) {
          const uint32_t rb4 = vpf_bound_strip_bytes_px4(jobs[p].sw, jobs[p].dw, 1024u, 256u);
          rb = !rb4 ? 4096u : rb4 > rb ? rb4 : rb;
        }
    const bool narrow = r >= 8 && rb <= 1024;  // the IT = 1 instantiations (8 and 16 rows)
    if (r == 16 && !narrow) continue;
    const uint32_t slots = band_slots(r, scy);
    if (slots > (uint32_t)(narrow ? 2 * kBandSlots : kBandSlots) || 4u * slots * rb + 16u > 64u * 1024u) continue;
    uint64_t groups = 0;
    for (int p = 0; p < njobs; p++) groups += (uint64_t)((jobs[p].dw + 64u * band_px(jobs[p].ch, p1) - 1) / (64u * band_px(jobs[p].ch, p1))) * ((jobs[p].dh + 4 * r - 1) / (4 * r)) * n;
    if (forced || groups >= kBandMinGroups) return {r, band_slots_exact(r, njobs, jobs, slots), narrow, rb};
  }
  return {1, 0, false, rb_packed};
}
// the whole decision: 8 pixels per lane on the 1-channel planes only with the 16-slot (narrow-strip) instantiations that exist for it.
// Where that form would run 8-row bands on a DOWN-scale (every plane's vertical factor in [1, 2]) the launch takes the MARCH form instead
// (RowBand4wm): bands of 4 rows, nb of them per wave one below the other — the next band's source rows are requested while this band is
// blended, the walk's lerps and the column taps carry over — with half the LDS per wave (four waves per SIMD instead of three) and the
// wave's fixed part paid once per 4 nb rows.  nb = 2 or 3 while the launch keeps kBandMinGroups workgroups.  Measured on the 1- / 2-channel
// planes of Y / NV12 (profiles/r04_bilinear_march.txt: Y 1080p -> 720p 0.85 -> 0.79 us, NV12 1.14 -> 1.06); 3-channel planes and up-scales
// lose with it and keep their forms.  VPF_TUNE_RESIZE_BAND = 4 | nb << 8 forces it where it applies.
struct BandPlan { int rows; uint32_t slots; bool narrow; int p1; uint32_t rb; uint32_t nb; };
static BandPlan plan_band(int njobs, const ResizeJob* jobs, uint32_t n, const BatchArgsL& a) {
  const int p1 = band_p1(njobs, jobs), knob = tuning(VPF_TUNE_RESIZE_BAND) & 0xffff, forced_nb = (knob & 0xff) == 4 ? knob >> 8 : 0;  // (other row counts: the field is the persistent launch's chunk)
  if (p1 == 8) {
    const uint32_t rb = band_strip_bytes(njobs, jobs, n, a, 8);
    const BandShape bs = band_rows(njobs, jobs, rb, n, 8);
    bool down = rb != 0 && rb <= 1024;
    float scy = 1.f;
    for (int p = 0; p < njobs && down; p++) {
      const float s = (float)jobs[p].sh / (float)jobs[p].dh;
      down = s >= 1.0f && s <= 2.0f;
      scy = s > scy ? s : scy;
    }
    if (down && (forced_nb ? bs.rows == 4 : ((knob & 0xff) == 0 && bs.rows == 8 && bs.narrow))) {
      uint64_t groups = 0;
      for (int p = 0; p < njobs; p++) groups += (uint64_t)((jobs[p].dw + 64u * band_px(jobs[p].ch, 8) - 1) / (64u * band_px(jobs[p].ch, 8))) * ((jobs[p].dh + 15) / 16) * n;
      const uint32_t nb = forced_nb ? (uint32_t)forced_nb : (uint32_t)std::min<uint64_t>(3, groups / kBandMinGroups);
      const uint32_t slots = band_slots_exact(4, njobs, jobs, band_slots(4, scy));
      if (nb >= (forced_nb ? 1u : 2u) && slots <= 9u && 4u * slots * rb + 16u <= 64u * 1024u) return {4, slots, true, 8, rb, nb};
    }
    if (bs.rows >= 8 && bs.narrow) return {bs.rows, bs.slots, true, 8, bs.rb, 0};
  }
  const uint32_t rb = band_strip_bytes(njobs, jobs, n, a, 4);
  const BandShape bs = band_rows(njobs, jobs, rb, n, 4);
  return {bs.rows, bs.slots, bs.narrow, 4, bs.rb, 0};
}

hipError_t launch_resize_jobs(hipStream_t st, bool f32, int interp, int njobs, const ResizeJob* jobs, uint32_t n, const BatchArgsL& a) {
  enum Fam { FAM_GATHER, FAM_LZ_GATHER, FAM_LZ_MFMA, FAM_LZ_TILE, FAM_HALF, FAM_HALF3, FAM_TILE, FAM_ROWPAIR };
  bool lz_tile_ok[3] = {false, false, false};  // a Lanczos plane the tiled kernel may take when the matrix-core kernel does not
  const int tune = tuning(VPF_TUNE_NV12_RGB_VARIANT);
  if (njobs < 1 || njobs > 3 || !n || n > (uint32_t)kMaxBatch) return hipErrorInvalidValue;
  Fam fam[3];
  int eff[3];
  PlaneGeom g[3];
  uint32_t rowb[3] = {0, 0, 0};
  // bilinear up-scales: the tiled kernel for a single frame, the row-band kernel (8 or 4 destination rows per wave, each source row's
  // horizontal lerp evaluated once per band, no barriers) when the launch is large enough for it — 1080p -> 4K batched 11.9 -> 6.9 us / frame
  bool band_up = !f32 && interp == VPF_INTERP_LINEAR && tune != 43 && tune != 9;
  if (band_up) band_up = plan_band(njobs, jobs, n, a).rows >= 4;
  for (int p = 0; p < njobs; p++) {
    const ResizeJob& j = jobs[p];
    const float scx = (float)j.sw / (float)j.dw, scy = (float)j.sh / (float)j.dh;
    const bool odd_int = j.sw % j.dw == 0 && j.sh % j.dh == 0 && ((j.sw / j.dw) & 1) && ((j.sh / j.dh) & 1) && j.sw < (1u << 22) && j.sh < (1u << 22) &&
                         tune != 40 && tune != 9;  // every filter returns the centre sample (see launch_resize)
    eff[p] = interp;
    if (f32) {
      if (interp != VPF_INTERP_NEAREST && odd_int) eff[p] = VPF_INTERP_NEAREST;
      g[p] = PlaneGeom{j.sw, j.sh, j.dw, j.dh, scx, scy, 0, 0, 0, 0, 0};
      fam[p] = FAM_GATHER;
      continue;
    }
    if (interp == VPF_INTERP_LANCZOS3 && odd_int) eff[p] = VPF_INTERP_NEAREST;
    const int vec_ok = planes_aligned(a, n, j.k, 0, j.ch == 2 ? 7 : 3) ? 1 : 0;
    g[p] = PlaneGeom{j.sw, j.sh, j.dw, j.dh, scx, scy, vec_ok, 0, 0, 0, 0};
    const bool src16 = tune != 9 && planes_aligned(a, n, j.k, 15, 0);
    if (eff[p] == VPF_INTERP_LINEAR && j.sw == 2 * j.dw && j.sh == 2 * j.dh && j.dw % 4 == 0 && tune != 40 && tune != 9 &&
        planes_aligned(a, n, j.k, 7, j.ch == 2 ? 7 : 3)) {
      if (j.ch == 3 && j.sw % 32 == 0 && planes_aligned(a, n, j.k, 15, 15)) {
        const uint32_t chunks = (j.sw + 1023) / 1024;
        g[p].a0 = chunks; g[p].a1 = chunks * j.dh;
        fam[p] = FAM_HALF3;
      } else {
        fam[p] = FAM_HALF;
      }
    } else if (eff[p] == VPF_INTERP_LANCZOS3) {
      fam[p] = src16 ? FAM_LZ_MFMA : FAM_LZ_GATHER;
      lz_tile_ok[p] = src16 && tune != 40;
    } else if (eff[p] == VPF_INTERP_LINEAR && (scy < 1.0f || tune == 43) && tune != 40 && src16 && !band_up) {
      fam[p] = FAM_TILE;
    } else if (eff[p] == VPF_INTERP_LINEAR && src16 && (rowb[p] = lds_strip_bytes(j.ch, j.sw, j.dw, a.f[0].s[j.k], a.f[0].sp[j.k], kResizeRowBytes)) != 0) {
      fam[p] = FAM_ROWPAIR;
    } else {
      fam[p] = FAM_GATHER;
    }
  }
  // ---- every plane tiled: one launch for the whole format
  bool all_tile = !f32, all_rowpair = !f32;
  for (int p = 0; p < njobs; p++) {
    all_tile = all_tile && fam[p] == FAM_TILE && eff[p] == eff[0];
    all_rowpair = all_rowpair && fam[p] == FAM_ROWPAIR;
  }
  {  // every plane Lanczos: one matrix-core launch for the whole format (k_lanczos_mfma.hip); planes it cannot take fall back to the gather form
    bool all_mfma = !f32;
    for (int p = 0; p < njobs; p++) all_mfma = all_mfma && fam[p] == FAM_LZ_MFMA;
    // the tiled separable kernel for every plane in ONE launch; false when it does not apply
    auto tile_all_planes = [&]() -> bool {
      int ch[3]; uint32_t dw[3], dh[3]; float sx[3], sy[3];
      for (int p = 0; p < njobs; p++) { ch[p] = jobs[p].ch; dw[p] = jobs[p].dw; dh[p] = jobs[p].dh; sx[p] = g[p].scx; sy[p] = g[p].scy; }
      const TileShape tl = plan_tile(true, njobs, ch, dw, dh, sx, sy, n);
      if (!tl.ok) return false;
      PlaneTable t{};
      t.np = (uint32_t)njobs;
      uint32_t gx = 0, gy = 0;
      for (int p = 0; p < njobs; p++) {
        t.g[p] = g[p]; t.k[p] = (uint32_t)jobs[p].k; t.ch[p] = (uint32_t)jobs[p].ch; t.by0[p] = gy;
        t.g[p].a0 = tl.ty; t.g[p].a1 = tl.nr; t.g[p].a2 = tl.rowq; t.g[p].a3 = tl.lshift;
        gx = std::max(gx, (jobs[p].dw + 63) / 64);
        gy += (jobs[p].dh + tl.ty - 1) / tl.ty;
      }
      if (tl.wpb == 8) launch_planes_mp<TileLz8>(st, dim3(gx, gy, n), tl.lds, a, t);
      else launch_planes_mp<TileLz4>(st, dim3(gx, gy, n), tl.lds, a, t);
      return true;
    };
    // ONE frame of a multi-plane format per dispatch with a small destination (all planes together <= 1 MB: up to ~1100 x 640): a lone launch
    // of the matrix-core kernel is a latency chain of 8-9 us whatever the picture (lanczos_single_prefers_tile above has the single-plane
    // rule), the tile kernel — whose time follows the destination — leaves such a frame after 6-7 (profiles/r04_lanczos_single_multiplane.txt:
    // YUV420 1080p -> 224 x 224 5.9 against 7.9 us, NV12 1080p -> 416 x 416 6.8 against 9.0; 1080p -> 720p stays on the matrix cores, 7.5
    // against 8.3).  Only where ONE tile launch takes every plane (the chroma plane of NV12 at 8 x falls out of its windows).
    if (all_mfma && n == 1 && njobs > 1 && !(tuning(VPF_TUNE_RESIZE_MFMA) & 0xfffff)) {
      uint64_t dst_b = 0;
      bool tile_ok = true;
      for (int p = 0; p < njobs; p++) { dst_b += (uint64_t)jobs[p].dw * jobs[p].dh * (uint32_t)jobs[p].ch; tile_ok = tile_ok && lz_tile_ok[p]; }
      if (tile_ok && dst_b <= 1000000ull && tile_all_planes()) return hipGetLastError();
    }
    if (all_mfma && launch_lanczos_mfma(st, njobs, jobs, n, a)) return hipGetLastError();
    for (int p = 0; p < njobs; p++)
      if (fam[p] == FAM_LZ_MFMA && !launch_lanczos_mfma(st, 1, &jobs[p], n, a)) fam[p] = FAM_LZ_GATHER;
    // what the matrix-core kernel does not take (the strongest down-scales): the tiled separable kernel — one launch for the whole format when
    // every plane is in that position, else plane by plane — and the gather kernel for what is left
    bool all_lzt = !f32;
    for (int p = 0; p < njobs; p++) all_lzt = all_lzt && fam[p] == FAM_LZ_GATHER && lz_tile_ok[p];
    if (all_lzt && tile_all_planes()) return hipGetLastError();
    for (int p = 0; p < njobs; p++)
      if (fam[p] == FAM_LZ_GATHER && lz_tile_ok[p]) {
        const float sx = g[p].scx, sy = g[p].scy;
        const TileShape t1 = plan_tile(true, 1, &jobs[p].ch, &jobs[p].dw, &jobs[p].dh, &sx, &sy, n);
        if (!t1.ok) continue;
        fam[p] = FAM_LZ_TILE;
        g[p].a0 = t1.ty; g[p].a1 = t1.nr; g[p].a2 = t1.rowq; g[p].a3 = t1.lshift;
        rowb[p] = t1.lds | ((uint32_t)t1.wpb << 24);  // carried to the launch below
      }
  }
  TileShape ts{false, 0, 0, 0, 0, 0, 4};
  if (all_tile) {
    int ch[3]; uint32_t dw[3], dh[3]; float sx[3], sy[3];
    for (int p = 0; p < njobs; p++) { ch[p] = jobs[p].ch; dw[p] = jobs[p].dw; dh[p] = jobs[p].dh; sx[p] = g[p].scx; sy[p] = g[p].scy; }
    ts = plan_tile(false, njobs, ch, dw, dh, sx, sy, n);
    if (!ts.ok) {  // the span does not fit a strip: every plane falls back to its gather form
      all_tile = false;
      for (int p = 0; p < njobs; p++) fam[p] = FAM_GATHER;
    }
  } else {
    for (int p = 0; p < njobs; p++)
      if (fam[p] == FAM_TILE) {  // mixed families: this plane is tiled on its own
        const float sx = g[p].scx, sy = g[p].scy;
        const TileShape t1 = plan_tile(false, 1, &jobs[p].ch, &jobs[p].dw, &jobs[p].dh, &sx, &sy, n);
        if (!t1.ok) { fam[p] = FAM_GATHER; continue; }
        g[p].a0 = t1.ty; g[p].a1 = t1.nr; g[p].a2 = t1.rowq; g[p].a3 = t1.lshift;
        rowb[p] = t1.lds | ((uint32_t)t1.wpb << 24);  // carried to the launch below
      }
  }
  if (all_tile || all_rowpair) {
    PlaneTable t{};
    t.np = (uint32_t)njobs;
    uint32_t gx = 0, gy = 0, it = 1, rb = 0;
    for (int p = 0; p < njobs && all_rowpair; p++) rb = rowb[p] > rb ? rowb[p] : rb;
    const BandPlan bs = all_rowpair ? plan_band(njobs, jobs, n, a) : BandPlan{1, 0, false, 4, 0, 0};
    const int band = bs.rows;
    if (band > 1) rb = bs.rb;
    const uint32_t band_nb = bs.nb ? bs.nb : 1u;  // bands per wave (the march form)
    for (int p = 0; p < njobs; p++) {
      t.g[p] = g[p]; t.k[p] = (uint32_t)jobs[p].k; t.ch[p] = (uint32_t)jobs[p].ch; t.by0[p] = gy;
      if (all_tile) {
        t.g[p].a0 = ts.ty; t.g[p].a1 = ts.nr; t.g[p].a2 = ts.rowq; t.g[p].a3 = ts.lshift;
        const uint32_t bx = (jobs[p].dw + 63) / 64;
        gx = bx > gx ? bx : gx;
        gy += (jobs[p].dh + ts.ty - 1) / ts.ty;
      } else {
        const uint32_t wcols = band > 1 ? 64u * band_px(jobs[p].ch, bs.p1) : 256u, bx = (jobs[p].dw + wcols - 1) / wcols;
        gx = bx > gx ? bx : gx;
        gy += (jobs[p].dh + 4 * band * band_nb - 1) / (4 * band * band_nb);
      }
    }
    const dim3 grid(gx, gy, n);
    if (all_tile) {
      if (ts.wpb == 8) launch_planes_mp<TileBl8>(st, grid, ts.lds, a, t);
      else launch_planes_mp<TileBl4>(st, grid, ts.lds, a, t);
    } else {
      for (int p = 0; p < njobs; p++) { t.g[p].a0 = rb / 16; t.g[p].a1 = bs.slots; t.g[p].a2 = bs.nb; }  // one strip size for the launch (the widest plane's)
      it = (rb + 1023) / 1024;
      const uint32_t lds = band > 1 ? 4 * bs.slots * rb + 16 : 4 * 2 * rb + 16;
#ifdef VPF_LAB_FORMS
      // the persistent form for the band kernels (lab builds, VPF_TUNE_RESIZE_BAND | 0x10000): chunks of bands instead of a grid
      const int pknob = (tuning(VPF_TUNE_RESIZE_BAND) >> 16) & 1;
      if (band > 1 && pknob == 1) {
        // chunks of `chunk` bands (knob bits 8..15; default 2), one counter per XCD; | 0x40000: a wave whose counter has run dry visits the other seven
        uint32_t nbx[3] = {0, 0, 0}, nbands[3] = {0, 0, 0};
        for (int p = 0; p < njobs; p++) {
          const uint32_t wcols = 64u * band_px(jobs[p].ch, bs.p1);
          nbx[p] = (jobs[p].dw + wcols - 1) / wcols; nbands[p] = (jobs[p].dh + band - 1) / band;
        }
        const uint32_t chunk = ((tuning(VPF_TUNE_RESIZE_BAND) >> 8) & 0xff) ? (uint32_t)((tuning(VPF_TUNE_RESIZE_BAND) >> 8) & 0xff) : 2u;
        const uint32_t hops = ((tuning(VPF_TUNE_RESIZE_BAND) >> 19) & 1) ? 0u : ((tuning(VPF_TUNE_RESIZE_BAND) >> 18) & 1) ? 8u : 1u;  // | 0x80000 (measurement): no counters, a wave strides through its XCD's share
        bool done = false;
        if (band == 4 && bs.p1 == 8) done = launch_planes_mp_persist<RowBand4wm>(st, lds, a, t, n, nbx, nbands, chunk, hops);
        else if (band == 16 && bs.p1 == 8) done = launch_planes_mp_persist<RowBand16w>(st, lds, a, t, n, nbx, nbands, chunk, hops);
        else if (band == 8 && bs.p1 == 8) done = launch_planes_mp_persist<RowBand8w>(st, lds, a, t, n, nbx, nbands, chunk, hops);
        else if (band == 16) done = launch_planes_mp_persist<RowBand16n>(st, lds, a, t, n, nbx, nbands, chunk, hops);
        else if (band == 8 && bs.narrow) done = launch_planes_mp_persist<RowBand8n>(st, lds, a, t, n, nbx, nbands, chunk, hops);
        else if (band == 8) done = launch_planes_mp_persist<RowBand8>(st, lds, a, t, n, nbx, nbands, chunk, hops);
        else if (band == 4) done = launch_planes_mp_persist<RowBand4>(st, lds, a, t, n, nbx, nbands, chunk, hops);
        else if (band == 2) done = launch_planes_mp_persist<RowBand2>(st, lds, a, t, n, nbx, nbands, chunk, hops);
        if (done) return hipGetLastError();
      }
#endif  // VPF_LAB_FORMS
      if (band == 4 && bs.p1 == 8) launch_planes_mp<RowBand4wm>(st, grid, lds, a, t);
      else       if (band == 16 && bs.p1 == 8) launch_planes_mp<RowBand16w>(st, grid, lds, a, t);
      else if (band == 8 && bs.p1 == 8) launch_planes_mp<RowBand8w>(st, grid, lds, a, t);
      else if (band == 16) launch_planes_mp<RowBand16n>(st, grid, lds, a, t);
      else if (band == 8 && bs.narrow) launch_planes_mp<RowBand8n>(st, grid, lds, a, t);
      else if (band == 8) launch_planes_mp<RowBand8>(st, grid, lds, a, t);
      else if (band == 4) launch_planes_mp<RowBand4>(st, grid, lds, a, t);
      else if (band == 2) launch_planes_mp<RowBand2>(st, grid, lds, a, t);
      else if (it == 1) launch_planes_mp<RowPair1>(st, grid, lds, a, t);
      else if (it == 2) launch_planes_mp<RowPair2>(st, grid, lds, a, t);
      else if (it == 3) launch_planes_mp<RowPair3>(st, grid, lds, a, t);
      else launch_planes_mp<RowPair4>(st, grid, lds, a, t);
    }
    return hipGetLastError();
  }
  // ---- otherwise: one launch per plane over all frames
  for (int p = 0; p < njobs; p++) {
    const ResizeJob& j = jobs[p];
    if (fam[p] == FAM_LZ_MFMA) {  // launched above
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      continue;
    }
    const dim3 grid1((j.dw + 63) / 64, (j.dh + 3) / 4, n), grid4(((j.dw + 3) / 4 + 63) / 64, (j.dh + 3) / 4, n);
    if (f32 && (eff[p] == VPF_INTERP_LANCZOS3 || (eff[p] == VPF_INTERP_LINEAR && g[p].scy < 1.0f && tune == 43)) && tune != 9 && tune != 40 && planes_aligned(a, n, j.k, 15, 0)) {
      const bool lz = eff[p] == VPF_INTERP_LANCZOS3;
      const float sx = g[p].scx, sy = g[p].scy;
      const TileShape t = plan_tile(lz, 1, &j.ch, &j.dw, &j.dh, &sx, &sy, n, 4);
      if (t.ok) {
        g[p].vec_ok = planes_aligned(a, n, j.k, 0, 15) ? 1 : 0;
        g[p].a0 = t.ty; g[p].a1 = t.nr; g[p].a2 = t.rowq; g[p].a3 = t.lshift;
        const dim3 tgrid((j.dw + 63) / 64, (j.dh + t.ty - 1) / t.ty, n);
#define VPF_TFB(C, L) do { if (t.wpb == 8) launch_plane_batch<TileTaskF32<C, L, 8>>(st, tgrid, t.lds, a, j.k, g[p]); else launch_plane_batch<TileTaskF32<C, L, 4>>(st, tgrid, t.lds, a, j.k, g[p]); } while (0)
        if (j.ch == 3) { if (lz) VPF_TFB(3, true); else VPF_TFB(3, false); } else { if (lz) VPF_TFB(1, true); else VPF_TFB(1, false); }
#undef VPF_TFB
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        continue;
      }
    }
    if (f32) {
#define VPF_F32B(C) do { if (eff[p] == VPF_INTERP_LANCZOS3) launch_plane_batch<FloatGatherTask<C, VPF_INTERP_LANCZOS3>>(st, grid1, 0, a, j.k, g[p]); \
                         else if (eff[p] == VPF_INTERP_LINEAR) launch_plane_batch<FloatGatherTask<C, VPF_INTERP_LINEAR>>(st, grid1, 0, a, j.k, g[p]); \
                         else launch_plane_batch<FloatGatherTask<C, VPF_INTERP_NEAREST>>(st, grid1, 0, a, j.k, g[p]); } while (0)
      if (j.ch == 3) VPF_F32B(3); else VPF_F32B(1);
#undef VPF_F32B
    } else if (fam[p] == FAM_HALF3) {
      launch_plane_batch<Half3R16Task>(st, dim3((g[p].a1 + 3) / 4, 1, n), 0, a, j.k, g[p]);
    } else if (fam[p] == FAM_HALF) {
      const dim3 hgrid((j.dw / 4 + 63) / 64, (j.dh + 3) / 4, n);
      if (j.ch == 1) launch_plane_batch<HalfTask<1>>(st, hgrid, 0, a, j.k, g[p]);
      else if (j.ch == 2) launch_plane_batch<HalfTask<2>>(st, hgrid, 0, a, j.k, g[p]);
      else launch_plane_batch<HalfTask<3>>(st, hgrid, 0, a, j.k, g[p]);
    } else if (fam[p] == FAM_TILE) {
      const uint32_t lds = rowb[p] & 0xffffffu, wpb = rowb[p] >> 24;
      const dim3 tgrid((j.dw + 63) / 64, (j.dh + g[p].a0 - 1) / g[p].a0, n);
#define VPF_TILEB(C) do { if (wpb == 8) launch_plane_batch<TileTask<C, 8>>(st, tgrid, lds, a, j.k, g[p]); \
                          else launch_plane_batch<TileTask<C, 4>>(st, tgrid, lds, a, j.k, g[p]); } while (0)
      if (j.ch == 1) VPF_TILEB(1); else if (j.ch == 2) VPF_TILEB(2); else VPF_TILEB(3);
#undef VPF_TILEB
    } else if (fam[p] == FAM_LZ_TILE) {
      const uint32_t lds = rowb[p] & 0xffffffu, wpb = rowb[p] >> 24;
      const dim3 tgrid((j.dw + 63) / 64, (j.dh + g[p].a0 - 1) / g[p].a0, n);
#define VPF_LZTB(C) do { if (wpb == 8) launch_plane_batch<LanczosTileTask<C, 8>>(st, tgrid, lds, a, j.k, g[p]); \
                         else launch_plane_batch<LanczosTileTask<C, 4>>(st, tgrid, lds, a, j.k, g[p]); } while (0)
      if (j.ch == 1) VPF_LZTB(1); else if (j.ch == 2) VPF_LZTB(2); else VPF_LZTB(3);
#undef VPF_LZTB
    } else if (fam[p] == FAM_ROWPAIR) {
      g[p].a0 = rowb[p] / 16;
      const BandPlan bs = plan_band(1, &j, n, a);
      const int band = bs.rows;
      if (band > 1) {
        g[p].a0 = bs.rb / 16; g[p].a1 = bs.slots; g[p].a2 = bs.nb;
        const uint32_t wcols = 64u * band_px(j.ch, bs.p1), nbw = bs.nb ? bs.nb : 1u;
        const dim3 bgrid((j.dw + wcols - 1) / wcols, (j.dh + 4 * band * nbw - 1) / (4 * band * nbw), n);
        const uint32_t blds = 4 * bs.slots * bs.rb + 16;
#define VPF_RBB(C) do { if (band == 4 && bs.p1 == 8) launch_plane_batch<RowBandTask<C, 4, 1, 8, true>>(st, bgrid, blds, a, j.k, g[p]); else if (band == 16 && bs.p1 == 8) launch_plane_batch<RowBandTask<C, 16, 1, 8>>(st, bgrid, blds, a, j.k, g[p]); else if (band == 8 && bs.p1 == 8) launch_plane_batch<RowBandTask<C, 8, 1, 8>>(st, bgrid, blds, a, j.k, g[p]); \
                        else if (band == 16) launch_plane_batch<RowBandTask<C, 16, 1>>(st, bgrid, blds, a, j.k, g[p]); else if (band == 8 && bs.narrow) launch_plane_batch<RowBandTask<C, 8, 1>>(st, bgrid, blds, a, j.k, g[p]); \
                        else if (band == 8) launch_plane_batch<RowBandTask<C, 8>>(st, bgrid, blds, a, j.k, g[p]); else if (band == 4) launch_plane_batch<RowBandTask<C, 4>>(st, bgrid, blds, a, j.k, g[p]); \
                        else launch_plane_batch<RowBandTask<C, 2>>(st, bgrid, blds, a, j.k, g[p]); } while (0)
        if (j.ch == 1) VPF_RBB(1); else if (j.ch == 2) VPF_RBB(2); else VPF_RBB(3);
#undef VPF_RBB
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        continue;
      }
      const uint32_t it = (rowb[p] + 1023) / 1024, lds = 4 * 2 * rowb[p] + 16;
#define VPF_RPB(C) do { if (it == 1) launch_plane_batch<RowPairTask<C, 1>>(st, grid4, lds, a, j.k, g[p]); else if (it == 2) launch_plane_batch<RowPairTask<C, 2>>(st, grid4, lds, a, j.k, g[p]); \
                        else if (it == 3) launch_plane_batch<RowPairTask<C, 3>>(st, grid4, lds, a, j.k, g[p]); else launch_plane_batch<RowPairTask<C, 4>>(st, grid4, lds, a, j.k, g[p]); } while (0)
      if (j.ch == 1) VPF_RPB(1); else if (j.ch == 2) VPF_RPB(2); else VPF_RPB(3);
#undef VPF_RPB
    } else if (fam[p] == FAM_LZ_GATHER) {
      if (j.ch == 1) launch_plane_batch<LanczosGatherTask<1>>(st, grid1, 0, a, j.k, g[p]);
      else if (j.ch == 2) launch_plane_batch<LanczosGatherTask<2>>(st, grid1, 0, a, j.k, g[p]);
      else launch_plane_batch<LanczosGatherTask<3>>(st, grid1, 0, a, j.k, g[p]);
    } else if (eff[p] == VPF_INTERP_LINEAR) {
      launch_gather_ch<GatherTask, VPF_INTERP_LINEAR>(st, grid4, a, j, g[p]);
    } else {
      launch_gather_ch<GatherTask, VPF_INTERP_NEAREST>(st, grid4, a, j, g[p]);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace vpf
VPF_WAVE_TIMES_EXPORT(vpf_lab_wave_times_resize)
