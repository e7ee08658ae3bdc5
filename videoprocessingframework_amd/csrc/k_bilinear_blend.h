// k_bilinear_blend.h — what the bilinear kernels of k_resize.hip (row-pair, row-band, tiled) and the fused convert + resize kernels of
// k_convert_resize.hip share: the sampling convention, wave-private LDS strips, and the blend of taps that sit in such strips.
//
// Sampling convention (SURVEY.md §8c [A8]): s = (d + 0.5) * (S / D) - 0.5 clamped to [0, S-1];
// i0 = floor(s), i1 = min(i0 + 1, S - 1), f = s - i0;
//   top = fma(fx, p01 - p00, p00); bot = fma(fx, p11 - p10, p10); out = sat_trunc(fma(fy, bot - top, top) + 0.5)
#ifndef VPF_K_BILINEAR_BLEND_H_
#define VPF_K_BILINEAR_BLEND_H_
#include "vpf_device.h"
#include "vpf_plan_bounds.h"

namespace vpf {

typedef float f32x2 __attribute__((ext_vector_type(2)));  // operand pair of the packed-fp32 instructions
struct Tap {
  uint32_t i0, i1;
  float f;
};
template <int INTERP>
VPF_DEV Tap make_tap(uint32_t d, float scale, uint32_t S) {
  Tap t;
  if constexpr (INTERP == VPF_INTERP_NEAREST) {
    uint32_t i = (uint32_t)(((float)d + 0.5f) * scale);
    t.i0 = t.i1 = (i > S - 1) ? S - 1 : i;
    t.f = 0.f;
  } else {
    float s = __builtin_fmaf((float)d + 0.5f, scale, -0.5f);
    s = fmaxf(s, 0.f);
    s = fminf(s, (float)(S - 1));
    t.i0 = (uint32_t)(int)s;
    t.i1 = (t.i0 + 1 < S) ? t.i0 + 1 : S - 1;
    t.f = s - (float)t.i0;
  }
  return t;
}
VPF_DEV float bilerp(float p00, float p01, float p10, float p11, float fx, float fy) {
  const float top = __builtin_fmaf(fx, p01 - p00, p00);
  const float bot = __builtin_fmaf(fx, p11 - p10, p10);
  return __builtin_fmaf(fy, bot - top, top) + 0.5f;
}

constexpr uint32_t kResizeRowBytes = 4096;  // largest strip (IT = 4 dense 1-KiB loads per source row)
// The strips live in dynamic LDS sized for THIS launch's scale factor (strip bytes = span rounded up to 256):
// a 3x down-scale of packed RGB needs 2.5 KiB per row instead of the 4 KiB worst case, so 8 workgroups fit a CU
// instead of 5 and the load latency of one wave hides behind the arithmetic of more neighbours.
extern __shared__ u32x4 dyn_strip[];

// Copy `nq` 16-byte units of a source row (starting at the 16-B aligned byte offset `base`) into an LDS strip.
// All loads are issued before the first LDS write (MAXIT is a compile-time bound), so the wave pays ONE memory
// latency per strip, not one per 1 KiB.
template <int MAXIT>
struct Span {
  u32x4 v[MAXIT];
  VPF_DEV void load(const uint8_t* row, uint32_t base, uint32_t nq, uint32_t lane) {
#pragma unroll
    for (int k = 0; k < MAXIT; k++)
      if (lane + 64 * k < nq) v[k] = ldg<false, u32x4>(row + base + 16 * (lane + 64 * k));
  }
  VPF_DEV void store(u32x4* lds, uint32_t nq, uint32_t lane) const {
#pragma unroll
    for (int k = 0; k < MAXIT; k++)
      if (lane + 64 * k < nq) lds[lane + 64 * k] = v[k];
  }
};
// The px4 form of Span (CH = 3 strips widened to R G B x, see ColTapsX below): a unit is FOUR pixels = 12 source bytes (three dwords, 4-B aligned:
// rows are 16-B aligned and a unit starts at 12 x its index) written as 16 LDS bytes.  `lim` = bytes of the row that may be read (the row
// bytes rounded up to 16: what the 16-B units of Span touch too); dwords beyond it stay unread — they can only belong to the pixel after the
// picture's right edge, whose dword is read by the tap window with weight exactly 0.
template <int MAXIT>
struct Span4 {
  uint32_t v[MAXIT][3];
  VPF_DEV void load(const uint8_t* row, uint32_t base, uint32_t nu, uint32_t lane, uint32_t lim) {
#pragma unroll
    for (int k = 0; k < MAXIT; k++) {
      const uint32_t u = lane + 64 * k, off = base + 12 * u;
      if (u < nu) {
        if (off + 12 <= lim) {
          const uint32_t* q = reinterpret_cast<const uint32_t*>(row + off);
          v[k][0] = q[0]; v[k][1] = q[1]; v[k][2] = q[2];
        } else {
#pragma unroll
          for (int j = 0; j < 3; j++) v[k][j] = off + 4 * j + 4 <= lim ? *reinterpret_cast<const uint32_t*>(row + off + 4 * j) : 0u;
        }
      }
    }
  }
  VPF_DEV void store(u32x4* lds, uint32_t nu, uint32_t lane) const {
#pragma unroll
    for (int k = 0; k < MAXIT; k++)
      if (lane + 64 * k < nu)
        lds[lane + 64 * k] = u32x4{v[k][0], __builtin_amdgcn_alignbyte(v[k][1], v[k][0], 3), __builtin_amdgcn_alignbyte(v[k][2], v[k][1], 2), v[k][2] >> 8};
  }
};
VPF_DEV void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the two 3-byte taps that start `a` bytes into an LDS strip (any alignment): 12-B window from the dword below + v_alignbyte_b32
VPF_DEV void strip_window_taps(const uint8_t* strip, uint32_t a, float* t0, float* t1) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(strip + (a & ~3u));
  const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
  t0[0] = ubyte<0>(lo); t0[1] = ubyte<1>(lo); t0[2] = ubyte<2>(lo);
  t1[0] = ubyte<3>(lo); t1[1] = ubyte<0>(hi); t1[2] = ubyte<1>(hi);
}

// The column side of PX (four; eight in the row-band kernel's 1-channel instantiation) consecutive destination pixels of a row-pair blend:
// tap offsets inside the LDS strips and weights.  They depend on the destination columns only, so a wave that blends several destination
// rows computes them once (RowBandTask, convert_strip_task).
template <int CH, int PX = 4>
struct ColTaps {
  uint32_t a[PX], b[PX];  // byte offsets of tap 0 / tap 1 from the strips' first byte
  float f[PX];
  bool allfx;  // wave-uniform: for each of the pixels some lane has fx != 0 (no exact-alignment shortcut applies on x)
  // 2-channel planes and 1-channel planes at 8 pixels per lane, the row-band walk (band_hlerp4): the four tap bytes of one pixel (CH = 2:
  // U V of tap 0, U V of tap 1) or of a pixel PAIR (CH = 1: a, a + 1 of both pixels) come out of ONE 8-byte LDS window that starts at the
  // dword below the first of them — wa = that dword's byte offset, ws = the v_perm_b32 selector that picks the four bytes out of the window —
  // instead of four ds_read_u8.  Tap 1 is read at a + CH whatever i1 says: where i1 == i0 (the picture's right edge) its weight is exactly
  // 0 and fma(0, finite, p) == p.  A pair's bytes fit the window up to a horizontal factor of 3; the 8-pixel forms exist for strips of at
  // most 1 KiB = 512 columns x 2 (plan_band), so the choice is a compile-time one and a / b cost no registers where the windows are used.
  static constexpr bool kWindowed = CH == 2 || (CH == 1 && PX == 8);
  static constexpr int NW = CH == 1 ? PX / 2 : PX;
  uint32_t wa[NW], ws[NW];
};
template <int CH, int PX = 4>
VPF_DEV ColTaps<CH, PX> make_col_taps(uint32_t base, uint32_t x0, uint32_t dw, uint32_t sw, float scx) {
  ColTaps<CH, PX> T;
  T.allfx = true;
#pragma unroll
  for (int k = 0; k < PX; k++) {
    const Tap t = make_tap<VPF_INTERP_LINEAR>((x0 + k < dw) ? x0 + k : dw - 1, scx, sw);
    T.a[k] = CH * t.i0 - base; T.b[k] = CH * t.i1 - base; T.f[k] = t.f;
    T.allfx = T.allfx && __builtin_amdgcn_ballot_w64(t.f != 0.f) != 0;
  }
  if constexpr (ColTaps<CH, PX>::kWindowed && CH == 1) {
#pragma unroll
    for (int j = 0; j < PX / 2; j++) {
      const uint32_t A = T.a[2 * j] & ~3u, o0 = T.a[2 * j] - A, o1 = T.a[2 * j + 1] - A;  // a never decreases with the column; o1 <= 3 + 2
      T.wa[j] = A;
      T.ws[j] = o0 | (o0 + 1u) << 8 | (o1 & 7u) << 16 | ((o1 + 1u) & 7u) << 24;
    }
  } else if constexpr (ColTaps<CH, PX>::kWindowed) {
#pragma unroll
    for (int k = 0; k < PX; k++) {
      const uint32_t A = T.a[k] & ~3u, o = T.a[k] - A;  // 0 or 2: a is even
      T.wa[k] = A;
      T.ws[k] = o | (o + 1u) << 8 | (o + 2u) << 16 | (o + 3u) << 24;
    }
  }
  return T;
}

// Four consecutive destination pixels of one row blended from two source rows that sit in LDS as byte strips (`base` of make_col_taps =
// byte offset of the strips' first byte inside the source row): the arithmetic of the row-pair kernels, shared by the plain resize
// (RowPairTask, RowBandTask) and the fused convert + resize (convert_strip_task), whose strips hold freshly converted RGB.
// o[] = pixel-major, + 0.5 already added.  row1 = (fy != 0), wave-uniform.
template <int CH>
VPF_DEV void rowpair_blend4(const uint8_t* r0, const uint8_t* r1, bool row1, float fy, const ColTaps<CH>& T, float* o) {
  if (row1 && T.allfx) {
    // The common case, on the packed-fp32 pipe: the four taps of all four pixels are fetched first, then every blend step runs on
    // PIXEL PAIRS (v_pk_add_f32 / v_pk_fma_f32: two independent IEEE operations per instruction — 3.5 instead of 7 instructions per pixel
    // and channel; in cycles the gain is small, a plain fp32 fma issues in ~2.4 cycles here and a packed one in ~4.4,
    // profiles/r02_probe_valu_rate.txt).  Each component goes through exactly
    // bilerp()'s operations in bilerp()'s order -> bit-identical to the scalar form below and to the other kernels.
    float p00[4][CH], p01[4][CH], p10[4][CH], p11[4][CH];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t a = T.a[k], b = T.b[k];
      if constexpr (CH == 3) {  // both taps of a row are 6 contiguous bytes: one 12-B LDS window + v_alignbyte_b32 (see strip_window_taps)
        strip_window_taps(r0, a, p00[k], p01[k]);
        strip_window_taps(r1, a, p10[k], p11[k]);
      } else {
#pragma unroll
        for (int c = 0; c < CH; c++) { p00[k][c] = (float)r0[a + c]; p01[k][c] = (float)r0[b + c]; p10[k][c] = (float)r1[a + c]; p11[k][c] = (float)r1[b + c]; }
      }
    }
    const f32x2 fy2 = {fy, fy}, half2 = {0.5f, 0.5f};
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int k0 = 2 * j, k1 = 2 * j + 1;
      const f32x2 fx2 = {T.f[k0], T.f[k1]};
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const f32x2 a00 = {p00[k0][c], p00[k1][c]}, a01 = {p01[k0][c], p01[k1][c]}, a10 = {p10[k0][c], p10[k1][c]}, a11 = {p11[k0][c], p11[k1][c]};
        const f32x2 top = __builtin_elementwise_fma(fx2, a01 - a00, a00), bot = __builtin_elementwise_fma(fx2, a11 - a10, a10);
        const f32x2 v = __builtin_elementwise_fma(fy2, bot - top, top) + half2;
        o[k0 * CH + c] = v[0]; o[k1 * CH + c] = v[1];
      }
    }
  } else {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t a = T.a[k], b = T.b[k];
    const float fx = T.f[k];
    const bool tap1 = __builtin_amdgcn_ballot_w64(fx != 0.f) != 0;  // wave-uniform
    if constexpr (CH == 3) {
      // packed RGB: both taps of a row are 6 contiguous bytes -> three aligned dword reads + v_alignbyte_b32 instead of six
      // ds_read_u8 (the kernel spends as long issuing LDS reads as VALU work).  At the right image edge i1 == i0 and the
      // window's second tap is whatever follows the row — its weight is exactly 0 there (fma(0, finite, p) == p).
      float t0[3], t1[3], u0[3], u1[3];
      strip_window_taps(r0, a, t0, t1);
      if (row1) strip_window_taps(r1, a, u0, u1);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float top = tap1 ? __builtin_fmaf(fx, t1[c] - t0[c], t0[c]) : t0[c];
        if (row1) {
          const float bot = tap1 ? __builtin_fmaf(fx, u1[c] - u0[c], u0[c]) : u0[c];
          o[k * 3 + c] = __builtin_fmaf(fy, bot - top, top) + 0.5f;
        } else {
          o[k * 3 + c] = top + 0.5f;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const float p00 = (float)r0[a + c];
        const float top = tap1 ? __builtin_fmaf(fx, (float)r0[b + c] - p00, p00) : p00;
        if (row1) {
          const float p10 = (float)r1[a + c];
          const float bot = tap1 ? __builtin_fmaf(fx, (float)r1[b + c] - p10, p10) : p10;
          o[k * CH + c] = __builtin_fmaf(fy, bot - top, top) + 0.5f;
        } else {
          o[k * CH + c] = top + 0.5f;
        }
      }
    }
  }
  }
}
// PX blended pixels (o[] of rowpair_blend4 / band_blend_rows) -> bytes of one destination row (PX = 8: 1-channel planes only)
template <int CH, int PX = 4>
VPF_DEV void store_blend4(uint8_t* out, const float* o, bool vec4, uint32_t nv /* valid pixels, 1..PX */) {
  static_assert(PX == 4 || (PX == 8 && CH == 1), "8 pixels per lane: 1-channel planes");
  if (vec4) {
    if constexpr (CH == 3) {
      uint32_t d0, d1, d2;
      pack12_trunc(o, d0, d1, d2);
      stg3<true>(out, d0, d1, d2);
    } else if constexpr (CH == 2) {
      stg<true, u32x2>(out, u32x2{pack4_trunc(o[0], o[1], o[2], o[3]), pack4_trunc(o[4], o[5], o[6], o[7])});
    } else {
#pragma unroll
      for (int q = 0; q < PX / 4; q++) stg<true, uint32_t>(out + 4 * q, pack4_trunc(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]));
    }
  } else {
    for (uint32_t i = 0; i < nv * CH; i++) out[i] = (uint8_t)(uint32_t)o[i];
  }
}

constexpr int kBandSlots = 8;  // source rows a wave's strips can hold (twice as many when a strip is a single 1-KiB staging pass: IT = 1)
// Horizontal lerps of one strip row for a lane's PX pixels, kept as PIXEL PAIRS: H[j * CH + c] = {pixel 2j, pixel 2j + 1} of channel c — the
// operand layout of v_pk_fma_f32, so the vertical blend consumes the pairs as they are (a pixel-major array in between cost 8 v_mov per
// row and lane and turned the vertical subtractions into scalar ones).
template <int CH, int PX = 4>
VPF_DEV void band_hlerp4(const uint8_t* r, const ColTaps<CH, PX>& T, f32x2* H) {
  float p0[PX][CH], p1[PX][CH];
  if constexpr (ColTaps<CH, PX>::kWindowed) {
    if constexpr (CH == 1) {
#pragma unroll
      for (int j = 0; j < PX / 2; j++) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(r + T.wa[j]);
        const uint32_t w = __builtin_amdgcn_perm(q[1], q[0], T.ws[j]);  // [tap 0, tap 1 of pixel 2j | tap 0, tap 1 of pixel 2j + 1]
        p0[2 * j][0] = ubyte<0>(w); p1[2 * j][0] = ubyte<1>(w); p0[2 * j + 1][0] = ubyte<2>(w); p1[2 * j + 1][0] = ubyte<3>(w);
      }
    } else if constexpr (CH == 2) {
#pragma unroll
      for (int k = 0; k < PX; k++) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(r + T.wa[k]);
        const uint32_t w = __builtin_amdgcn_perm(q[1], q[0], T.ws[k]);  // [U V of tap 0 | U V of tap 1]
        p0[k][0] = ubyte<0>(w); p0[k][1] = ubyte<1>(w); p1[k][0] = ubyte<2>(w); p1[k][1] = ubyte<3>(w);
      }
    }
  } else {
#pragma unroll
  for (int k = 0; k < PX; k++) {
    if constexpr (CH == 3) {
      strip_window_taps(r, T.a[k], p0[k], p1[k]);
    } else {
#pragma unroll
      for (int c = 0; c < CH; c++) { p0[k][c] = (float)r[T.a[k] + c]; p1[k][c] = (float)r[T.b[k] + c]; }
    }
  }
  }
#pragma unroll
  for (int j = 0; j < PX / 2; j++) {
    const f32x2 fx2 = {T.f[2 * j], T.f[2 * j + 1]};
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const f32x2 a0 = {p0[2 * j][c], p0[2 * j + 1][c]}, a1 = {p1[2 * j][c], p1[2 * j + 1][c]};
      H[j * CH + c] = __builtin_elementwise_fma(fx2, a1 - a0, a0);
    }
  }
}
// ---- "px4" strips (round 6): packed RGB widened to FOUR bytes per pixel (R G B x) on its way into LDS.  A tap is then one aligned dword and both
// taps of a pixel one ds_read2_b32: against the packed strip's 12-B window a horizontal lerp of four pixels loses 8 v_alignbyte_b32, 8 v_and_b32
// and the address adds that feed them (22 of its 58 VALU instructions) and half its LDS reads, for a third more LDS per row.  The fused kernel
// writes its converted pixels that way (same cvt count: three bytes per pixel either way); the row-band resize widens 12 source bytes to 16 when
// it stages them (two v_alignbyte_b32 + one shift per four SOURCE pixels, once).  Same bytes into the same fp32 operations: same pixels.
struct ColTapsX {
  uint32_t a[4];  // byte offset of tap 0 from the strips' first byte (= 4 x pixels from the strip's first pixel); tap 1 is the next dword whatever i1
                  // says: where i1 == i0 (the picture's right edge) its weight is exactly 0 and fma(0, finite, p) == p
  float f[4];
};
VPF_DEV ColTapsX make_col_taps_x(uint32_t base_px, uint32_t x0, uint32_t dw, uint32_t sw, float scx) {
  ColTapsX T;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const Tap t = make_tap<VPF_INTERP_LINEAR>((x0 + k < dw) ? x0 + k : dw - 1, scx, sw);
    T.a[k] = 4u * (t.i0 - base_px); T.f[k] = t.f;
  }
  return T;
}
VPF_DEV void band_hlerp(const uint8_t* r, const ColTapsX& T, f32x2* H) {
  uint32_t d0[4], d1[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(r + T.a[k]);
    d0[k] = q[0]; d1[k] = q[1];
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const f32x2 fx2 = {T.f[2 * j], T.f[2 * j + 1]};
    const uint32_t a0 = d0[2 * j], a1 = d1[2 * j], b0 = d0[2 * j + 1], b1 = d1[2 * j + 1];
    const f32x2 r0 = {ubyte<0>(a0), ubyte<0>(b0)}, r1 = {ubyte<0>(a1), ubyte<0>(b1)};
    const f32x2 g0 = {ubyte<1>(a0), ubyte<1>(b0)}, g1 = {ubyte<1>(a1), ubyte<1>(b1)};
    const f32x2 c0 = {ubyte<2>(a0), ubyte<2>(b0)}, c1 = {ubyte<2>(a1), ubyte<2>(b1)};
    H[j * 3 + 0] = __builtin_elementwise_fma(fx2, r1 - r0, r0);
    H[j * 3 + 1] = __builtin_elementwise_fma(fx2, g1 - g0, g0);
    H[j * 3 + 2] = __builtin_elementwise_fma(fx2, c1 - c0, c0);
  }
}
template <int CH, int PX>
VPF_DEV void band_hlerp(const uint8_t* r, const ColTaps<CH, PX>& T, f32x2* H) { band_hlerp4<CH, PX>(r, T, H); }

// The row taps (i0, i1, fy) of a band's destination rows ya..yb, row i on lane i: one make_tap for the whole band instead of one per row on
// every lane (~12 VALU instructions per row).  Call it while ALL lanes of the wave are active (before the columns beyond the picture's
// right edge leave): band_blend_rows reads lanes 0..R-1 with v_readlane.  The empty asm pins the computation to this place — without it the
// compiler sinks it below the early return, where lanes that have left no longer compute their row.
VPF_DEV Tap band_row_taps(uint32_t ya, uint32_t yb, float scy, uint32_t sh) {
  const uint32_t ly = ya + (threadIdx.x & 63u);
  Tap t = make_tap<VPF_INTERP_LINEAR>(ly < yb ? ly : yb, scy, sh);
  asm volatile("" : "+v"(t.i0), "+v"(t.i1), "+v"(t.f));
  return t;
}
// Destination rows ya..yb (at most R <= 64) of a band whose source rows r_lo.. sit in LDS `rowbytes` apart; `rows` = band_row_taps(ya, yb, ..).
// The horizontal lerps of the two current source rows stay in registers and move up (Hb -> Ha) as the destination rows walk down.
// put(y, o) receives o[] = pixel-major, + 0.5 added.
// (Measured and not kept: exchanging the ROLES of Ha / Hb instead of copying — as a two-way branch the compiler folds the two blends back
// into one behind more copies; as two alternating loops over the rows the walk is 1.7 x slower at R = 4: profiles/r03_band_walk_variants.txt.)
// The walk's state: the horizontal lerps (pixel pairs, band_hlerp4) of source rows ida (upper tap) and idb (lower tap).  A wave that walks
// down several bands in a row (RowBandTask with bands per wave > 1) keeps it from band to band: the next band's first source rows are
// this band's last, and their lerps are not evaluated again.
template <int CH, int PX = 4>
struct BandWalk {
  static constexpr int NP = PX / 2 * CH;
  f32x2 Ha[NP], Hb[NP];
  uint32_t ida = 0xffffffffu, idb = 0xffffffffu;
};
template <int CH, int R, int PX = 4, class TAPS, class Put>  // TAPS: ColTaps<CH, PX>, or ColTapsX (px4 strips: CH = 3, PX = 4)
VPF_DEV void band_blend_rows(const uint8_t* strips, uint32_t rowbytes, uint32_t r_lo, uint32_t ya, uint32_t yb, const Tap& rows, const TAPS& T, BandWalk<CH, PX>& w,
                             Put&& put) {
  static_assert(R <= 64, "one lane per row of the band");
  constexpr int NP = PX / 2 * CH;
#pragma unroll
  for (int i = 0; i < R; i++) {
    if (ya + i > yb) break;
    const uint32_t i0 = __builtin_amdgcn_readlane(rows.i0, i), i1 = __builtin_amdgcn_readlane(rows.i1, i);
    const float fy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rows.f), i));
    if (i0 != w.ida) {
      if (i0 == w.idb) {
#pragma unroll
        for (int q = 0; q < NP; q++) w.Ha[q] = w.Hb[q];
      } else {
        band_hlerp(strips + (size_t)(i0 - r_lo) * rowbytes, T, w.Ha);
      }
      w.ida = i0;
    }
    if (i1 != w.idb) {
      if (i1 == w.ida) {
#pragma unroll
        for (int q = 0; q < NP; q++) w.Hb[q] = w.Ha[q];
      } else {
        band_hlerp(strips + (size_t)(i1 - r_lo) * rowbytes, T, w.Hb);
      }
      w.idb = i1;
    }
    float o[PX * CH];
    const f32x2 fy2 = {fy, fy}, half2 = {0.5f, 0.5f};
#pragma unroll
    for (int j = 0; j < PX / 2; j++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const f32x2 t = w.Ha[j * CH + c], b = w.Hb[j * CH + c];
        const f32x2 v = __builtin_elementwise_fma(fy2, b - t, t) + half2;
        o[2 * j * CH + c] = v[0]; o[(2 * j + 1) * CH + c] = v[1];
      }
    }
    put(ya + i, o);
  }
}
template <int CH, int R, int PX = 4, class TAPS, class Put>
VPF_DEV void band_blend_rows(const uint8_t* strips, uint32_t rowbytes, uint32_t r_lo, uint32_t ya, uint32_t yb, const Tap& rows, const TAPS& T, Put&& put) {
  BandWalk<CH, PX> w;
  band_blend_rows<CH, R, PX>(strips, rowbytes, r_lo, ya, yb, rows, T, w, static_cast<Put&&>(put));
}

// strip bytes a wave needs for the source span of its `cols` destination columns (vpf_bound_strip_bytes, checked on the CPU);
// 0 when the LDS path does not apply (forced generic, unaligned source, span above the cap)
static inline uint32_t lds_strip_bytes(int ch, uint32_t sw, uint32_t dw, const void* src, uint32_t sp, uint32_t row_bytes_cap, uint32_t cols = 256) {
  if (tuning(VPF_TUNE_NV12_RGB_VARIANT) == 9) return 0;  // forced generic
  if (((uintptr_t)src | sp) & 15) return 0;
  return vpf_bound_strip_bytes(ch, sw, dw, row_bytes_cap, cols);
}

}  // namespace vpf
#endif  // VPF_K_BILINEAR_BLEND_H_
