// vpf_device.h — device-side helpers shared by the gfx950 kernels.
#pragma once
#include "vpf_internal.h"

namespace vpf {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define VPF_DEV __device__ __forceinline__

// byte k of a dword as float: compiles to v_cvt_f32_ubyte{k} (one VALU op, no shift/mask)
template <int K>
VPF_DEV float ubyte(uint32_t d) {
  return (float)((d >> (8 * K)) & 0xffu);
}

// saturate to [0,255] and truncate; `t` already carries the +0.5 of round-half-up
// (used by resize / remap / RGB->YUV: round half up)
VPF_DEV uint32_t sat_trunc(float t) { return (uint32_t)__builtin_amdgcn_fmed3f(t, 0.0f, 255.0f); }

// saturate to [0,255] with round-to-nearest-even: ONE instruction, v_cvt_pk_u8_f32.  Its semantics
// (saturating, ties to even, NaN -> 0) were measured on gfx950 with tools/probe_cvt.hip
// (profiles/r01_probe_cvt_pk_u8_f32.txt): 0 mismatches against clamp(rint(x)) over a dense sweep.
// This is the rounding of the whole YUV -> RGB family.
VPF_DEV uint32_t sat_rne(float t) { return __builtin_amdgcn_cvt_pk_u8_f32(t, 0, 0u); }
// the same value spelled with language-level operations (cross-check variant)
VPF_DEV uint32_t sat_rne_explicit(float t) { return (uint32_t)__builtin_rintf(__builtin_amdgcn_fmed3f(t, 0.0f, 255.0f)); }

// pack four channel values into one dword, byte 0 first, each saturated + rounded to nearest even.
// PACK = 1: four v_cvt_pk_u8_f32 (one op per byte, writes straight into the destination byte lane)
// PACK = 0: v_med3_f32 + v_rndne_f32 + v_cvt_u32_f32 + shifts/ors; bit-identical, kept as a cross-check
template <int PACK>
VPF_DEV uint32_t pack4(float a, float b, float c, float d) {
  if constexpr (PACK == 1) {
    uint32_t o = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0u);
    o = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, o);
    o = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, o);
    o = __builtin_amdgcn_cvt_pk_u8_f32(d, 3, o);
    return o;
  } else {
    return sat_rne_explicit(a) | (sat_rne_explicit(b) << 8) | (sat_rne_explicit(c) << 16) | (sat_rne_explicit(d) << 24);
  }
}
// three channel values -> bytes 0..2 of a dword (byte 3 = 0): one pixel of a "px4" LDS strip (k_bilinear_blend.h), vpf_convert's rounding
VPF_DEV uint32_t pack3(float a, float b, float c) {
  uint32_t o = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0u);
  o = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, o);
  return __builtin_amdgcn_cvt_pk_u8_f32(c, 2, o);
}
// four values already carrying +0.5: saturate + truncate (round half up), for resize / remap / RGB -> YUV.
// v_cvt_pk_u8_f32 follows the fp32 rounding field of the MODE register (measured: tools/lab/probes/probe_rtz_pack.hip, profiles/
// r03_probe_rtz_pack.txt — 0 of 4 M groups differ from med3 + v_cvt_u32_f32, values from -1e30 to 1e30 included), so under round-toward-zero
// it IS "saturate, truncate, insert byte": four instructions per dword between two s_setreg instead of 4 v_med3 + 4 v_cvt_u32 + 3 v_lshl_or.
// The two s_setreg sit inside the one asm statement, so no other floating-point instruction can run under the changed mode.
VPF_DEV uint32_t pack4_trunc(float a, float b, float c, float d) {
  uint32_t o;
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %0, %2, 1, %0\n\t"
      "v_cvt_pk_u8_f32 %0, %3, 2, %0\n\t"
      "v_cvt_pk_u8_f32 %0, %4, 3, %0\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=&v"(o) : "v"(a), "v"(b), "v"(c), "v"(d));
  return o;
}
// twelve values -> three dwords under one mode switch (packed RGB: four pixels)
VPF_DEV void pack12_trunc(const float* v, uint32_t& d0, uint32_t& d1, uint32_t& d2) {
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %2, %11, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %12, 1, %2\n\t"
      "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %13, 2, %2\n\t"
      "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=&v"(d0), "=&v"(d1), "=&v"(d2)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]));
}
// a * b + c for a, b < 2^24 (low 32 bits): v_mad_u32_u24, full rate (v_mul_lo_u32 / v_mad_u64_u32 are quarter-rate)
VPF_DEV uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }

struct Chroma {
  float rc, gc, bc;
};
VPF_DEV Chroma chroma_terms(const Yuv2RgbCoef& c, float u, float v) {
  Chroma k;
  k.rc = __builtin_fmaf(v, c.rv, c.br);
  k.gc = __builtin_fmaf(u, c.gu, __builtin_fmaf(v, c.gv, c.bg));
  k.bc = __builtin_fmaf(u, c.bu, c.bb);
  return k;
}

// One dword of luma (4 px) + the two chroma samples covering it -> three channel quads.
struct Quad {
  float r[4], g[4], b[4];
};
VPF_DEV Quad convert4(const Yuv2RgbCoef& c, uint32_t yd, const Chroma& k0, const Chroma& k1) {
  Quad q;
  float y0 = ubyte<0>(yd), y1 = ubyte<1>(yd), y2 = ubyte<2>(yd), y3 = ubyte<3>(yd);
  q.r[0] = __builtin_fmaf(y0, c.cy, k0.rc); q.g[0] = __builtin_fmaf(y0, c.cy, k0.gc); q.b[0] = __builtin_fmaf(y0, c.cy, k0.bc);
  q.r[1] = __builtin_fmaf(y1, c.cy, k0.rc); q.g[1] = __builtin_fmaf(y1, c.cy, k0.gc); q.b[1] = __builtin_fmaf(y1, c.cy, k0.bc);
  q.r[2] = __builtin_fmaf(y2, c.cy, k1.rc); q.g[2] = __builtin_fmaf(y2, c.cy, k1.gc); q.b[2] = __builtin_fmaf(y2, c.cy, k1.bc);
  q.r[3] = __builtin_fmaf(y3, c.cy, k1.rc); q.g[3] = __builtin_fmaf(y3, c.cy, k1.gc); q.b[3] = __builtin_fmaf(y3, c.cy, k1.bc);
  return q;
}
// loads / stores with optional non-temporal hint (streamed once, never re-read)
template <bool NT, typename T>
VPF_DEV T ldg(const void* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
  else return *reinterpret_cast<const T*>(p);
}
template <bool NT, typename T>
VPF_DEV void stg(void* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
  else *reinterpret_cast<T*>(p) = v;
}
// 12-byte store (three dwords); the backend merges these into one global_store_dwordx3
template <bool NT>
VPF_DEV void stg3(void* p, uint32_t a, uint32_t b, uint32_t c) {
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
  stg<NT, uint32_t>(q, a);
  stg<NT, uint32_t>(q + 1, b);
  stg<NT, uint32_t>(q + 2, c);
}

// The chroma of 16 luma pixels as four dwords of interleaved U V U V (NV12's native form).  YUV420 (I420, what software
// decoders hand over) has U and V in separate half-width planes: two 8-B loads and four v_perm_b32 re-create the same form.
template <int SRC, bool NT>
VPF_DEV u32x4 load_uv16(const FrameDesc& f, uint32_t rp, uint32_t x) {
  if constexpr (SRC == FC_NV12) {
    return ldg<NT, u32x4>(f.s[1] + (size_t)rp * f.sp[1] + x);
  } else {
    const u32x2 u = ldg<NT, u32x2>(f.s[1] + (size_t)rp * f.sp[1] + (x >> 1)), v = ldg<NT, u32x2>(f.s[2] + (size_t)rp * f.sp[2] + (x >> 1));
    return u32x4{__builtin_amdgcn_perm(v[0], u[0], 0x05010400u), __builtin_amdgcn_perm(v[0], u[0], 0x07030602u),
                 __builtin_amdgcn_perm(v[1], u[1], 0x05010400u), __builtin_amdgcn_perm(v[1], u[1], 0x07030602u)};
  }
}

// make one wave's LDS writes visible to its own other lanes (wave-private tiles: no workgroup barrier needed)
VPF_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- byte shuffles on groups of 4 packed 3-channel pixels (12 B = 3 dwords), all v_perm_b32 ----
// d0 = c0a c1a c2a c0b ; d1 = c1b c2b c0c c1c ; d2 = c2c c0d c1d c2d   (perm(hi, lo, sel): sel 0-3 -> lo, 4-7 -> hi)
VPF_DEV void deint4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t& c0, uint32_t& c1, uint32_t& c2) {
  c0 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(d1, d0, 0x00000300u), __builtin_amdgcn_perm(d2, d1, 0x00000502u), 0x01000504u);
  c1 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(d1, d0, 0x00000401u), __builtin_amdgcn_perm(d2, d1, 0x00000603u), 0x01000504u);
  c2 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(d1, d0, 0x00000502u), __builtin_amdgcn_perm(d2, d1, 0x00000704u), 0x01000504u);
}
VPF_DEV void inter4(uint32_t r, uint32_t g, uint32_t b, uint32_t& d0, uint32_t& d1, uint32_t& d2) {
  const uint32_t rg_lo = __builtin_amdgcn_perm(g, r, 0x05010400u);  // R0 G0 R1 G1
  const uint32_t rg_hi = __builtin_amdgcn_perm(g, r, 0x07030602u);  // R2 G2 R3 G3
  d0 = __builtin_amdgcn_perm(b, rg_lo, 0x02040100u);                // R0 G0 B0 R1
  d1 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(b, rg_lo, 0x00000503u) /* G1 B1 . . */, rg_hi, 0x01000504u);  // G1 B1 R2 G2
  d2 = __builtin_amdgcn_perm(b, rg_hi, 0x07030206u);                // B2 R3 G3 B3
}
VPF_DEV void swap4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t& o0, uint32_t& o1, uint32_t& o2) {
  // o0 = c2a c1a c0a c2b ; o1 = c1b c0b c2c c1c ; o2 = c0c c2d c1d c0d
  o0 = __builtin_amdgcn_perm(d1, d0, 0x05000102u);
  o1 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(d2, d0, 0x04000003u) /* c0b . . c2c */, d1, 0x03070400u);
  o2 = __builtin_amdgcn_perm(d2, d1, 0x05060702u);
}

// One wave <-> one 3-KiB run of packed pixels (1024 px x 3 B) through a wave-private LDS tile of 192 x 16 B:
// global side = three dense 1-KiB accesses (lane-contiguous 16 B), register side = 48 contiguous bytes (16 px) per lane.
// ds_write/read_b128 at 48*lane + 16*j and 16*(64k + lane) are both bank-conflict-free (profiles/r01_pmc_sq.json).
// `row` points at byte 0 of the run's row, `base` = byte offset of the run, `row_bytes` = 3 * width (multiple of 16).
// the global half: three dense 16-B loads per lane (kernels that read several rows issue all of them before transposing any)
VPF_DEV void run48_fetch(const uint8_t* row, uint32_t base, uint32_t row_bytes, uint32_t lane, u32x4 q[3]) {
#pragma unroll
  for (int k = 0; k < 3; k++) {  // clamp instead of predicate: all three loads issue back to back; the clamped
    uint32_t off = base + (k * 64 + lane) * 16;  // duplicates land in slots only out-of-range lanes would read
    off = off < row_bytes ? off : row_bytes - 16;
    q[k] = ldg<true, u32x4>(row + off);
  }
}
// the LDS half: lane-contiguous 16-B units -> 48 contiguous bytes per lane
VPF_DEV void run48_transpose(u32x4* t, uint32_t lane, const u32x4 q[3], uint32_t d[12]) {
#pragma unroll
  for (int k = 0; k < 3; k++) t[k * 64 + lane] = q[k];
  wave_sync();
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const u32x4 v = t[lane * 3 + j];
    d[4 * j] = v[0]; d[4 * j + 1] = v[1]; d[4 * j + 2] = v[2]; d[4 * j + 3] = v[3];
  }
}
VPF_DEV void load_run48(u32x4* t, const uint8_t* row, uint32_t base, uint32_t row_bytes, uint32_t lane, uint32_t d[12]) {
  u32x4 q[3];
  run48_fetch(row, base, row_bytes, lane, q);
  run48_transpose(t, lane, q, d);
}
VPF_DEV void store_run48(u32x4* t, uint8_t* row, uint32_t base, uint32_t row_bytes, uint32_t lane, const uint32_t d[12]) {
#pragma unroll
  for (int j = 0; j < 3; j++) t[lane * 3 + j] = u32x4{d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]};
  wave_sync();
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t off = base + (k * 64 + lane) * 16;
    if (off < row_bytes) stg<true, u32x4>(row + off, t[k * 64 + lane]);
  }
}

}  // namespace vpf
