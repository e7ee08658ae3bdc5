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
// four values already carrying +0.5: saturate + truncate (round half up), for resize / remap
VPF_DEV uint32_t pack4_trunc(float a, float b, float c, float d) {
  return sat_trunc(a) | (sat_trunc(b) << 8) | (sat_trunc(c) << 16) | (sat_trunc(d) << 24);
}

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

}  // namespace vpf
