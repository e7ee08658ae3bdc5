// tools/probe_valu_rate.hip — issue rate of the VALU instructions the resize / remap kernels lean on (gfx950), measured, not assumed:
// each kernel runs a long unrolled stream of ONE instruction kind on 8 independent register chains per lane, with enough waves to fill
// every SIMD (8 per SIMD), and reports wave-instructions per SIMD-cycle from hipEvent time and the device clock rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu_rate.hip -o tools/probe_valu_rate.bin && tools/probe_valu_rate.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Op { FMA, DOT2, DOT4, PERM, ALIGNBYTE, CVT_F32_UBYTE, CVT_F32_I32, CVT_PK_U8, MED3, MAD_U24, MUL_LO, PK_FMA, LSHL_OR, RNDNE, ADD_U32, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_dot2c_i32_i16", "v_dot4_u32_u8", "v_perm_b32", "v_alignbyte_b32", "v_cvt_f32_ubyte0", "v_cvt_f32_i32",
                                     "v_cvt_pk_u8_f32", "v_med3_f32", "v_mad_u32_u24", "v_mul_lo_u32", "v_pk_fma_f32 (2 fma)", "v_lshl_or_b32", "v_rndne_f32", "v_add_u32"};

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8];
  float f[8];
  f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = seed * (i + 1) + threadIdx.x; f[i] = (float)a[i] * 1e-3f; p[i] = f32x2{f[i], f[i] + 1.f}; }
  const uint32_t s = seed | 1u;
  const float fs = (float)s * 1e-9f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 8; rep++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if constexpr (OP == FMA) f[i] = __builtin_fmaf(f[i], fs, 1.0f);
        else if constexpr (OP == DOT2) a[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a[i]), __builtin_bit_cast(s16x2, s), (int)a[i], false);
        else if constexpr (OP == DOT4) a[i] = __builtin_amdgcn_udot4(a[i], s, a[i], false);
        else if constexpr (OP == PERM) a[i] = __builtin_amdgcn_perm(a[i], s, a[(i + 1) & 7] | 0x01020304u);
        else if constexpr (OP == ALIGNBYTE) a[i] = __builtin_amdgcn_alignbyte(a[i], s, a[(i + 1) & 7]);
        else if constexpr (OP == CVT_F32_UBYTE) f[i] = (float)(__float_as_uint(f[i]) & 0xffu) + 0.0f;
        else if constexpr (OP == CVT_F32_I32) f[i] = (float)(int)__float_as_uint(f[i]);
        else if constexpr (OP == CVT_PK_U8) a[i] = __builtin_amdgcn_cvt_pk_u8_f32(__uint_as_float(a[i]), 1, a[i]);
        else if constexpr (OP == MED3) f[i] = __builtin_amdgcn_fmed3f(f[i], fs, 255.0f);
        else if constexpr (OP == MAD_U24) a[i] = __umul24(a[i], s) + a[i];
        else if constexpr (OP == MUL_LO) a[i] = a[i] * s;
        else if constexpr (OP == PK_FMA) p[i] = __builtin_elementwise_fma(p[i], f32x2{fs, fs}, f32x2{1.0f, 1.0f});
        else if constexpr (OP == LSHL_OR) a[i] = (a[i] << 16) | s;
        else if constexpr (OP == RNDNE) f[i] = __builtin_rintf(f[i]) * 1.0f;
        else a[i] = a[i] + s;
      }
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r ^= a[i] ^ __float_as_uint(f[i]) ^ __float_as_uint(p[i][0]) ^ __float_as_uint(p[i][1]);
  if (r == 0x12345u) out[threadIdx.x] = r;
}

template <int OP>
static void run(uint32_t* out, int cus, double mhz) {
  const int iters = 2000, per_iter = 64;
  const dim3 grid(cus * 8), block(256);  // 8 workgroups x 4 waves per CU = 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 10, 3u);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, iters, 3u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = (double)grid.x * 4 * iters * per_iter, simd_cycles = ms * 1e-3 * mhz * 1e6 * cus * 4;
  printf("%-24s %7.3f ms  %6.3f wave-instr / SIMD-cycle  (%.2f cycles per wave-instruction)\n", kNames[OP], ms, wave_instr / simd_cycles, simd_cycles / wave_instr);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double mhz = p.clockRate / 1000.0;
  printf("%s: %d CUs, clockRate %.0f MHz (rates below assume that clock for the whole run)\n", p.name, p.multiProcessorCount, mhz);
  uint32_t* out;
  hipMalloc(&out, 4096);
  run<FMA>(out, p.multiProcessorCount, mhz); run<DOT2>(out, p.multiProcessorCount, mhz); run<DOT4>(out, p.multiProcessorCount, mhz);
  run<PERM>(out, p.multiProcessorCount, mhz); run<ALIGNBYTE>(out, p.multiProcessorCount, mhz); run<CVT_F32_UBYTE>(out, p.multiProcessorCount, mhz);
  run<CVT_F32_I32>(out, p.multiProcessorCount, mhz); run<CVT_PK_U8>(out, p.multiProcessorCount, mhz); run<MED3>(out, p.multiProcessorCount, mhz);
  run<MAD_U24>(out, p.multiProcessorCount, mhz); run<MUL_LO>(out, p.multiProcessorCount, mhz); run<PK_FMA>(out, p.multiProcessorCount, mhz);
  run<LSHL_OR>(out, p.multiProcessorCount, mhz); run<RNDNE>(out, p.multiProcessorCount, mhz); run<ADD_U32>(out, p.multiProcessorCount, mhz);
  return 0;
}
