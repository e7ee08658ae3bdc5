// Probe: issue rate of the integer MFMAs the Lanczos kernel uses, on one wave per SIMD and on two (cycles per instruction per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o probe_mfma_rate probe_mfma_rate.hip && ./probe_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k_rate(int* out, int iters, unsigned long long* cyc) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  v16i d0 = {}, d1 = {};
  v4f f0 = {0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0;
  v8bf ba, bb;
  int x[16];
  for (int k = 0; k < 16; k++) x[k] = threadIdx.x * (k + 1);
  for (int i = 0; i < 8; i++) { ba[i] = (__bf16)(float)(threadIdx.x + i); bb[i] = (__bf16)(float)(i + 1); }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if constexpr (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
    } else if constexpr (KIND == 1) {
      d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, d1, 0, 0, 0);
    } else if constexpr (KIND == 2) {
      f0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, f0, 0, 0, 0); f1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, f1, 0, 0, 0);
      f2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, f2, 0, 0, 0); f3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, f3, 0, 0, 0);
    } else if constexpr (KIND == 3) {  // 4 MFMAs + 16 independent 4-cycle VALU instructions: do they hide behind the matrix pipe?
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 16; k++) x[k] = __builtin_amdgcn_alignbyte(x[k], i, 1);
    } else if constexpr (KIND == 4) {  // the same matrix work as two 32x32x32 instructions, the same 16 VALU
      d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 16; k++) x[k] = __builtin_amdgcn_alignbyte(x[k], i, 1);
    } else if constexpr (KIND == 6) {  // the CDNA3-era K = 32 form (8-byte operands): half the products — half the time?
      const long la = ((long)a[1] << 32) | (unsigned)a[0], lb = ((long)b[1] << 32) | (unsigned)b[0];
      c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c3, 0, 0, 0);
    } else if constexpr (KIND == 7) {  // MFMA and VALU interleaved one to four INSIDE the instruction stream (KIND 3 has them in two blocks)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (q == 0) c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        if (q == 1) c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        if (q == 2) c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
        if (q == 3) c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; k++) x[4 * q + k] = __builtin_amdgcn_alignbyte(x[4 * q + k], i, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (KIND == 8) {  // one MFMA per TWO VALU instructions (pass 2's ratio), interleaved
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (q % 4 == 0) c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        if (q % 4 == 1) c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        if (q % 4 == 2) c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
        if (q % 4 == 3) c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 2; k++) x[2 * q + k] = __builtin_amdgcn_alignbyte(x[2 * q + k], i, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {  // the 16 VALU alone
#pragma unroll
      for (int k = 0; k < 16; k++) x[k] = __builtin_amdgcn_alignbyte(x[k], i, 1);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int k = 0; k < 4; k++) s += c0[k] + c1[k] + c2[k] + c3[k] + (int)f0[k] + (int)f1[k] + (int)f2[k] + (int)f3[k] + a[k] + b[k];
  for (int k = 0; k < 16; k++) s += d0[k] + d1[k] + x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND>
static void run(const char* name, int threads, int wgs_per_cu = 4) {
  int* out; unsigned long long* cyc; hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k_rate<KIND><<<256 * wgs_per_cu, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  // waves per SIMD over the whole run: 1024 workgroups x (threads / 64) waves on 1024 SIMDs, each wave 4 * iters MFMAs
  const double waves_per_simd = 256.0 * wgs_per_cu * (threads / 64) / 1024.0;
  const double cyc_per = best * 1e-3 * 2.4e9 / (waves_per_simd * 4.0 * iters);
  unsigned long long hc = 0; (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("[mfma-rate] %-46s %4.1f waves/SIMD: %.3f ms -> %.1f cycles per quarter-iteration per SIMD at 2.4 GHz | s_memtime of one wave: %llu ticks = %.1f per quarter-iteration per SIMD\n", name, waves_per_simd, best, cyc_per,
         hc, (double)hc / (waves_per_simd * 4.0 * iters));
}
int main() {
  run<0>("v_mfma_i32_16x16x64_i8", 256, 1); run<0>("v_mfma_i32_16x16x64_i8", 256, 2); run<0>("v_mfma_i32_16x16x64_i8", 256); run<0>("v_mfma_i32_16x16x64_i8", 512);
  run<1>("v_mfma_i32_32x32x32_i8", 256);
  run<2>("v_mfma_f32_16x16x32_bf16", 256, 1); run<2>("v_mfma_f32_16x16x32_bf16", 256);
  run<3>("v_mfma_i32_16x16x64_i8 x4 + 16 v_alignbyte_b32", 256, 1); run<3>("v_mfma_i32_16x16x64_i8 x4 + 16 v_alignbyte_b32", 256, 2); run<3>("v_mfma_i32_16x16x64_i8 x4 + 16 v_alignbyte_b32", 256); run<3>("v_mfma_i32_16x16x64_i8 x4 + 16 v_alignbyte_b32", 512);
  run<4>("v_mfma_i32_32x32x32_i8 x2 + 16 v_alignbyte_b32", 256, 1); run<4>("v_mfma_i32_32x32x32_i8 x2 + 16 v_alignbyte_b32", 256, 2); run<4>("v_mfma_i32_32x32x32_i8 x2 + 16 v_alignbyte_b32", 256); run<4>("v_mfma_i32_32x32x32_i8 x2 + 16 v_alignbyte_b32", 512);
  run<6>("v_mfma_i32_16x16x32_i8", 256, 1); run<6>("v_mfma_i32_16x16x32_i8", 256, 2); run<6>("v_mfma_i32_16x16x32_i8", 256);
  run<7>("interleaved: (1 MFMA + 4 v_alignbyte_b32) x 4", 256, 1); run<7>("interleaved: (1 MFMA + 4 v_alignbyte_b32) x 4", 256, 2); run<7>("interleaved: (1 MFMA + 4 v_alignbyte_b32) x 4", 256); run<7>("interleaved: (1 MFMA + 4 v_alignbyte_b32) x 4", 512);
  run<8>("interleaved: (1 MFMA + 2 v_alignbyte_b32) x 8 [per 2 MFMAs]", 256, 1); run<8>("interleaved: (1 MFMA + 2 v_alignbyte_b32) x 8 [per 2 MFMAs]", 256, 2); run<8>("interleaved: (1 MFMA + 2 v_alignbyte_b32) x 8 [per 2 MFMAs]", 256);
  run<5>("16 v_alignbyte_b32 alone", 256, 1); run<5>("16 v_alignbyte_b32 alone", 256); run<5>("16 v_alignbyte_b32 alone", 512);
  return 0;
}
