// Probe: how should one wave's instruction stream alternate between the matrix pipe and the vector ALU when TWO waves share a SIMD (the
// matrix-core Lanczos kernel's occupancy)?  Each iteration issues 16 MFMAs and 80 VALU instructions (5 per MFMA: the kernel's mix), in
// blocks of B MFMAs followed by 5 B VALU instructions, B = 1 .. 16; the first 4 VALU of every group of 5 read the MFMA's result (DEP) or
// not.  Wall-clock cycles per MFMA-plus-5-VALU per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o probe_mfma_block probe_mfma_block.hip && ./probe_mfma_block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int B, bool DEP>
__global__ __launch_bounds__(256, 2) void k_block(int* out, int iters) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  v4i c[16];
  int x[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { c[k] = v4i{k, 0, 0, 0}; x[k] = threadIdx.x * (k + 1); }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int g = 0; g < 16 / B; g++) {
#pragma unroll
      for (int m = 0; m < B; m++) c[g * B + m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[g * B + m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < B; m++) {
        const int k = g * B + m;
        if constexpr (DEP) {
          x[k] = (x[k] << 8) + c[k][0]; x[(k + 1) & 15] = (x[(k + 1) & 15] << 8) + c[k][1];
          x[(k + 2) & 15] = (x[(k + 2) & 15] << 8) + c[k][2]; x[(k + 3) & 15] = (x[(k + 3) & 15] << 8) + c[k][3];
        } else {
          x[k] = (x[k] << 8) + i; x[(k + 1) & 15] = (x[(k + 1) & 15] << 8) + i;
          x[(k + 2) & 15] = (x[(k + 2) & 15] << 8) + i; x[(k + 3) & 15] = (x[(k + 3) & 15] << 8) + i;
        }
        x[(k + 4) & 15] = __builtin_amdgcn_alignbyte(x[(k + 4) & 15], i, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  int s = 0;
  for (int k = 0; k < 16; k++) s += x[k] + c[k][0] + c[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int B, bool DEP>
static void run(int wgs_per_cu) {
  int* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 5000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0);
    k_block<B, DEP><<<256 * wgs_per_cu, 256>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("[mfma-block] B = %2d %s  %d waves/SIMD: %.3f ms -> %.1f cycles per (MFMA + 5 VALU) per SIMD at 2.4 GHz\n", B, DEP ? "dependent  " : "independent", wgs_per_cu, best,
         best * 1e-3 * 2.4e9 / (wgs_per_cu * 16.0 * iters));
  (void)hipFree(out);
}
int main() {
  for (int w = 1; w <= 3; w++) {
    run<1, true>(w); run<2, true>(w); run<4, true>(w); run<8, true>(w); run<16, true>(w);
    run<1, false>(w); run<2, false>(w); run<4, false>(w); run<8, false>(w); run<16, false>(w);
  }
  return 0;
}
