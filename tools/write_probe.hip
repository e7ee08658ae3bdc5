// write_probe: which store geometry reaches the HBM write ceiling on gfx950?  Linear 1 GiB buffer, write-only kernels.
//   A  wave writes 6 x 1 KiB (16 B/lane, lanes contiguous), one 6 KiB task per wave, 4 waves/block   (converter's pattern)
//   B  same stores, persistent grid (2048 blocks), block-contiguous ranges
//   C  same stores, persistent grid, grid-strided tasks
//   D  thread writes 64 contiguous bytes (4 x 16 B), block covers 16 KiB                      (ATen-style vectorised loop)
//   E  A with 1024-thread blocks;  F  A with plain (non-NT) stores;  G  16 KiB per wave (16 stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT, int PER_WAVE_KIB>
__global__ void kA(u32x4* out, size_t n16) {  // n16 = number of 16-byte units
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  const size_t base = wave * (PER_WAVE_KIB * 64);
#pragma unroll
  for (int k = 0; k < PER_WAVE_KIB; k++) { const size_t i = base + k * 64 + lane; if (i < n16) st<NT>(out + i, u32x4{(uint32_t)i, 1, 2, 3}); }
}
// MODE 0: round-robin inside the block (round k: the block's 4 waves write 4 contiguous KiB; block covers 24 KiB)
// MODE 1: far apart: store k of wave t goes to plane k (each plane = n16/NS units, covered linearly like kernel H)
template <int MODE, int NS>
__global__ void kQ(u32x4* out, size_t n16) {
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + wv;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    size_t i;
    if constexpr (MODE == 0) i = (((size_t)blockIdx.x * NS + k) * 4 + wv) * 64 + lane;
    else i = (size_t)k * (n16 / NS) + wave * 64 + lane;
    if (i < n16) __builtin_nontemporal_store(u32x4{(uint32_t)i, 1, 2, 3}, out + i);
  }
}
template <bool CONTIG>
__global__ void kB(u32x4* out, size_t n16) {  // persistent: 6 KiB tasks
  const size_t ntasks = n16 / 384, nw = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  const size_t per = (ntasks + nw - 1) / nw;
  for (size_t j = 0; j < per; j++) {
    const size_t t = CONTIG ? w * per + j : j * nw + w;
    if (t >= ntasks) break;
#pragma unroll
    for (int k = 0; k < 6; k++) __builtin_nontemporal_store(u32x4{(uint32_t)t, 1, 2, 3}, out + t * 384 + k * 64 + lane);
  }
}
template <int PER>
__global__ void k12(uint32_t* out, size_t n12) {  // 12-byte units (global_store_dwordx3), PER per lane, wave-contiguous
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const size_t i = (wave * PER + k) * 64 + lane;
    if (i < n12) { uint32_t* p = out + 3 * i; __builtin_nontemporal_store((uint32_t)i, p); __builtin_nontemporal_store(1u, p + 1); __builtin_nontemporal_store(2u, p + 2); }
  }
}
// cache-policy bits of the store (gfx942+: sc0 / sc1 / nt), 6 KiB per wave like A
#define ST_ASM(POL) asm volatile("global_store_dwordx4 %0, %1, off " POL :: "v"(p), "v"(v) : "memory")
template <int POL>
__global__ void kPol(u32x4* out, size_t n16) {
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const size_t i = wave * 384 + k * 64 + lane;
    if (i < n16) {
      u32x4* p = out + i; const u32x4 v = {(uint32_t)i, 1, 2, 3};
      if constexpr (POL == 0) ST_ASM(""); else if constexpr (POL == 1) ST_ASM("sc0"); else if constexpr (POL == 2) ST_ASM("sc1");
      else if constexpr (POL == 3) ST_ASM("sc0 sc1"); else if constexpr (POL == 4) ST_ASM("nt"); else if constexpr (POL == 5) ST_ASM("sc0 nt");
      else if constexpr (POL == 6) ST_ASM("sc1 nt"); else ST_ASM("sc0 sc1 nt");
    }
  }
}
// XCD-aware block -> range mapping: workgroup b runs on XCD b % 8; MODE 0 gives each XCD one contiguous eighth of the
// buffer, MODE 1 gives each XCD a contiguous 1-MiB region at a time (8 MiB super-tiles)
template <int MODE>
__global__ void kXcd(u32x4* out, size_t n16) {
  const size_t nb = gridDim.x, b = blockIdx.x, xcd = b & 7, j = b >> 3;
  size_t tile;
  if constexpr (MODE == 0) tile = xcd * (nb / 8) + j;
  else { const size_t per = 1048576 / (4 * 6144); tile = ((j / per) * 8 + xcd) * per + (j % per); }
  const size_t wave = tile * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 6; k++) { const size_t i = wave * 384 + k * 64 + lane; if (i < n16) __builtin_nontemporal_store(u32x4{(uint32_t)i, 1, 2, 3}, out + i); }
}
// read:write = 1:2 like NV12->RGB.  NS = stores per wave (each 1 KiB dense); the wave first loads NS x 512 B (8 B / lane).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int NS>
__global__ void kMix(const u32x2* in, u32x4* out, size_t n16) {
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  u32x2 v[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) { const size_t i = (wave * NS + k) * 64 + lane; v[k] = i < n16 ? __builtin_nontemporal_load(in + i) : u32x2{0, 0}; }
#pragma unroll
  for (int k = 0; k < NS; k++) { const size_t i = (wave * NS + k) * 64 + lane; if (i < n16) __builtin_nontemporal_store(u32x4{v[k][0], v[k][1], v[k][0] ^ 1u, v[k][1] ^ 2u}, out + i); }
}
__global__ void kD(u32x4* out, size_t n16) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
#pragma unroll
  for (int k = 0; k < 4; k++) if (i + k < n16) out[i + k] = u32x4{(uint32_t)i, 1, 2, 3};
}
int main() {
  const size_t B = (size_t)1 << 30, n16 = B / 16;
  u32x4* d; hipMalloc(&d, B); hipMemset(d, 0, B); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0, 0); for (int r = 0; r < 10; r++) launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("[wprobe] %-46s %7.0f GB/s\n", name, B * 10.0 / (ms * 1e-3) / 1e9);
  };
  run("A  6 KiB/wave, 256-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 6>), dim3((n16 / 384 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("F  6 KiB/wave, 256-thr blocks, plain", [&] { hipLaunchKernelGGL((kA<false, 6>), dim3((n16 / 384 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("E  6 KiB/wave, 1024-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 6>), dim3((n16 / 384 + 15) / 16), dim3(1024), 0, 0, d, n16); });
  run("G  16 KiB/wave, 256-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 16>), dim3((n16 / 1024 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("H  1 KiB/wave, 256-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 1>), dim3((n16 / 64 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("I  1 KiB/wave, 256-thr blocks, plain", [&] { hipLaunchKernelGGL((kA<false, 1>), dim3((n16 / 64 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("J  2 KiB/wave, 256-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 2>), dim3((n16 / 128 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("K  3 KiB/wave, 256-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 3>), dim3((n16 / 192 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("L  2 KiB/wave, 256-thr blocks, plain", [&] { hipLaunchKernelGGL((kA<false, 2>), dim3((n16 / 128 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("N  1 KiB/wave, 64-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 1>), dim3(n16 / 64), dim3(64), 0, 0, d, n16); });
  run("O  12 B/lane x1 (768 B/wave), NT", [&] { hipLaunchKernelGGL((k12<1>), dim3((B / 768 + 3) / 4), dim3(256), 0, 0, (uint32_t*)d, B / 12); });
  run("P  12 B/lane x2 (1536 B/wave), NT", [&] { hipLaunchKernelGGL((k12<2>), dim3((B / 1536 + 3) / 4), dim3(256), 0, 0, (uint32_t*)d, B / 12); });
  run("Q  6 stores/wave, block round-robin (4 KiB rounds)", [&] { hipLaunchKernelGGL((kQ<0, 6>), dim3((n16 / 384 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("R  6 stores/wave, far apart (6 linear planes)", [&] { hipLaunchKernelGGL((kQ<1, 6>), dim3((n16 / 384 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("S  3 stores/wave, far apart (3 linear planes)", [&] { hipLaunchKernelGGL((kQ<1, 3>), dim3((n16 / 192 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("T  2 stores/wave, far apart", [&] { hipLaunchKernelGGL((kQ<1, 2>), dim3((n16 / 128 + 3) / 4), dim3(256), 0, 0, d, n16); });
  run("U  1 KiB/wave, 512-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 1>), dim3((n16 / 64 + 7) / 8), dim3(512), 0, 0, d, n16); });
  run("V  1 KiB/wave, 128-thr blocks, NT", [&] { hipLaunchKernelGGL((kA<true, 1>), dim3((n16 / 64 + 1) / 2), dim3(128), 0, 0, d, n16); });
  run("B  persistent 2048 blocks, contiguous ranges", [&] { hipLaunchKernelGGL((kB<true>), dim3(2048), dim3(256), 0, 0, d, n16); });
  run("C  persistent 2048 blocks, grid-strided", [&] { hipLaunchKernelGGL((kB<false>), dim3(2048), dim3(256), 0, 0, d, n16); });
  run("D  64 B per thread (ATen-like), plain", [&] { hipLaunchKernelGGL(kD, dim3((n16 / 4 + 255) / 256), dim3(256), 0, 0, d, n16); });
  {
    const dim3 g((n16 / 384 + 3) / 4);
    run("pol \"\"          6 KiB/wave", [&] { hipLaunchKernelGGL((kPol<0>), g, dim3(256), 0, 0, d, n16); });
    run("pol sc0         6 KiB/wave", [&] { hipLaunchKernelGGL((kPol<1>), g, dim3(256), 0, 0, d, n16); });
    run("pol sc1         6 KiB/wave", [&] { hipLaunchKernelGGL((kPol<2>), g, dim3(256), 0, 0, d, n16); });
    run("pol sc0 sc1     6 KiB/wave", [&] { hipLaunchKernelGGL((kPol<3>), g, dim3(256), 0, 0, d, n16); });
    run("pol nt          6 KiB/wave", [&] { hipLaunchKernelGGL((kPol<4>), g, dim3(256), 0, 0, d, n16); });
    run("pol sc0 nt      6 KiB/wave", [&] { hipLaunchKernelGGL((kPol<5>), g, dim3(256), 0, 0, d, n16); });
    run("pol sc1 nt      6 KiB/wave", [&] { hipLaunchKernelGGL((kPol<6>), g, dim3(256), 0, 0, d, n16); });
    run("pol sc0 sc1 nt  6 KiB/wave", [&] { hipLaunchKernelGGL((kPol<7>), g, dim3(256), 0, 0, d, n16); });
    const dim3 g8(((n16 / 384 + 3) / 4) / 8 * 8);
    run("X0 XCD owns a contiguous eighth", [&] { hipLaunchKernelGGL((kXcd<0>), g8, dim3(256), 0, 0, d, n16); });
    run("X1 XCD owns 1-MiB regions", [&] { hipLaunchKernelGGL((kXcd<1>), g8, dim3(256), 0, 0, d, n16); });
  }
  {
    u32x2* src; hipMalloc(&src, B / 2); hipMemset(src, 1, B / 2); hipDeviceSynchronize();
    auto runmix = [&](const char* name, auto launch) {
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0, 0); for (int r = 0; r < 10; r++) launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("[wprobe] %-46s %7.0f GB/s (read + write)\n", name, 1.5 * B * 10.0 / (ms * 1e-3) / 1e9);
    };
    runmix("mix 1:2, 1 store/wave (512 B in, 1 KiB out)", [&] { hipLaunchKernelGGL((kMix<1>), dim3((n16 / 64 + 3) / 4), dim3(256), 0, 0, src, d, n16); });
    runmix("mix 1:2, 2 stores/wave", [&] { hipLaunchKernelGGL((kMix<2>), dim3((n16 / 128 + 3) / 4), dim3(256), 0, 0, src, d, n16); });
    runmix("mix 1:2, 3 stores/wave", [&] { hipLaunchKernelGGL((kMix<3>), dim3((n16 / 192 + 3) / 4), dim3(256), 0, 0, src, d, n16); });
    runmix("mix 1:2, 6 stores/wave", [&] { hipLaunchKernelGGL((kMix<6>), dim3((n16 / 384 + 3) / 4), dim3(256), 0, 0, src, d, n16); });
    runmix("mix 1:2, 1 store/wave, 512-thr blocks", [&] { hipLaunchKernelGGL((kMix<1>), dim3((n16 / 64 + 7) / 8), dim3(512), 0, 0, src, d, n16); });
    runmix("mix 1:2, 1 store/wave, 1024-thr blocks", [&] { hipLaunchKernelGGL((kMix<1>), dim3((n16 / 64 + 15) / 16), dim3(1024), 0, 0, src, d, n16); });
  }
  run("M  hipMemsetAsync", [&] { hipMemsetAsync(d, 1, B, 0); });
  return 0;
}
