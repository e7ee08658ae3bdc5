// pmc_calib: streaming kernels with KNOWN byte counts, used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE
// on gfx950 in the access patterns our converters use (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads half of a
// wide coalesced stream; other widths and WRITE_SIZE must be calibrated on a known byte count).
//   copy16      : 16 B/lane loads, 16 B/lane stores            (p16 kernels' global pattern)
//   copy16_nt   : 16 B/lane loads, non-temporal 16 B stores
//   copy4_12    : 4 B/lane loads x3, one 12 B/lane nt store      (p4 kernels' pattern, read:write 1:1 here)
//   mix_1r2w    : reads N bytes, writes 2N bytes (the converter's 1:2 read:write mix), 16 B accesses, nt stores; every wave store is a DENSE
//                 1-KiB run (the two output halves are separate streams).  The round-1/2 form stored to out[2 i] and out[2 i + 1] — two 16-B
//                 stores at a 32-B lane stride = partial-line writes, WRITE_SIZE 1.64x, 2.7 TB/s: a strided-store probe, not a 1:2 ceiling
//   mix_1r2w_strided : that old form, kept under its real name (what lane-strided stores cost)
// Each runs over buffers far larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void copy16(const u32x4* in, u32x4* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy16_nt(const u32x4* in, u32x4* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(in[i], &out[i]);
}
__global__ void copy4_12(const uint32_t* in, uint32_t* out, size_t n3) {  // n3 = number of 12-byte groups
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t a = in[i], b = in[n3 + i], c = in[2 * n3 + i];
    __builtin_nontemporal_store(a, &out[3 * i]); __builtin_nontemporal_store(b, &out[3 * i + 1]); __builtin_nontemporal_store(c, &out[3 * i + 2]);
  }
}
__global__ void mix_1r2w(const u32x4* in, u32x4* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 v = in[i];
    __builtin_nontemporal_store(v, &out[i]); __builtin_nontemporal_store(v ^ 1u, &out[n + i]);  // lane-contiguous: two dense 1-KiB runs per wave
  }
}
__global__ void mix_1r2w_strided(const u32x4* in, u32x4* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 v = in[i];
    __builtin_nontemporal_store(v, &out[2 * i]); __builtin_nontemporal_store(v ^ 1u, &out[2 * i + 1]);  // 32-B lane stride: partial lines
  }
}
int main() {
  const size_t B = (size_t)1 << 30;  // 1 GiB read per kernel
  void *a, *b;
  if (hipMalloc(&a, B) != hipSuccess || hipMalloc(&b, 2 * B) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, B); hipMemset(b, 2, 2 * B); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double bytes) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-17s %8.1f GB/s  (%.0f MiB read+written per launch)\n", name, bytes * 5 / (ms * 1e-3) / 1e9, bytes / 1048576.0);
  };
  run("copy16", [&] { hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, B / 16); }, 2.0 * B);
  run("copy16_nt", [&] { hipLaunchKernelGGL(copy16_nt, dim3(8192), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, B / 16); }, 2.0 * B);
  run("copy4_12", [&] { hipLaunchKernelGGL(copy4_12, dim3(8192), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, B / 12); }, 2.0 * (B / 12) * 12);
  run("mix_1r2w", [&] { hipLaunchKernelGGL(mix_1r2w, dim3(8192), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, B / 16); }, 3.0 * B);
  run("mix_1r2w_strided", [&] { hipLaunchKernelGGL(mix_1r2w_strided, dim3(8192), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, B / 16); }, 3.0 * B);
  return 0;
}
