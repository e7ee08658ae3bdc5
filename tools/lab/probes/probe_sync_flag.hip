// Probe: what does "queue a small kernel and block until it is done" cost, by mechanism?  The reference's resizer / remaper block after every
// frame (Tasks.cpp:1630-1640); the Task layer here waits on a completion flag in page-locked memory written by hipStreamWriteValue32.
//   A  kernel + hipStreamSynchronize
//   B  kernel + hipStreamWriteValue32 + spin on the flag            (what Tasks.cpp does)
//   C  kernel whose LAST workgroup writes the flag itself (release fences + arrival counter) + spin
//   D  two kernels + (B)      E  two kernels, the second as (C)      -> the chain shape conv -> resize -> wait
// Kernel: G workgroups of 256 threads, each copying 16 KB (a stand-in for a 224 x 224 resize: short, many workgroups).
// hipcc --offload-arch=gfx950 -O3 -o probe_sync_flag probe_sync_flag.hip && ./probe_sync_flag
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <immintrin.h>

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d) {
  const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++) d[i + 256 * k] = s[i + 256 * k];
}
__global__ __launch_bounds__(256) void k_copy_flag(const uint4* __restrict__ s, uint4* __restrict__ d, uint32_t* counter, volatile uint32_t* flag, uint32_t want) {
  const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++) d[i + 256 * k] = s[i + 256 * k];
  __threadfence();       // this workgroup's stores are visible device-wide before it counts as arrived
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t n = atomicAdd(counter, 1u);
    if (n == gridDim.x - 1) {
      *counter = 0;      // (the next launch on this stream starts from zero)
      __threadfence_system();
      *flag = want;
    }
  }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void spin(volatile uint32_t* f, uint32_t want) { while ((int32_t)(*f - want) < 0) _mm_pause(); }

int main() {
  const int G = 96, N = 4000;
  uint4 *s, *d, *d2; uint32_t* counter; uint32_t* flag; void* flag_dev;
  hipMalloc(&s, G * 16384); hipMalloc(&d, G * 16384); hipMalloc(&d2, G * 16384); hipMalloc(&counter, 4); hipMemset(counter, 0, 4);
  hipHostMalloc((void**)&flag, 64, hipHostMallocMapped); hipHostGetDevicePointer(&flag_dev, flag, 0); *flag = 0;
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  uint32_t seq = 0;
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, st, s, d); hipStreamSynchronize(st); }
    const double a = (now() - t0) / N;
    t0 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, st, s, d); hipStreamWriteValue32(st, flag_dev, ++seq, 0); spin(flag, seq); }
    const double b = (now() - t0) / N;
    t0 = now();
    for (int i = 0; i < N; i++) { ++seq; hipLaunchKernelGGL(k_copy_flag, dim3(G), dim3(256), 0, st, s, d, counter, (volatile uint32_t*)flag_dev, seq); spin(flag, seq); }
    const double c = (now() - t0) / N;
    t0 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, st, s, d); hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, st, d, d2); hipStreamWriteValue32(st, flag_dev, ++seq, 0); spin(flag, seq); }
    const double dd = (now() - t0) / N;
    t0 = now();
    for (int i = 0; i < N; i++) { ++seq; hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, st, s, d); hipLaunchKernelGGL(k_copy_flag, dim3(G), dim3(256), 0, st, d, d2, counter, (volatile uint32_t*)flag_dev, seq); spin(flag, seq); }
    const double e = (now() - t0) / N;
    // host cost of the calls alone (nothing waited for until the end)
    t0 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, st, s, d);
    const double l = (now() - t0) / N; hipStreamSynchronize(st);
    t0 = now();
    for (int i = 0; i < N; i++) hipStreamWriteValue32(st, flag_dev, ++seq, 0);
    const double w = (now() - t0) / N; hipStreamSynchronize(st);
    printf("[sync-flag] pass %d: A launch+hipStreamSynchronize %.2f us | B launch+WriteValue32+spin %.2f | C kernel writes the flag+spin %.2f | D 2 kernels+B %.2f | E 2 kernels, second as C %.2f | host cost: launch %.2f, WriteValue32 %.2f\n",
           rep, a, b, c, dd, e, l, w);
  }
  return 0;
}
