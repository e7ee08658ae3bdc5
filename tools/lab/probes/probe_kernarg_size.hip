// probe_kernarg_size.hip — how large may a by-value kernel argument be on this runtime?  The batch entries carry their frame table in the
// kernarg segment (32 frames x 72 B); more frames per dispatch = fewer dispatch boundaries (3.4 us each against 24 us of kernel for a batch of small
// planes).  Launches kernels whose argument is a struct of 2, 4, 6, 9, 16 and 32 KB, checks the last word arrives, and times 20 launches of each.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blob { uint32_t w[N]; };
template <int N> __global__ void k_take(const Blob<N> b, uint32_t* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = b.w[N - 1] + b.w[blockIdx.x & (N - 1)]; }
template <int N> static void go(uint32_t* out) {
  Blob<N> b;
  for (int i = 0; i < N; i++) b.w[i] = 7u * (uint32_t)i;
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_take<N>, dim3(64), dim3(64), 0, 0, b, out);
  const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
  uint32_t got = 0;
  hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_take<N>, dim3(64), dim3(64), 0, 0, b, out);
  const auto t1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  const auto t2 = std::chrono::steady_clock::now();
  printf("[kernarg] %6zu bytes: launch %s, sync %s, last word %s; host issue %.2f us per launch, %.2f us per launch incl. execution\n", sizeof(b), hipGetErrorName(e1), hipGetErrorName(e2),
         got == 7u * (N - 1) ? "ok" : "WRONG", std::chrono::duration<double, std::micro>(t1 - t0).count() / 20, std::chrono::duration<double, std::micro>(t2 - t0).count() / 20);
}
int main() {
  uint32_t* out;
  hipMalloc(&out, 64);
  go<512>(out); go<1024>(out); go<1536>(out); go<2304>(out); go<4096>(out); go<8192>(out);
  return 0;
}
