// probe_atomic_rate.hip — what a work counter costs on gfx950: N waves (all CUs busy, 4096 resident) each draw K tickets with a device-scope
// atomicAdd (one lane per wave, the wave waits for its ticket as a work-fetching wave does) from (a) ONE counter, (b) one counter per XCD
// (the wave's own), (c) one counter per wave (no sharing: the latency of the operation alone).  Prints ns per ticket as the chip sees it
// (kernel time / tickets) and per wave (kernel time / K).  hipcc --offload-arch=gfx950 -O3 -o probe_atomic_rate probe_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_tickets(uint32_t* ctr, int mode, int K, uint32_t* sink) {
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t* p = mode == 0 ? ctr : mode == 1 ? ctr + 32 * (xcc & 7u) : ctr + 32 * (8 + wave);
  uint32_t acc = 0;
  for (int k = 0; k < K; k++) {
    uint32_t t = 0;
    if ((threadIdx.x & 63u) == 0) t = atomicAdd(p, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    acc += t;
  }
  if (acc == 0xffffffffu) sink[0] = acc;
}
int main() {
  const int groups = 1024, K = 16;
  uint32_t *ctr, *sink;
  hipMalloc(&ctr, 4 * 32 * (8 + groups * 4) + 4096); hipMalloc(&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"one counter", "one counter per XCD", "one counter per wave"};
  for (int mode = 0; mode < 3; mode++) {
    hipMemset(ctr, 0, 4 * 32 * (8 + groups * 4));
    hipLaunchKernelGGL(k_tickets, dim3(groups), dim3(256), 0, 0, ctr, mode, K, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_tickets, dim3(groups), dim3(256), 0, 0, ctr, mode, K, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 5, tickets = (double)groups * 4 * K;
    printf("[atomic] %-22s: %d waves x %d tickets: kernel %.1f us = %.1f ns per ticket (chip-wide), %.2f us per ticket as a wave waits for it\n", names[mode], groups * 4, K, us, us * 1e3 / tickets, us / K);
  }
  return 0;
}
