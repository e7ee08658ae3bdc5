// probe: semantics of v_cvt_pk_u8_f32 on gfx950 (rounding of ties, saturation, NaN) — hardware facts the
// kernels' rounding definition depends on.  Build: hipcc --offload-arch=gfx950 tools/probe_cvt.hip -o /tmp/probe_cvt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
  float h[] = {-1e9f, -1.f, -0.75f, -0.5f, -0.25f, 0.f, 0.25f, 0.49999997f, 0.5f, 0.50000006f, 0.75f, 1.f, 1.5f, 2.5f, 3.5f, 4.5f,
               126.5f, 127.5f, 253.5f, 254.49998f, 254.5f, 254.50002f, 255.f, 255.49998f, 255.5f, 256.f, 300.f, 1e9f, INFINITY, -INFINITY, NAN,
               100.49999f, 100.5f, 100.50001f, 101.5f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* d; unsigned* o; unsigned r[64];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++) printf("cvt_pk_u8_f32(%.9g) = %u   [rint=%g]\n", h[i], r[i], std::nearbyint(h[i]));
  // exhaustive check against saturate(rint) over a dense sweep
  int bad = 0;
  const int M = 1 << 20;
  float* hs = new float[M]; unsigned* rs = new unsigned[M];
  for (int i = 0; i < M; i++) hs[i] = -8.f + (float)i * (272.f / M);
  float* d2; unsigned* o2; hipMalloc(&d2, M * 4); hipMalloc(&o2, M * 4);
  hipMemcpy(d2, hs, M * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(M / 256), dim3(256), 0, 0, d2, o2, M);
  hipMemcpy(rs, o2, M * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < M; i++) {
    float t = std::nearbyint(hs[i]); t = t < 0 ? 0 : (t > 255 ? 255 : t);
    if ((unsigned)t != rs[i]) { if (bad < 5) printf("MISMATCH %.9g hw=%u rne=%g\n", hs[i], rs[i], t); bad++; }
  }
  printf("dense sweep vs saturate(rint-to-nearest-even): %d mismatches of %d\n", bad, M);
  return 0;
}
