/* tools/abi_launch_rate.c — how many vpf_convert calls per second does ONE host thread issue from plain C (no Python)?
 * 1080p NV12 -> RGB_PLANAR, one frame per call, N calls back to back on one stream.  Build:
 *   gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/abi_launch_rate.c -o /tmp/abi_launch_rate \
 *       -Lvideoprocessingframework_amd -lvpfhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/videoprocessingframework_amd -Wl,-rpath,/opt/rocm/lib */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <time.h>
#include "vpf_hip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(void) {
  enum { W = 1920, H = 1080, P = 2048, N = 20000 };
  unsigned char *src, *dst;
  if (hipMalloc((void**)&src, (size_t)P * H * 3 / 2) != hipSuccess || hipMalloc((void**)&dst, (size_t)P * H * 3) != hipSuccess) return 2;
  hipStream_t st;
  hipStreamCreate(&st);
  const vpf_exec ex = {-1, 0, st};
  const vpf_size sz = {W, H};
  vpf_plane s[3] = {{src, P, 0}, {src + (size_t)H * P, P, 0}, {0, 0, 0}};
  vpf_plane d[3] = {{dst, P, 0}, {dst + (size_t)H * P, P, 0}, {dst + (size_t)2 * H * P, P, 0}};
  for (int i = 0; i < 100; i++) vpf_convert(&ex, VPF_FMT_NV12, VPF_FMT_RGB_PLANAR, VPF_BT_709, VPF_MPEG, sz, s, d);
  hipStreamSynchronize(st);
  const double t0 = now();
  for (int i = 0; i < N; i++)
    if (vpf_convert(&ex, VPF_FMT_NV12, VPF_FMT_RGB_PLANAR, VPF_BT_709, VPF_MPEG, sz, s, d) != VPF_OK) return 3;
  const double t1 = now();
  hipStreamSynchronize(st);
  const double t2 = now();
  printf("[abi] 1080p NV12->RGB_PLANAR from C: host issue %.2f us/call, end to end %.2f us/frame (%.0f Gpix/s)\n", (t1 - t0) / N * 1e6,
         (t2 - t0) / N * 1e6, (double)W * H * N / (t2 - t0) / 1e9);
  return 0;
}
