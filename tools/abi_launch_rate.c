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
  /* one vpf_resize per frame from C over a ring of frames past the Infinity Cache (what a per-frame caller pays without Python in the loop):
   * packed RGB 1080p -> 720p, bilinear and Lanczos-3; a 1-channel plane */
  enum { RING = 48, DW = 1280, DH = 720, SP = 3 * 1920, DP = 3 * 1280, M = 4000 };
  unsigned char *rs[RING], *rd[RING];
  for (int i = 0; i < RING; i++)
    if (hipMalloc((void**)&rs[i], (size_t)SP * H) != hipSuccess || hipMalloc((void**)&rd[i], (size_t)DP * DH) != hipSuccess) return 4;
  const vpf_size dsz = {DW, DH};
  for (int fmt = 0; fmt < 2; fmt++)
    for (int interp = 1; interp <= 2; interp++) {
      const int f = fmt ? VPF_FMT_Y : VPF_FMT_RGB;
      const unsigned sp = fmt ? 1920 : SP, dp = fmt ? 1280 : DP;
      for (int i = 0; i < 200; i++) { vpf_plane a[3] = {{rs[i % RING], sp, 0}}, b[3] = {{rd[i % RING], dp, 0}}; vpf_resize(&ex, f, interp, sz, a, dsz, b); }
      hipStreamSynchronize(st);
      const double r0 = now();
      for (int i = 0; i < M; i++) {
        vpf_plane a[3] = {{rs[i % RING], sp, 0}}, b[3] = {{rd[i % RING], dp, 0}};
        if (vpf_resize(&ex, f, interp, sz, a, dsz, b) != VPF_OK) return 5;
      }
      const double r1 = now();
      hipStreamSynchronize(st);
      const double r2 = now();
      printf("[abi] %s 1920x1080->1280x720 %s, one vpf_resize per frame from C: host issue %.2f us/call, end to end %.2f us/frame\n", fmt ? "Y  " : "RGB",
             interp == 1 ? "bilinear" : "lanczos3", (r1 - r0) / M * 1e6, (r2 - r0) / M * 1e6);
    }
  return 0;
}
