/* tests/c/abi_smoke.c — a plain C99 client of include/vpf_hip.h (no Python, no C++, no torch): what a host written in
 * any language with a C FFI does.  Usage: abi_smoke Y U V  ->  prints "R G B" of an NV12 -> RGB (BT.709 limited range)
 * conversion of a constant 64 x 16 frame, then a 2x fused convert+resize of the same frame and the batch entries (convert -> resize -> remap
 * of two frames), "ok" on success.
 * Compiled and run by tests/test_gpu_parity.py::test_plain_c_client_of_the_abi. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vpf_hip.h"

#define CHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
  if (argc != 4) return 64;
  const unsigned char yv = (unsigned char)atoi(argv[1]), uv = (unsigned char)atoi(argv[2]), vv = (unsigned char)atoi(argv[3]);
  enum { W = 64, H = 16, P = 256, PD = 256 };
  if (vpf_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 3; }
  unsigned char *src = NULL, *dst = NULL, *dst2 = NULL, host[H * 3 / 2 * P], out[H * PD];
  for (int r = 0; r < H; r++) memset(host + r * P, yv, P);
  for (int r = 0; r < H / 2; r++)
    for (int x = 0; x < P; x += 2) { host[(H + r) * P + x] = uv; host[(H + r) * P + x + 1] = vv; }
  CHECK(hipMalloc((void**)&src, sizeof host));
  CHECK(hipMalloc((void**)&dst, sizeof out));
  CHECK(hipMalloc((void**)&dst2, sizeof out));
  CHECK(hipMemcpy(src, host, sizeof host, hipMemcpyHostToDevice));
  CHECK(hipMemset(dst, 0, sizeof out));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  const vpf_exec ex = {-1, 0, st};
  const vpf_size sz = {W, H}, half = {W / 2, H / 2};
  vpf_plane s[3] = {{src, P, 0}, {src + (size_t)H * P, P, 0}, {0, 0, 0}}, d[3] = {{dst, PD, 0}, {0, 0, 0}, {0, 0, 0}};
  vpf_plane d2[3] = {{dst2, PD, 0}, {0, 0, 0}, {0, 0, 0}};
  if (!vpf_convert_supported(VPF_FMT_NV12, VPF_FMT_RGB, VPF_BT_709, VPF_MPEG)) return 4;
  vpf_status rc = vpf_convert(&ex, VPF_FMT_NV12, VPF_FMT_RGB, VPF_BT_709, VPF_MPEG, sz, s, d);
  if (rc != VPF_OK) { fprintf(stderr, "vpf_convert: %s\n", vpf_status_string(rc)); return 5; }
  rc = vpf_convert_resize(&ex, VPF_FMT_NV12, VPF_FMT_RGB, VPF_BT_709, VPF_MPEG, sz, s, half, d2);
  if (rc != VPF_OK) { fprintf(stderr, "vpf_convert_resize: %s\n", vpf_status_string(rc)); return 6; }
  CHECK(hipStreamSynchronize(st));
  CHECK(hipMemcpy(out, dst, sizeof out, hipMemcpyDeviceToHost));
  for (int r = 0; r < H; r++)
    for (int x = 0; x < 3 * W; x++)
      if (out[r * PD + x] != out[x % 3]) { fprintf(stderr, "frame is not constant at (%d,%d)\n", x, r); return 7; }
  const unsigned char rgb[3] = {out[0], out[1], out[2]};
  CHECK(hipMemcpy(out, dst2, sizeof out, hipMemcpyDeviceToHost));
  for (int r = 0; r < H / 2; r++)
    for (int x = 0; x < 3 * W / 2; x++)
      if (out[r * PD + x] != rgb[x % 3]) { fprintf(stderr, "resized frame differs at (%d,%d)\n", x, r); return 8; }
  /* the batch entries: two frames (the same source twice) converted, resized to half size with both filters (a constant picture stays
   * constant: every filter's weights sum to one) and remapped through an identity map, each in one call */
  {
    unsigned char *b1 = NULL, *b2 = NULL, *b3 = NULL;
    float *mx = NULL, *my = NULL, hx[H / 2][W / 2], hy[H / 2][W / 2];
    CHECK(hipMalloc((void**)&b1, 2 * sizeof out)); CHECK(hipMalloc((void**)&b2, 2 * sizeof out)); CHECK(hipMalloc((void**)&b3, 2 * sizeof out));
    CHECK(hipMalloc((void**)&mx, sizeof hx)); CHECK(hipMalloc((void**)&my, sizeof hy));
    for (int r = 0; r < H / 2; r++) for (int x = 0; x < W / 2; x++) { hx[r][x] = (float)x; hy[r][x] = (float)r; }
    CHECK(hipMemcpy(mx, hx, sizeof hx, hipMemcpyHostToDevice)); CHECK(hipMemcpy(my, hy, sizeof hy, hipMemcpyHostToDevice));
    CHECK(hipMemset(b3, 0, 2 * sizeof out));
    vpf_frame_io cv[2], rs[2], rm[2];
    memset(cv, 0, sizeof cv); memset(rs, 0, sizeof rs); memset(rm, 0, sizeof rm);
    for (int i = 0; i < 2; i++) {
      cv[i].src[0] = s[0]; cv[i].src[1] = s[1];
      cv[i].dst[0].ptr = b1 + i * sizeof out; cv[i].dst[0].pitch = PD;
      rs[i].src[0] = cv[i].dst[0]; rs[i].dst[0].ptr = b2 + i * sizeof out; rs[i].dst[0].pitch = PD;
      rm[i].src[0] = rs[i].dst[0]; rm[i].dst[0].ptr = b3 + i * sizeof out; rm[i].dst[0].pitch = PD;
    }
    if (vpf_convert_batch(&ex, VPF_FMT_NV12, VPF_FMT_RGB, VPF_BT_709, VPF_MPEG, sz, 2, cv) != VPF_OK) return 11;
    for (int interp = VPF_INTERP_LINEAR; interp <= VPF_INTERP_LANCZOS3; interp++) {
      if (vpf_resize_batch(&ex, VPF_FMT_RGB, interp, sz, half, 2, rs) != VPF_OK) return 12;
      if (vpf_remap_batch(&ex, VPF_FMT_RGB, half, mx, (W / 2) * 4, my, (W / 2) * 4, half, 2, rm) != VPF_OK) return 13;
      CHECK(hipStreamSynchronize(st));
      for (int i = 0; i < 2; i++) {
        CHECK(hipMemcpy(out, b3 + i * sizeof out, sizeof out, hipMemcpyDeviceToHost));
        for (int r = 0; r < H / 2; r++)
          for (int x = 0; x < 3 * W / 2; x++)
            if (out[r * PD + x] != rgb[x % 3]) { fprintf(stderr, "batch chain (filter %d) differs at frame %d (%d,%d)\n", interp, i, x, r); return 14; }
      }
    }
    if (vpf_resize_batch(&ex, VPF_FMT_RGB, VPF_INTERP_LINEAR, sz, half, 0, rs) != VPF_ERR_BAD_ARG) return 15;
    hipFree(b1); hipFree(b2); hipFree(b3); hipFree(mx); hipFree(my);
  }
  /* an unsupported pair and a bad argument come back as status codes, never as a crash */
  if (vpf_convert(&ex, VPF_FMT_NV12, VPF_FMT_YUV444, VPF_BT_709, VPF_MPEG, sz, s, d) != VPF_ERR_UNSUPPORTED) return 9;
  if (vpf_convert(&ex, VPF_FMT_NV12, VPF_FMT_RGB, VPF_BT_709, VPF_MPEG, sz, NULL, d) != VPF_ERR_BAD_ARG) return 10;
  printf("%d %d %d\n%s\nok\n", rgb[0], rgb[1], rgb[2], vpf_version());
  hipFree(src); hipFree(dst); hipFree(dst2); hipStreamDestroy(st);
  return 0;
}
