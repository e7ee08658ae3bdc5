/*
 * vpf_oracle.c — CPU ORACLE (test infrastructure only; see vpf_oracle.h for the rules and the
 * "PARITY UNPINNED" statement).  Plain C, no dependency on the product.
 *
 * Every function cites the reference lines (relative to /root/reference) whose behaviour it
 * restates.  The arithmetic itself is NOT in the reference tree (closed-source NPP); formulas are
 * the ones published in NPP's colour-conversion documentation and the BT.601/BT.709 matrices
 * (assumption register A1-A10).
 */
#include "vpf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* reference Pixel_Format values: src/TC/inc/MemoryInterfaces.hpp:30-49 */
enum {
  F_UNDEFINED = 0, F_Y = 1, F_RGB = 2, F_NV12 = 3, F_YUV420 = 4, F_RGB_PLANAR = 5, F_BGR = 6,
  F_YCBCR = 7, F_YUV444 = 8, F_RGB_32F = 9, F_RGB_32F_PLANAR = 10, F_YUV422 = 11, F_P10 = 12,
  F_P12 = 13
};
enum { CS_601 = 0, CS_709 = 1 };
enum { CR_MPEG = 0, CR_JPEG = 1 };

static int g_threads = 1;
int vpfo_set_threads(int n) {
  int p = g_threads;
  if (n >= 1) g_threads = n;
  return p;
}
/* tear the OpenMP thread pool down (threads left over from a larger team would otherwise idle-spin next to a smaller one) */
void vpfo_release_threads(void) {
#ifdef _OPENMP
  omp_pause_resource_all(omp_pause_soft);
#endif
}
/* Assumption switches: what NPP does where its documentation is silent and a wrong guess
 * moves results by far more than 1 LSB.  Defaults are the conventions the HIP kernels implement.  Non-default values
 * change VPFO_EXACT only (FP32 restates the kernels and returns VPFO_UNSUPPORTED under a non-default switch), so that a
 * mismatch on first contact with real NPP output (tests/test_reference_fixtures.py) is settled by flipping a switch. */
static int g_a2 = 0, g_a6 = 0, g_a8 = 0, g_a10 = 0;
int vpfo_set_assumption(int key, int value) {
  int* g = key == VPFO_A2_CHROMA_UPSAMPLE ? &g_a2 : key == VPFO_A6_CHROMA_DECIMATE ? &g_a6 : key == VPFO_A8_RESIZE_COORDS ? &g_a8
           : key == VPFO_A10_LANCZOS_MINIFY ? &g_a10 : 0;
  const int hi = key == VPFO_A6_CHROMA_DECIMATE || key == VPFO_A10_LANCZOS_MINIFY ? 1 : 2;
  if (!g || value < 0 || value > hi) return -1;
  const int prev = *g;
  *g = value;
  return prev;
}
int vpfo_get_assumption(int key) {
  return key == VPFO_A2_CHROMA_UPSAMPLE ? g_a2 : key == VPFO_A6_CHROMA_DECIMATE ? g_a6 : key == VPFO_A8_RESIZE_COORDS ? g_a8
         : key == VPFO_A10_LANCZOS_MINIFY ? g_a10 : -1;
}
const char* vpfo_version(void) { return "vpf-oracle 1 (parity unpinned: NPP closed source)"; }

/* Hot loops are instantiated twice from one body: baseline x86-64 (fmaf = libm call, correct everywhere) and
 * AVX2+FMA (vectorised vfmadd), chosen at run time by CPU FEATURE.  (target_clones("arch=haswell") would dispatch
 * on the CPU *model* and never fire on EPYC / newer Xeons.)  Results are bit-identical: fmaf is exact either way. */
#define VPFO_MULTIVERSION
#define VPFO_TARGET_AVX2 __attribute__((target("avx2,fma")))
static int has_avx2_fma(void) { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); }

/* ------------------------------------------------------------------------------------------
 * YUV -> RGB matrices.  Decimal coefficients x 1e6, exactly as printed in SURVEY.md §8c:
 *   709-MPEG  nppiNV12ToRGB_709CSC   (TasksColorCvt.cpp:148)  BT.709, 219/224 scaling      [A1]
 *   709-JPEG  nppiNV12ToRGB_709HDTV  (TasksColorCvt.cpp:145)  BT.709 full range            [A1]
 *   601-JPEG  nppiNV12ToRGB / nppiYUV420ToRGB / nppiYUVToRGB (:154,:351,:536) NPP "YUV"    [A4]
 *   601-MPEG  nppiYCbCr420ToRGB / nppiYCbCrToBGR (:354,:473)  NPP "YCbCr"                  [A4]
 * R = cy*(Y-off) + rv*(V-128);  G = cy*(Y-off) + gu*(U-128) + gv*(V-128);  B = cy*(Y-off) + bu*(U-128)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int64_t cy, rv, gu, gv, bu; /* x 1e6 */
  int off;
} yuv2rgb_dec;

static const yuv2rgb_dec k_yuv2rgb[2][2] = {
    /* [color_space][color_range] */
    {/* 601 */ {1164000, 1596000, -392000, -813000, 2017000, 16} /* MPEG */,
     {1000000, 1140000, -394000, -581000, 2032000, 0} /* JPEG */},
    {/* 709 */ {1164384, 1792741, -213249, -532909, 2112402, 16} /* MPEG */,
     {1000000, 1574800, -187324, -468124, 1855600, 0} /* JPEG */}};

static inline int64_t floordiv(int64_t a, int64_t b) { /* b > 0 */
  int64_t q = a / b, r = a % b;
  return (r < 0) ? q - 1 : q;
}
static inline uint8_t clamp_u8(int64_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
/* exact round-half-up of num/1e6, clamped */
static inline uint8_t round6(int64_t num) { return clamp_u8(floordiv(num + 500000, 1000000)); }

static inline void yuv2rgb_exact(const yuv2rgb_dec* m, int y, int u, int v, uint8_t* r, uint8_t* g,
                                 uint8_t* b) {
  int64_t yy = m->cy * (y - m->off), uu = u - 128, vv = v - 128;
  *r = round6(yy + m->rv * vv);
  *g = round6(yy + m->gu * uu + m->gv * vv);
  *b = round6(yy + m->bu * uu);
}

/* the same with chroma given in sixteenths (interpolated chroma, A2 != 0): exact, round half up */
static inline void yuv2rgb_exact_q16(const yuv2rgb_dec* m, int y, int u16, int v16, uint8_t* r, uint8_t* g, uint8_t* b) {
  const int64_t yy = 16 * m->cy * (y - m->off), uu = u16 - 128 * 16, vv = v16 - 128 * 16;
  *r = clamp_u8(floordiv(yy + m->rv * vv + 8000000, 16000000));
  *g = clamp_u8(floordiv(yy + m->gu * uu + m->gv * vv + 8000000, 16000000));
  *b = clamp_u8(floordiv(yy + m->bu * uu + 8000000, 16000000));
}

/* fp32 restatement of the HIP kernels' operation order (csrc/vpf_device.h chroma_terms / sat_rne,
 * csrc/vpf_abi.hip make_yuv2rgb): biases fold -off*cy and -128*coef; three fused multiply-adds for
 * chroma, one per channel for luma, then saturate to [0,255] with round-to-nearest-EVEN — the
 * semantics of the one instruction the kernels use (v_cvt_pk_u8_f32, measured on gfx950 by
 * tools/probe_cvt.hip, profiles/r01_probe_cvt_pk_u8_f32.txt). */
typedef struct {
  float cy, rv, gu, gv, bu, br, bg, bb;
} yuv2rgb_f32;

static yuv2rgb_f32 make_f32(const yuv2rgb_dec* m) {
  yuv2rgb_f32 c;
  c.cy = (float)((double)m->cy / 1e6);
  c.rv = (float)((double)m->rv / 1e6);
  c.gu = (float)((double)m->gu / 1e6);
  c.gv = (float)((double)m->gv / 1e6);
  c.bu = (float)((double)m->bu / 1e6);
  /* exact integers / 1e6: one correctly rounded division, one correctly rounded narrowing */
  c.br = (float)((double)(-(int64_t)m->off * m->cy - 128 * m->rv) / 1e6);
  c.bg = (float)((double)(-(int64_t)m->off * m->cy - 128 * (m->gu + m->gv)) / 1e6);
  c.bb = (float)((double)(-(int64_t)m->off * m->cy - 128 * m->bu) / 1e6);
  return c;
}
static inline uint8_t sat_trunc(float t) {
  t = t < 0.f ? 0.f : t;
  t = t > 255.f ? 255.f : t;
  return (uint8_t)(int)t;
}
/* saturate + round to nearest even (default FP environment) */
static inline uint8_t sat_rne(float t) {
  t = t < 0.f ? 0.f : t;
  t = t > 255.f ? 255.f : t;
  return (uint8_t)(int)nearbyintf(t);
}
static inline void yuv2rgb_fp32(const yuv2rgb_f32* c, int y, int u, int v, uint8_t* r, uint8_t* g,
                                uint8_t* b) {
  float yf = (float)y, uf = (float)u, vf = (float)v;
  float rc = __builtin_fmaf(vf, c->rv, c->br);
  float gc = __builtin_fmaf(uf, c->gu, __builtin_fmaf(vf, c->gv, c->bg));
  float bc = __builtin_fmaf(uf, c->bu, c->bb);
  *r = sat_rne(__builtin_fmaf(yf, c->cy, rc));
  *g = sat_rne(__builtin_fmaf(yf, c->cy, gc));
  *b = sat_rne(__builtin_fmaf(yf, c->cy, bc));
}

static int valid_cscr(int cs, int cr) { return (cs == CS_601 || cs == CS_709) && (cr == CR_MPEG || cr == CR_JPEG); }

int vpfo_yuv2rgb_px(int mode, int cs, int cr, int y, int u, int v, uint8_t rgb[3]) {
  if (!valid_cscr(cs, cr)) return VPFO_UNSUPPORTED;
  const yuv2rgb_dec* m = &k_yuv2rgb[cs][cr];
  if (mode == VPFO_EXACT) {
    yuv2rgb_exact(m, y, u, v, &rgb[0], &rgb[1], &rgb[2]);
  } else {
    yuv2rgb_f32 c = make_f32(m);
    yuv2rgb_fp32(&c, y, u, v, &rgb[0], &rgb[1], &rgb[2]);
  }
  return VPFO_OK;
}

VPFO_MULTIVERSION
static int exhaustive_yuv2rgb_impl(const yuv2rgb_dec* m, uint64_t* n_diff) {
  yuv2rgb_f32 c = make_f32(m);
  int maxd = 0;
  uint64_t nd = 0;
  for (int y = 0; y < 256; y++)
    for (int u = 0; u < 256; u++)
      for (int v = 0; v < 256; v++) {
        uint8_t a[3], b[3];
        yuv2rgb_exact(m, y, u, v, &a[0], &a[1], &a[2]);
        yuv2rgb_fp32(&c, y, u, v, &b[0], &b[1], &b[2]);
        int d = 0;
        for (int k = 0; k < 3; k++) {
          int e = abs((int)a[k] - (int)b[k]);
          if (e > d) d = e;
        }
        if (d) nd++;
        if (d > maxd) maxd = d;
      }
  *n_diff = nd;
  return maxd;
}
int vpfo_yuv2rgb_exhaustive(int cs, int cr, uint64_t* n_diff) {
  if (!valid_cscr(cs, cr)) return -1;
  return exhaustive_yuv2rgb_impl(&k_yuv2rgb[cs][cr], n_diff);
}

/* ------------------------------------------------------------------------------------------
 * RGB -> YUV (BT.601 only, as in the reference: TasksColorCvt.cpp:636-639,736-739,791-794,893-896)
 *   JPEG -> NPP "YUV":   Y=.299R+.587G+.114B; U=.492(B-Y)+128; V=.877(R-Y)+128     (:658,:754,:811,:912) [A5]
 *   MPEG -> NPP "YCbCr": Y=.257R+.504G+.098B+16; Cb=-.148R-.291G+.439B+128; Cr=.439R-.368G-.071B+128
 *                                                                                 (:655,:709,:815,:916) [A5]
 * Expressed as  out_k = (a_k*R + b_k*G + c_k*B)/1e6 + d_k  with exact integer coefficients.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int64_t m[3][3]; /* x 1e6 */
  int d[3];
} rgb2yuv_dec;
static const rgb2yuv_dec k_rgb2yuv[2] = {
    /* MPEG / YCbCr */
    {{{257000, 504000, 98000}, {-148000, -291000, 439000}, {439000, -368000, -71000}}, {16, 128, 128}},
    /* JPEG / YUV: U = .492*(B-Y) -> -.147108R -.288804G +.435912B ; V = .877*(R-Y) -> .614777R -.514799G -.099978B */
    {{{299000, 587000, 114000}, {-147108, -288804, 435912}, {614777, -514799, -99978}}, {0, 128, 128}}};

typedef struct {
  float m[3][3];
  float d[3]; /* d + 0.5 */
} rgb2yuv_f32;
static rgb2yuv_f32 make_rgb2yuv_f32(const rgb2yuv_dec* m) {
  rgb2yuv_f32 c;
  for (int k = 0; k < 3; k++) {
    for (int j = 0; j < 3; j++) c.m[k][j] = (float)((double)m->m[k][j] / 1e6);
    c.d[k] = (float)((double)((int64_t)m->d[k] * 1000000 + 500000) / 1e6);
  }
  return c;
}
static inline int64_t rgb2yuv_num(const rgb2yuv_dec* m, int k, int r, int g, int b) {
  return m->m[k][0] * r + m->m[k][1] * g + m->m[k][2] * b + (int64_t)m->d[k] * 1000000;
}
static inline float rgb2yuv_f(const rgb2yuv_f32* c, int k, float r, float g, float b) {
  /* kernel op order: fma(r, m0, fma(g, m1, fma(b, m2, d+0.5))) */
  return __builtin_fmaf(r, c->m[k][0], __builtin_fmaf(g, c->m[k][1], __builtin_fmaf(b, c->m[k][2], c->d[k])));
}
int vpfo_rgb2yuv_px(int mode, int cr, int r, int g, int b, uint8_t yuv[3]) {
  if (cr != CR_MPEG && cr != CR_JPEG) return VPFO_UNSUPPORTED;
  const rgb2yuv_dec* m = &k_rgb2yuv[cr];
  if (mode == VPFO_EXACT) {
    for (int k = 0; k < 3; k++) yuv[k] = round6(rgb2yuv_num(m, k, r, g, b));
  } else {
    rgb2yuv_f32 c = make_rgb2yuv_f32(m);
    for (int k = 0; k < 3; k++) yuv[k] = sat_trunc(rgb2yuv_f(&c, k, (float)r, (float)g, (float)b));
  }
  return VPFO_OK;
}
VPFO_MULTIVERSION
static int exhaustive_rgb2yuv_impl(const rgb2yuv_dec* m, uint64_t* n_diff) {
  rgb2yuv_f32 c = make_rgb2yuv_f32(m);
  int maxd = 0;
  uint64_t nd = 0;
  for (int r = 0; r < 256; r++)
    for (int g = 0; g < 256; g++)
      for (int b = 0; b < 256; b++) {
        int d = 0;
        for (int k = 0; k < 3; k++) {
          int e = abs((int)round6(rgb2yuv_num(m, k, r, g, b)) -
                      (int)sat_trunc(rgb2yuv_f(&c, k, (float)r, (float)g, (float)b)));
          if (e > d) d = e;
        }
        if (d) nd++;
        if (d > maxd) maxd = d;
      }
  *n_diff = nd;
  return maxd;
}
int vpfo_rgb2yuv_exhaustive(int cr, uint64_t* n_diff) {
  if (cr != CR_MPEG && cr != CR_JPEG) return -1;
  return exhaustive_rgb2yuv_impl(&k_rgb2yuv[cr], n_diff);
}

/* ------------------------------------------------------------------------------------------
 * helpers
 * ------------------------------------------------------------------------------------------ */
static inline uint8_t* prow(const vpfo_plane* p, uint32_t y) { return (uint8_t*)p->ptr + (size_t)y * p->pitch; }
static inline uint32_t cdiv2(uint32_t v) { return (v + 1) >> 1; }

static int is_yuv_src(int f) { return f == F_NV12 || f == F_YUV420 || f == F_YUV444 || f == F_YCBCR; }
static int is_rgb3(int f) { return f == F_RGB || f == F_BGR || f == F_RGB_PLANAR; }
static int nplanes(int f) {
  switch (f) {
    case F_Y: case F_RGB: case F_BGR: case F_RGB_32F: return 1;
    case F_NV12: case F_P10: case F_P12: return 2;
    case F_YUV420: case F_YCBCR: case F_YUV444: case F_RGB_PLANAR: case F_RGB_32F_PLANAR: return 3;
    default: return 0;
  }
}
/* bytes per row of plane k for width w */
static uint32_t row_bytes(int f, int k, uint32_t w) {
  switch (f) {
    case F_Y: return w;
    case F_RGB: case F_BGR: return 3 * w;
    case F_NV12: return k == 0 ? w : 2 * cdiv2(w);
    case F_YUV420: case F_YCBCR: return k == 0 ? w : cdiv2(w);
    case F_YUV444: case F_RGB_PLANAR: return w;
    case F_RGB_32F: return 12 * w;
    case F_RGB_32F_PLANAR: return 4 * w;
    case F_P10: case F_P12: return k == 0 ? 2 * w : 4 * cdiv2(w);
    default: return 0;
  }
}
static int check_planes(int f, uint32_t w, const vpfo_plane* p) {
  int n = nplanes(f);
  if (!n || !p) return 0;
  for (int k = 0; k < n; k++)
    if (!p[k].ptr || p[k].pitch < row_bytes(f, k, w)) return 0;
  return 1;
}

/* fetch (Y,U,V) of pixel (x,y) from any YUV source; chroma replicated over the 2x2 quad [A2] */
static inline void fetch_yuv(int f, const vpfo_plane* s, uint32_t x, uint32_t y, int* Y, int* U, int* V) {
  *Y = prow(&s[0], y)[x];
  switch (f) {
    case F_NV12: {
      const uint8_t* uv = prow(&s[1], y >> 1) + 2 * (x >> 1);
      *U = uv[0];
      *V = uv[1];
    } break;
    case F_YUV420: case F_YCBCR:
      *U = prow(&s[1], y >> 1)[x >> 1];
      *V = prow(&s[2], y >> 1)[x >> 1];
      break;
    default: /* YUV444 */
      *U = prow(&s[1], y)[x];
      *V = prow(&s[2], y)[x];
  }
}
/* A2 alternatives: 4:2:0 chroma interpolated bilinearly to luma resolution, in sixteenths, edges clamped.
 *   siting 1 (centred, JPEG / MPEG-1): the chroma sample sits in the middle of its 2x2 luma quad: weights 3/4, 1/4 per axis
 *   siting 2 (left, MPEG-2 / H.264 default): co-sited with even luma columns horizontally (1 or 1/2, 1/2), centred vertically */
static inline int chroma_at(int f, const vpfo_plane* s, int k /*0=U,1=V*/, int32_t cx, int32_t cy, uint32_t cw, uint32_t ch) {
  cx = cx < 0 ? 0 : (cx > (int32_t)cw - 1 ? (int32_t)cw - 1 : cx);
  cy = cy < 0 ? 0 : (cy > (int32_t)ch - 1 ? (int32_t)ch - 1 : cy);
  if (f == F_NV12) return prow(&s[1], (uint32_t)cy)[2 * cx + k];
  return prow(&s[1 + k], (uint32_t)cy)[cx];
}
static inline void fetch_chroma_q16(int f, int siting, const vpfo_plane* s, uint32_t x, uint32_t y, uint32_t w, uint32_t h, int* U16, int* V16) {
  const uint32_t cw = cdiv2(w), ch = cdiv2(h);
  int32_t x0, y0;
  int wx0, wx1, wy0, wy1; /* quarters */
  if (siting == 1) { if (x & 1) { x0 = (int32_t)(x >> 1); wx0 = 3; wx1 = 1; } else { x0 = (int32_t)(x >> 1) - 1; wx0 = 1; wx1 = 3; } }
  else { x0 = (int32_t)(x >> 1); if (x & 1) { wx0 = 2; wx1 = 2; } else { wx0 = 4; wx1 = 0; } }
  if (y & 1) { y0 = (int32_t)(y >> 1); wy0 = 3; wy1 = 1; } else { y0 = (int32_t)(y >> 1) - 1; wy0 = 1; wy1 = 3; }
  int acc[2];
  for (int k = 0; k < 2; k++)
    acc[k] = wy0 * (wx0 * chroma_at(f, s, k, x0, y0, cw, ch) + wx1 * chroma_at(f, s, k, x0 + 1, y0, cw, ch)) +
             wy1 * (wx0 * chroma_at(f, s, k, x0, y0 + 1, cw, ch) + wx1 * chroma_at(f, s, k, x0 + 1, y0 + 1, cw, ch));
  *U16 = acc[0]; *V16 = acc[1];
}
static inline void store_rgb(int f, const vpfo_plane* d, uint32_t x, uint32_t y, uint8_t r, uint8_t g, uint8_t b) {
  switch (f) {
    case F_RGB: { uint8_t* p = prow(&d[0], y) + 3 * x; p[0] = r; p[1] = g; p[2] = b; } break;
    case F_BGR: { uint8_t* p = prow(&d[0], y) + 3 * x; p[0] = b; p[1] = g; p[2] = r; } break;
    default: prow(&d[0], y)[x] = r; prow(&d[1], y)[x] = g; prow(&d[2], y)[x] = b;
  }
}
static inline void fetch_rgb(int f, const vpfo_plane* s, uint32_t x, uint32_t y, int* r, int* g, int* b) {
  switch (f) {
    case F_RGB: { const uint8_t* p = prow(&s[0], y) + 3 * x; *r = p[0]; *g = p[1]; *b = p[2]; } break;
    case F_BGR: { const uint8_t* p = prow(&s[0], y) + 3 * x; *b = p[0]; *g = p[1]; *r = p[2]; } break;
    default: *r = prow(&s[0], y)[x]; *g = prow(&s[1], y)[x]; *b = prow(&s[2], y)[x];
  }
}

/* ------------------------------------------------------------------------------------------
 * YUV -> RGB family:  C1 nv12_rgb (:122-182), C2 nv12_bgr (:53-108), C6 yuv420_rgb (:322-369),
 * C7 yuv420_bgr (:383-430), C8 yuv444_bgr (:444-490), C9 yuv444_rgb (:504-550),
 * C10 yuv444_rgb_planar (:564-612), and the fused NV12/YUV420 -> RGB_PLANAR (C1', = C1 then C18).
 * ------------------------------------------------------------------------------------------ */
VPFO_MULTIVERSION
static void yuv_to_rgb(int mode, int sf, int df, const yuv2rgb_dec* m, uint32_t w, uint32_t h,
                       const vpfo_plane* s, const vpfo_plane* d) {
  yuv2rgb_f32 c = make_f32(m);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t yy = 0; yy < (int64_t)h; yy++) {
    uint32_t y = (uint32_t)yy;
    for (uint32_t x = 0; x < w; x++) {
      int Y, U, V;
      uint8_t r, g, b;
      fetch_yuv(sf, s, x, y, &Y, &U, &V);
      if (mode == VPFO_EXACT && g_a2 && (sf == F_NV12 || sf == F_YUV420 || sf == F_YCBCR)) {
        int U16, V16;
        fetch_chroma_q16(sf, g_a2, s, x, y, w, h, &U16, &V16);
        yuv2rgb_exact_q16(m, Y, U16, V16, &r, &g, &b);
      } else if (mode == VPFO_EXACT) yuv2rgb_exact(m, Y, U, V, &r, &g, &b);
      else yuv2rgb_fp32(&c, Y, U, V, &r, &g, &b);
      store_rgb(df, d, x, y, r, g, b);
    }
  }
}

/* fast row-specialised NV12 -> packed RGB/BGR, FP32 mode; used for the cpu_baseline timing so the CPU number is
 * an honest vectorised, multi-threaded port rather than the generic per-pixel switch.  Same operation order as
 * yuv2rgb_fp32 (bit-identical results; tests compare the two).  Stage 1 (vectorisable): per row, chroma terms per
 * pair and three planar u8 channel rows; stage 2: interleave to packed 3 B/px. */
static inline uint8_t rne_u8(float t) {
  union { float f; uint32_t u; } m;
  m.f = t + 12582912.0f;
  return (uint8_t)(m.u & 0xffu);
}
/* clamp to [0,255] then round to nearest even WITHOUT a float->int instruction: adding 1.5*2^23 leaves the
 * RNE-rounded integer in the low mantissa bits (ulp is 1 there; default rounding mode).  Vectorises to add+and. */
#define VPFO_CLAMP_RNE(t) rne_u8((t) < 0.f ? 0.f : ((t) > 255.f ? 255.f : (t)))
/* one chroma pair + its two luma samples per iteration; interleaved groups of 2, so the loop vectorises */
#define VPFO_DEFINE_ROW_FAST(NAME, ATTR)                                                                          \
  ATTR static void NAME(const yuv2rgb_f32* c, const uint8_t* restrict yr, const uint8_t* restrict uvr, uint32_t w, \
                        uint8_t* restrict r8, uint8_t* restrict g8, uint8_t* restrict b8) {                        \
    const uint32_t np = w >> 1;                                                                                    \
    for (uint32_t p = 0; p < np; p++) {                                                                            \
      const float uf = (float)uvr[2 * p], vf = (float)uvr[2 * p + 1];                                              \
      const float rc = __builtin_fmaf(vf, c->rv, c->br);                                                           \
      const float gc = __builtin_fmaf(uf, c->gu, __builtin_fmaf(vf, c->gv, c->bg));                                \
      const float bc = __builtin_fmaf(uf, c->bu, c->bb);                                                           \
      const float y0 = (float)yr[2 * p], y1 = (float)yr[2 * p + 1];                                                \
      r8[2 * p] = VPFO_CLAMP_RNE(__builtin_fmaf(y0, c->cy, rc));                                                   \
      r8[2 * p + 1] = VPFO_CLAMP_RNE(__builtin_fmaf(y1, c->cy, rc));                                               \
      g8[2 * p] = VPFO_CLAMP_RNE(__builtin_fmaf(y0, c->cy, gc));                                                   \
      g8[2 * p + 1] = VPFO_CLAMP_RNE(__builtin_fmaf(y1, c->cy, gc));                                               \
      b8[2 * p] = VPFO_CLAMP_RNE(__builtin_fmaf(y0, c->cy, bc));                                                   \
      b8[2 * p + 1] = VPFO_CLAMP_RNE(__builtin_fmaf(y1, c->cy, bc));                                               \
    }                                                                                                              \
    if (w & 1) {                                                                                                   \
      const uint32_t x = w - 1;                                                                                    \
      yuv2rgb_fp32(c, yr[x], uvr[x], uvr[x + 1], &r8[x], &g8[x], &b8[x]);                                          \
    }                                                                                                              \
  }
VPFO_DEFINE_ROW_FAST(nv12_row_fast_base, )
VPFO_DEFINE_ROW_FAST(nv12_row_fast_avx2, VPFO_TARGET_AVX2)
static void nv12_to_rgb_fast(int bgr, const yuv2rgb_dec* m, uint32_t w, uint32_t h, const vpfo_plane* s,
                             const vpfo_plane* d) {
  const yuv2rgb_f32 c = make_f32(m);
  const int fast = has_avx2_fma();
#pragma omp parallel num_threads(g_threads)
  {
    uint8_t* tmp = (uint8_t*)malloc(3 * (size_t)w + 64);
    uint8_t *r8 = tmp, *g8 = tmp + w, *b8 = tmp + 2 * (size_t)w;
#pragma omp for schedule(static)
    for (int64_t yy = 0; yy < (int64_t)h; yy++) {
      if (fast) nv12_row_fast_avx2(&c, prow(&s[0], (uint32_t)yy), prow(&s[1], (uint32_t)yy >> 1), w, r8, g8, b8);
      else nv12_row_fast_base(&c, prow(&s[0], (uint32_t)yy), prow(&s[1], (uint32_t)yy >> 1), w, r8, g8, b8);
      uint8_t* o = prow(&d[0], (uint32_t)yy);
      const uint8_t *a0 = bgr ? b8 : r8, *a2 = bgr ? r8 : b8;
      for (uint32_t x = 0; x < w; x++) { o[3 * x] = a0[x]; o[3 * x + 1] = g8[x]; o[3 * x + 2] = a2[x]; }
    }
    free(tmp);
  }
}

/* ------------------------------------------------------------------------------------------
 * RGB -> YUV family: C11 bgr_yuv444 (:626-672), C12 bgr_ycbcr (:686-717), C13 rgb_yuv444 (:731-772;
 * the reference's MPEG branch writes packed YCbCr into plane 0 of a planar surface — a bug we do
 * not replicate: planar YCbCr is produced), C14 rgb_planar_yuv444 (:786-830), C16 rgb_yuv420
 * (:887-931).  4:2:0 chroma = mean of the (up to) 2x2 quad's unrounded chroma [A6].
 * ------------------------------------------------------------------------------------------ */
static void rgb_to_yuv(int mode, int sf, int df, const rgb2yuv_dec* m, uint32_t w, uint32_t h,
                       const vpfo_plane* s, const vpfo_plane* d) {
  rgb2yuv_f32 c = make_rgb2yuv_f32(m);
  const int sub = (df == F_YUV420 || df == F_YCBCR);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t yy = 0; yy < (int64_t)h; yy++) {
    uint32_t y = (uint32_t)yy;
    for (uint32_t x = 0; x < w; x++) {
      int r, g, b;
      fetch_rgb(sf, s, x, y, &r, &g, &b);
      prow(&d[0], y)[x] = (mode == VPFO_EXACT) ? round6(rgb2yuv_num(m, 0, r, g, b))
                                              : sat_trunc(rgb2yuv_f(&c, 0, (float)r, (float)g, (float)b));
      if (!sub) {
        for (int k = 1; k < 3; k++)
          prow(&d[k], y)[x] = (mode == VPFO_EXACT) ? round6(rgb2yuv_num(m, k, r, g, b))
                                                  : sat_trunc(rgb2yuv_f(&c, k, (float)r, (float)g, (float)b));
      }
    }
  }
  if (sub) {
    uint32_t cw = cdiv2(w), ch = cdiv2(h);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t yy = 0; yy < (int64_t)ch; yy++) {
      uint32_t cy = (uint32_t)yy;
      for (uint32_t cx = 0; cx < cw; cx++) {
        /* edge quads replicate the last column/row so that every quad has four taps */
        uint32_t x0 = 2 * cx, y0 = 2 * cy, x1 = (x0 + 1 < w) ? x0 + 1 : x0, y1 = (y0 + 1 < h) ? y0 + 1 : y0;
        uint32_t xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
        for (int k = 1; k < 3; k++) {
          if (mode == VPFO_EXACT && g_a6 == 1) { /* A6 alternative: the quad's top-left pixel alone */
            int r, g, b;
            fetch_rgb(sf, s, x0, y0, &r, &g, &b);
            prow(&d[k], cy)[cx] = round6(rgb2yuv_num(m, k, r, g, b));
          } else if (mode == VPFO_EXACT) {
            int64_t acc = 0;
            for (int t = 0; t < 4; t++) {
              int r, g, b;
              fetch_rgb(sf, s, xs[t], ys[t], &r, &g, &b);
              acc += rgb2yuv_num(m, k, r, g, b);
            }
            /* mean of four: (acc/4)/1e6 rounded half up == floor((acc + 2e6) / 4e6) */
            prow(&d[k], cy)[cx] = clamp_u8(floordiv(acc + 2000000, 4000000));
          } else {
            /* kernel op order: sum the four taps' R,G,B as exact small integers, then one
             * matrix row on the sums scaled by 0.25 (exact in fp32) */
            int rs = 0, gs = 0, bs = 0;
            for (int t = 0; t < 4; t++) {
              int r, g, b;
              fetch_rgb(sf, s, xs[t], ys[t], &r, &g, &b);
              rs += r; gs += g; bs += b;
            }
            prow(&d[k], cy)[cx] = sat_trunc(rgb2yuv_f(&c, k, 0.25f * (float)rs, 0.25f * (float)gs, 0.25f * (float)bs));
          }
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * pure re-layout converters
 * ------------------------------------------------------------------------------------------ */
static void copy_plane(const vpfo_plane* s, const vpfo_plane* d, uint32_t bytes, uint32_t rows) {
  for (uint32_t y = 0; y < rows; y++) memcpy(prow(d, y), prow(s, y), bytes);
}
/* C3 nv12_yuv420 (TasksColorCvt.cpp:196-240): both NPP branches are a pure re-layout [A3] */
static void nv12_to_yuv420(uint32_t w, uint32_t h, const vpfo_plane* s, const vpfo_plane* d) {
  copy_plane(&s[0], &d[0], w, h);
  uint32_t cw = cdiv2(w), ch = cdiv2(h);
  for (uint32_t y = 0; y < ch; y++) {
    const uint8_t* uv = prow(&s[1], y);
    uint8_t *u = prow(&d[1], y), *v = prow(&d[2], y);
    for (uint32_t x = 0; x < cw; x++) { u[x] = uv[2 * x]; v[x] = uv[2 * x + 1]; }
  }
}
/* C4 yuv420_nv12 (TasksColorCvt.cpp:945-975) */
static void yuv420_to_nv12(uint32_t w, uint32_t h, const vpfo_plane* s, const vpfo_plane* d) {
  copy_plane(&s[0], &d[0], w, h);
  uint32_t cw = cdiv2(w), ch = cdiv2(h);
  for (uint32_t y = 0; y < ch; y++) {
    const uint8_t *u = prow(&s[1], y), *v = prow(&s[2], y);
    uint8_t* uv = prow(&d[1], y);
    for (uint32_t x = 0; x < cw; x++) { uv[2 * x] = u[x]; uv[2 * x + 1] = v[x]; }
  }
}
/* C18 rgb8_deinterleave (:1059-1088), C19 rgb8_interleave (:1102-1131), C20 rgb_bgr/bgr_rgb
 * (:1145-1170,1184-1209: nppiSwapChannels order {2,1,0}) — all via fetch/store */
static void rgb_relayout(int sf, int df, uint32_t w, uint32_t h, const vpfo_plane* s, const vpfo_plane* d) {
  for (uint32_t y = 0; y < h; y++)
    for (uint32_t x = 0; x < w; x++) {
      int r, g, b;
      fetch_rgb(sf, s, x, y, &r, &g, &b);
      store_rgb(df, d, x, y, (uint8_t)r, (uint8_t)g, (uint8_t)b);
    }
}
/* C17 p16_nv12 (:990-1045): per 16-bit sample nppiDivC_16u_C1RSfs(256, sf=0) then Convert_16u8u.
 * [A7] NPP's integer DivC rounds to nearest (half away from zero for unsigned = half up), result
 * saturated to 8 bits: (v + 128) >> 8, min 255. */
static void p16_to_nv12(uint32_t w, uint32_t h, const vpfo_plane* s, const vpfo_plane* d) {
  uint32_t cw2 = 2 * cdiv2(w), ch = cdiv2(h);
  for (uint32_t y = 0; y < h; y++) {
    const uint16_t* i = (const uint16_t*)prow(&s[0], y);
    uint8_t* o = prow(&d[0], y);
    for (uint32_t x = 0; x < w; x++) { uint32_t v = ((uint32_t)i[x] + 128) >> 8; o[x] = (uint8_t)(v > 255 ? 255 : v); }
  }
  for (uint32_t y = 0; y < ch; y++) {
    const uint16_t* i = (const uint16_t*)prow(&s[1], y);
    uint8_t* o = prow(&d[1], y);
    for (uint32_t x = 0; x < cw2; x++) { uint32_t v = ((uint32_t)i[x] + 128) >> 8; o[x] = (uint8_t)(v > 255 ? 255 : v); }
  }
}

/* supported-combination table of the kernel library (a superset of what each reference *_Impl
 * accepts; the per-impl rejections of TasksColorCvt.cpp live in the Task layer, not here) */
int vpfo_convert_supported(int sf, int df, int cs, int cr) {
  if (is_yuv_src(sf) && sf != F_YCBCR && is_rgb3(df)) return valid_cscr(cs, cr);
  if (is_rgb3(sf) && (df == F_YUV444 || df == F_YUV420 || df == F_YCBCR)) return cs == CS_601 && (cr == CR_MPEG || cr == CR_JPEG);
  if (sf == F_NV12 && (df == F_YUV420 || df == F_Y)) return 1;
  if (sf == F_YUV420 && df == F_NV12) return 1;
  if (is_rgb3(sf) && is_rgb3(df) && sf != df) return 1;
  if (sf == F_Y && df == F_YUV444) return 1;
  if (is_rgb3(sf) && df == F_Y) return 1;
  if (sf == F_RGB && df == F_RGB_32F) return 1;
  if (sf == F_RGB_32F && df == F_RGB_32F_PLANAR) return 1;
  if ((sf == F_P10 || sf == F_P12) && df == F_NV12) return 1;
  return 0;
}

int vpfo_convert(int mode, int sf, int df, int cs, int cr, uint32_t w, uint32_t h, const vpfo_plane s[3],
                 const vpfo_plane d[3]) {
  if (!vpfo_convert_supported(sf, df, cs, cr)) return VPFO_UNSUPPORTED;
  if (mode != VPFO_EXACT && (g_a2 || g_a6)) return VPFO_UNSUPPORTED; /* FP32 restates the kernels: default conventions only */
  if (!w || !h || !check_planes(sf, w, s) || !check_planes(df, w, d)) return VPFO_BAD_ARG;
  if (is_yuv_src(sf) && is_rgb3(df)) {
    if (mode == VPFO_FP32 && sf == F_NV12 && df != F_RGB_PLANAR) nv12_to_rgb_fast(df == F_BGR, &k_yuv2rgb[cs][cr], w, h, s, d);
    else yuv_to_rgb(mode, sf, df, &k_yuv2rgb[cs][cr], w, h, s, d);
  } else if (is_rgb3(sf) && (df == F_YUV444 || df == F_YUV420 || df == F_YCBCR)) {
    rgb_to_yuv(mode, sf, df, &k_rgb2yuv[cr], w, h, s, d);
  } else if (sf == F_NV12 && df == F_YUV420) {
    nv12_to_yuv420(w, h, s, d);
  } else if (sf == F_YUV420 && df == F_NV12) {
    yuv420_to_nv12(w, h, s, d);
  } else if (sf == F_NV12 && df == F_Y) { /* C5 nv12_y (:254-279): luma copy */
    copy_plane(&s[0], &d[0], w, h);
  } else if (is_rgb3(sf) && is_rgb3(df)) {
    rgb_relayout(sf, df, w, h, s, d);
  } else if (sf == F_Y && df == F_YUV444) { /* C15 y_yuv444 (:844-873): chroma = 128 */
    copy_plane(&s[0], &d[0], w, h);
    for (uint32_t y = 0; y < h; y++) { memset(prow(&d[1], y), 128, w); memset(prow(&d[2], y), 128, w); }
  } else if (df == F_Y) { /* C23 rbg8_y (:293-308) nppiRGBToGray: .299R+.587G+.114B [A5] */
    for (uint32_t y = 0; y < h; y++)
      for (uint32_t x = 0; x < w; x++) {
        int r, g, b;
        fetch_rgb(sf, s, x, y, &r, &g, &b);
        if (mode == VPFO_EXACT) prow(&d[0], y)[x] = clamp_u8(floordiv((int64_t)299 * r + 587 * g + 114 * b + 500, 1000));
        else prow(&d[0], y)[x] = sat_trunc(__builtin_fmaf((float)r, 0.299f, __builtin_fmaf((float)g, 0.587f, __builtin_fmaf((float)b, 0.114f, 0.5f))));
      }
  } else if (sf == F_RGB && df == F_RGB_32F) { /* C21 rbg8_rgb32f (:1222-1254) nppiScale_8u32f(0,1): v/255 */
    for (uint32_t y = 0; y < h; y++) {
      const uint8_t* i = prow(&s[0], y);
      float* o = (float*)prow(&d[0], y);
      for (uint32_t x = 0; x < 3 * w; x++) o[x] = (float)i[x] / 255.0f;
    }
  } else if (sf == F_RGB_32F && df == F_RGB_32F_PLANAR) { /* C22 rgb32f_deinterleave (:1268-1297) */
    for (uint32_t y = 0; y < h; y++) {
      const float* i = (const float*)prow(&s[0], y);
      for (uint32_t x = 0; x < w; x++)
        for (int k = 0; k < 3; k++) ((float*)prow(&d[k], y))[x] = i[3 * x + k];
    }
  } else if (sf == F_P10 || sf == F_P12) {
    p16_to_nv12(w, h, s, d);
  } else {
    return VPFO_UNSUPPORTED;
  }
  return VPFO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Resize.  Reference: NppResizeSurfacePacked3C_Impl::Run Tasks.cpp:1162-1203 (nppiResize_8u_C3R),
 * NppResizeSurfacePlanar_Impl::Run :1217-1261 (nppiResize_8u_C1R per plane),
 * ResizeSurfaceSemiPlanar_Impl :1285-1324 (NV12 = C3 -> R2 -> C4).
 * The reference passes NPPI_INTER_LANCZOS (:1190); north_star asks for bilinear, which is what is
 * restated here (interp = 1).  Pixel-centre mapping with edge clamp [A8]:
 *   s = (d + 0.5) * (S/D) - 0.5, clamped to [0, S-1]
 * ------------------------------------------------------------------------------------------ */
static inline uint8_t bilerp_u8(int mode, int p00, int p01, int p10, int p11, double fx, double fy, float fxf, float fyf) {
  if (mode == VPFO_EXACT) {
    double top = p00 + fx * (p01 - p00), bot = p10 + fx * (p11 - p10);
    double v = top + fy * (bot - top);
    double r = floor(v + 0.5);
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
  }
  float top = __builtin_fmaf(fxf, (float)(p01 - p00), (float)p00);
  float bot = __builtin_fmaf(fxf, (float)(p11 - p10), (float)p10);
  float v = __builtin_fmaf(fyf, bot - top, top);
  return sat_trunc(v + 0.5f);
}

typedef struct { uint32_t i0, i1; double f; float ff; } tap;
/* A8: source coordinate of destination index d (EXACT mode).  0 (default): pixel centres, s = (d + 0.5) S/D - 0.5;
 * 1: top-left origin, s = d S/D;  2: corners aligned, s = d (S-1)/(D-1) */
static inline double src_coord(uint32_t d, uint32_t S, uint32_t D) {
  if (g_a8 == 1) return d * ((double)S / (double)D);
  if (g_a8 == 2) return D > 1 ? d * ((double)(S - 1) / (double)(D - 1)) : 0.0;
  return (d + 0.5) * ((double)S / (double)D) - 0.5;
}
static void make_taps(int mode, int interp, uint32_t S, uint32_t D, tap* t) {
  double sc = (double)S / (double)D;
  float scf = (float)S / (float)D;
  for (uint32_t d = 0; d < D; d++) {
    if (interp == 0) { /* nearest: floor((d+0.5)*S/D) */
      uint32_t i = (mode == VPFO_EXACT) ? (uint32_t)floor(g_a8 ? src_coord(d, S, D) + 0.5 : (d + 0.5) * sc) : (uint32_t)(((float)d + 0.5f) * scf);
      if (i > S - 1) i = S - 1;
      t[d].i0 = t[d].i1 = i; t[d].f = 0; t[d].ff = 0;
      continue;
    }
    if (mode == VPFO_EXACT) {
      double s = src_coord(d, S, D);
      if (s < 0) s = 0;
      if (s > (double)(S - 1)) s = (double)(S - 1);
      uint32_t i0 = (uint32_t)floor(s);
      t[d].i0 = i0; t[d].i1 = (i0 + 1 < S) ? i0 + 1 : S - 1; t[d].f = s - i0; t[d].ff = (float)t[d].f;
    } else {
      float s = __builtin_fmaf((float)d + 0.5f, scf, -0.5f);
      s = s < 0.f ? 0.f : s;
      s = s > (float)(S - 1) ? (float)(S - 1) : s;
      uint32_t i0 = (uint32_t)(int)s;
      t[d].i0 = i0; t[d].i1 = (i0 + 1 < S) ? i0 + 1 : S - 1; t[d].ff = s - (float)i0; t[d].f = t[d].ff;
    }
  }
}
static int resize_plane_lanczos(int mode, int ch, uint32_t sw, uint32_t sh, const vpfo_plane* s, uint32_t dw, uint32_t dh,
                                const vpfo_plane* d);
static int resize_plane(int mode, int interp, int ch, uint32_t sw, uint32_t sh, const vpfo_plane* s, uint32_t dw,
                        uint32_t dh, const vpfo_plane* d) {
  if (interp == 2) return resize_plane_lanczos(mode, ch, sw, sh, s, dw, dh, d);
  tap* tx = (tap*)malloc(sizeof(tap) * dw);
  tap* ty = (tap*)malloc(sizeof(tap) * dh);
  if (!tx || !ty) { free(tx); free(ty); return VPFO_BAD_ARG; }
  make_taps(mode, interp, sw, dw, tx);
  make_taps(mode, interp, sh, dh, ty);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t yy = 0; yy < (int64_t)dh; yy++) {
    const uint8_t *r0 = prow(s, ty[yy].i0), *r1 = prow(s, ty[yy].i1);
    uint8_t* o = prow(d, (uint32_t)yy);
    for (uint32_t x = 0; x < dw; x++)
      for (int k = 0; k < ch; k++)
        o[ch * x + k] = bilerp_u8(mode, r0[ch * tx[x].i0 + k], r0[ch * tx[x].i1 + k], r1[ch * tx[x].i0 + k],
                                  r1[ch * tx[x].i1 + k], tx[x].f, ty[yy].f, tx[x].ff, ty[yy].ff);
  }
  free(tx); free(ty);
  return VPFO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Lanczos-3 (interp = 2).  The reference's resizer asks NPP for NPPI_INTER_LANCZOS (Tasks.cpp:1190,1248); NPP's
 * exact kernel support / normalisation is unpublished, so this is the textbook separable Lanczos-3:
 *   s = (d + 0.5) * S/D - 0.5;  i0 = floor(s);  f = s - i0;  taps i0-2 .. i0+3 (indices clamped to the image),
 *   w_k = L(f - (k - 2)),  L(t) = sinc(t) sinc(t/3),  weights normalised to sum 1, no widening when minifying [A10 = 0].
 * A10 = 1 (EXACT mode only; round 5): the OTHER plausible reading of "NPPI_INTER_LANCZOS" — the anti-aliasing form PIL and
 * swscale implement: when minifying, the kernel is stretched by fs = max(1, S/D) per axis, taps = every source sample i with
 * |i - s| < 3 fs, w_i = L((i - s) / fs), normalised (resize_plane_lanczos_wide).  On up-scales (fs = 1) the two coincide tap for tap;
 * on a 1.5 x down-scale they differ by a mean of ~8 LSB (max 30+) on noise, so which one NPP follows decides whether the kernels'
 * six-tap windows are the right filter at all (the pin kit classifies it from NPP's impulse response).
 * EXACT: double + libm.  FP32: the kernels' arithmetic — sin(pi f), sin(pi f/3), cos(pi f/3) from fixed fma polynomials (so host
 * and device agree bit for bit), the six taps from angle-addition identities; on 8-bit surfaces BOTH passes then run in integers on
 * Q14 weights (lanczos_weights_q14) with the row sums rounded to Q6 in between and the vertical products formed from byte-wide partial
 * products (see resize_plane_lanczos; round 3 — the vertical pass was an fp32 fma chain before, which an integer matrix unit cannot
 * reproduce); float surfaces stay fp32 throughout.
 * ------------------------------------------------------------------------------------------ */
static inline float lz_sinpi_poly(float g) { /* sin(pi g), g in [0, 0.5]; odd Taylor polynomial in x = pi g, degree 11 */
  const float x = 3.14159274f * g, x2 = x * x;
  float p = __builtin_fmaf(x2, -2.50521084e-8f, 2.75573192e-6f);
  p = __builtin_fmaf(x2, p, -1.98412698e-4f);
  p = __builtin_fmaf(x2, p, 8.33333333e-3f);
  p = __builtin_fmaf(x2, p, -1.66666667e-1f);
  p = __builtin_fmaf(x2, p, 1.0f);
  return x * p;
}
static inline float lz_cos_poly(float x) { /* cos(x), x in [0, pi/3]; even Taylor polynomial, degree 10 */
  const float x2 = x * x;
  float p = __builtin_fmaf(x2, -2.75573192e-7f, 2.48015873e-5f);
  p = __builtin_fmaf(x2, p, -1.38888889e-3f);
  p = __builtin_fmaf(x2, p, 4.16666667e-2f);
  p = __builtin_fmaf(x2, p, -0.5f);
  return __builtin_fmaf(x2, p, 1.0f);
}
static void lanczos_weights_fp32(float f, float w[6]) {
  if (f == 0.f) { w[0] = w[1] = w[3] = w[4] = w[5] = 0.f; w[2] = 1.f; return; }
  const float s1 = lz_sinpi_poly(f <= 0.5f ? f : 1.0f - f); /* sin(pi f) */
  const float s3 = lz_sinpi_poly(f * 0.333333343f);           /* sin(pi f / 3), argument in [0, 1/3) */
  const float c3 = lz_cos_poly(1.04719758f * f);              /* cos(pi f / 3) */
  /* tap k: t = f - m, m = k - 2.  sin(pi t) = (-1)^m sin(pi f);  sin(pi t/3) = s3 cos(m pi/3) - c3 sin(m pi/3) */
  static const float cm[6] = {-0.5f, 0.5f, 1.0f, 0.5f, -0.5f, -1.0f};                        /* cos(m pi/3), m=-2..3 */
  static const float sm[6] = {-0.866025388f, -0.866025388f, 0.0f, 0.866025388f, 0.866025388f, 0.0f}; /* sin(m pi/3) */
  static const float sg[6] = {1.0f, -1.0f, 1.0f, -1.0f, 1.0f, -1.0f};                        /* (-1)^m */
  /* w_k = n_k D_k / sum_j n_j D_j, n_k = sin(pi t_k) sin(pi t_k / 3), D_k = prod_{j != k} t_j^2: the kernels' form, one division per
   * weight set (numerator and denominator of L(t_k) / sum L(t_j) multiplied by prod t_j^2; the 3 / pi^2 cancels) */
  float n[6], u[6], pre[6], suf[6];
  for (int k = 0; k < 6; k++) {
    const float t = f - (float)(k - 2);
    u[k] = t * t;
    n[k] = (sg[k] * s1) * __builtin_fmaf(s3, cm[k], -(c3 * sm[k]));
  }
  pre[0] = 1.0f; suf[5] = 1.0f;
  for (int k = 1; k < 6; k++) pre[k] = pre[k - 1] * u[k - 1];
  for (int k = 4; k >= 0; k--) suf[k] = suf[k + 1] * u[k + 1];
  float sum = 0.f;
  for (int k = 0; k < 6; k++) { w[k] = n[k] * (pre[k] * suf[k]); sum += w[k]; }
  const float inv = 1.0f / sum;
  for (int k = 0; k < 6; k++) w[k] *= inv;
}
/* The kernels run Lanczos on 8-bit surfaces in integers (v_mfma_i32_16x16x64_i8 / v_dot2_i32_i16): the six normalised fp32 weights
 * become Q14 fixed point, q_k = rint(w_k * 16384) (ties to even), and tap 2 absorbs the rounding residue so that the six sum to
 * exactly 16384 (a flat picture stays flat).  H = sum q_k * p_k is then exact in 32 bits, whatever the order. */
static void lanczos_weights_q14(const float w[6], int32_t q[6]) {
  int32_t sum = 0;
  for (int k = 0; k < 6; k++) { q[k] = (int32_t)nearbyintf(w[k] * 16384.0f); sum += q[k]; }
  q[2] += 16384 - sum;
}
/* arithmetic shift right (floor division by 2^n) spelled so that it does not depend on the compiler's choice for negative operands */
static inline int32_t asr32(int32_t v, int n) { return v >= 0 ? v >> n : -(int32_t)(((uint32_t)(-(v + 1)) >> n) + 1u); }
static void lanczos_weights_exact(double f, double w[6]) {
  double sum = 0;
  for (int k = 0; k < 6; k++) {
    const double t = f - (k - 2);
    w[k] = (t == 0.0) ? 1.0 : 3.0 * sin(M_PI * t) * sin(M_PI * t / 3.0) / (M_PI * M_PI * t * t);
    sum += w[k];
  }
  for (int k = 0; k < 6; k++) w[k] /= sum;
}
typedef struct { int32_t idx[6]; double w[6]; float wf[6]; int32_t q[6]; } ltap;
static void make_ltaps(int mode, uint32_t S, uint32_t D, ltap* t) {
  const float scf = (float)S / (float)D;
  for (uint32_t d = 0; d < D; d++) {
    int32_t i0;
    if (mode == VPFO_EXACT) {
      const double s = src_coord(d, S, D);
      i0 = (int32_t)floor(s);
      lanczos_weights_exact(s - i0, t[d].w);
      for (int k = 0; k < 6; k++) t[d].wf[k] = (float)t[d].w[k];
    } else {
      const float s = __builtin_fmaf((float)d + 0.5f, scf, -0.5f);
      const float fl = floorf(s);
      i0 = (int32_t)fl;
      lanczos_weights_fp32(s - fl, t[d].wf);
      lanczos_weights_q14(t[d].wf, t[d].q);
      for (int k = 0; k < 6; k++) t[d].w[k] = t[d].wf[k];
    }
    for (int k = 0; k < 6; k++) {
      int32_t i = i0 + k - 2;
      t[d].idx[k] = i < 0 ? 0 : (i > (int32_t)S - 1 ? (int32_t)S - 1 : i);
    }
  }
}
/* the FP32-mode taps of one axis, for tests that model a kernel's data flow on the CPU: i0[d] = floor of the source coordinate (taps
 * i0 - 2 .. i0 + 3, to be clamped by the caller), q[6 d + k] = Q14 weights */
int vpfo_lanczos_taps_q14(uint32_t S, uint32_t D, int32_t* i0, int32_t* q) {
  if (!S || !D || !i0 || !q) return VPFO_BAD_ARG;
  const float scf = (float)S / (float)D;
  for (uint32_t d = 0; d < D; d++) {
    const float s = __builtin_fmaf((float)d + 0.5f, scf, -0.5f);
    const float fl = floorf(s);
    float wf[6];
    i0[d] = (int32_t)fl;
    lanczos_weights_fp32(s - fl, wf);
    lanczos_weights_q14(wf, q + 6 * d);
  }
  return VPFO_OK;
}
/* A10 = 1: Lanczos-3 with the support scaled by the minification factor (EXACT arithmetic: double + libm; 8-bit: round half up, clamp;
 * float: no rounding).  Separable, horizontal pass first into a double image, indices beyond the picture clamp onto the edge sample
 * (the same border rule as the six-tap form, so that A10 is the ONLY thing the switch changes: at fs = 1 the taps are i0-2 .. i0+3 with
 * the same weights).  Source coordinate by A8. */
typedef struct { int32_t first, n; double* w; } wtap;
static int make_wtaps(uint32_t S, uint32_t D, wtap** out) {
  const double fs = (double)S / (double)D > 1.0 ? (double)S / (double)D : 1.0, sup = 3.0 * fs;
  wtap* t = (wtap*)calloc(D, sizeof(wtap));
  if (!t) return 0;
  for (uint32_t d = 0; d < D; d++) {
    const double s = src_coord(d, S, D);
    const int32_t lo = (int32_t)ceil(s - sup), hi = (int32_t)floor(s + sup);
    t[d].first = lo; t[d].n = hi - lo + 1;
    t[d].w = (double*)malloc(sizeof(double) * (size_t)t[d].n);
    if (!t[d].w) { for (uint32_t k = 0; k < d; k++) free(t[k].w); free(t); return 0; }
    double sum = 0;
    for (int32_t k = 0; k < t[d].n; k++) {
      const double x = ((double)(lo + k) - s) / fs;
      const double w = x == 0.0 ? 1.0 : fabs(x) >= 3.0 ? 0.0 : 3.0 * sin(M_PI * x) * sin(M_PI * x / 3.0) / (M_PI * M_PI * x * x);
      t[d].w[k] = w; sum += w;
    }
    for (int32_t k = 0; k < t[d].n; k++) t[d].w[k] /= sum;
  }
  *out = t;
  return 1;
}
static void free_wtaps(wtap* t, uint32_t D) { if (t) { for (uint32_t d = 0; d < D; d++) free(t[d].w); free(t); } }
static int resize_plane_lanczos_wide(int is_f32, int ch, uint32_t sw, uint32_t sh, const vpfo_plane* s, uint32_t dw, uint32_t dh, const vpfo_plane* d) {
  wtap *tx = 0, *ty = 0;
  const size_t rowv = (size_t)dw * (size_t)ch;
  double* H = (double*)malloc(sizeof(double) * rowv * sh);
  if (!H || !make_wtaps(sw, dw, &tx) || !make_wtaps(sh, dh, &ty)) { free(H); free_wtaps(tx, dw); free_wtaps(ty, dh); return VPFO_BAD_ARG; }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t y = 0; y < (int64_t)sh; y++) {
    const uint8_t* r8 = prow(s, (uint32_t)y);
    const float* rf = (const float*)r8;
    for (uint32_t x = 0; x < dw; x++)
      for (int c = 0; c < ch; c++) {
        double a = 0;
        for (int32_t k = 0; k < tx[x].n; k++) {
          int32_t i = tx[x].first + k;
          i = i < 0 ? 0 : (i > (int32_t)sw - 1 ? (int32_t)sw - 1 : i);
          a += tx[x].w[k] * (is_f32 ? (double)rf[ch * i + c] : (double)r8[ch * i + c]);
        }
        H[(size_t)y * rowv + (size_t)ch * x + c] = a;
      }
  }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t y = 0; y < (int64_t)dh; y++) {
    uint8_t* o8 = prow(d, (uint32_t)y);
    float* of = (float*)o8;
    for (size_t v = 0; v < rowv; v++) {
      double a = 0;
      for (int32_t k = 0; k < ty[y].n; k++) {
        int32_t i = ty[y].first + k;
        i = i < 0 ? 0 : (i > (int32_t)sh - 1 ? (int32_t)sh - 1 : i);
        a += ty[y].w[k] * H[(size_t)i * rowv + v];
      }
      if (is_f32) of[v] = (float)a;
      else { const double r = floor(a + 0.5); o8[v] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); }
    }
  }
  free(H); free_wtaps(tx, dw); free_wtaps(ty, dh);
  return VPFO_OK;
}
static int resize_plane_lanczos(int mode, int ch, uint32_t sw, uint32_t sh, const vpfo_plane* s, uint32_t dw, uint32_t dh,
                                const vpfo_plane* d) {
  if (mode == VPFO_EXACT && g_a10 == 1) return resize_plane_lanczos_wide(0, ch, sw, sh, s, dw, dh, d);
  ltap* tx = (ltap*)malloc(sizeof(ltap) * dw);
  ltap* ty = (ltap*)malloc(sizeof(ltap) * dh);
  if (!tx || !ty) { free(tx); free(ty); return VPFO_BAD_ARG; }
  make_ltaps(mode, sw, dw, tx);
  make_ltaps(mode, sh, dh, ty);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t yy = 0; yy < (int64_t)dh; yy++) {
    uint8_t* o = prow(d, (uint32_t)yy);
    for (uint32_t x = 0; x < dw; x++)
      for (int c = 0; c < ch; c++) {
        if (mode == VPFO_EXACT) {
          double acc = 0;
          for (int ky = 0; ky < 6; ky++) {
            const uint8_t* r = prow(s, (uint32_t)ty[yy].idx[ky]);
            double ra = 0;
            for (int kx = 0; kx < 6; kx++) ra += tx[x].w[kx] * r[ch * tx[x].idx[kx] + c];
            acc += ty[yy].w[ky] * ra;
          }
          const double v = floor(acc + 0.5);
          o[ch * x + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        } else { /* kernel arithmetic, all integers and therefore order-free (the MFMA kernel sums 64 or 128 terms per instruction, the gather
                    kernel six).  H = sum q_x p exact (Q14); Hr = (H + 128) >> 8 = H rounded half up to Q6, which fits 16 bits (Lanczos
                    overshoot included: -4.5k .. 20.8k).  The vertical pass multiplies 16-bit Hr by 15-bit q_y the way a byte-wide matrix unit
                    does: both factors as two signed bytes about a centre, z = Hr - 8192 = 256 zh + zl and q_y = 256 qh + ql with zl, ql in
                    [-128, 127], and the product as its three upper partial products — qh zh 2^16 + (qh zl + ql zh) 2^8; the lowest one, ql zl
                    (|.| <= 2^14 per tap, < 0.1 LSB of the result over six taps, 0.013 LSB rms), is NOT formed (round 3: it costs a fourth
                    matrix instruction and a shift-add per output byte in a kernel that is issue-bound).  V = sum (q_y z - ql zl) + 8192 * 16384
                    is a multiple of 256; out = clamp((V / 256 + 2^11) >> 12).  Identity (q = 0 0 16384 0 0 0: ql = 0) stays exact, a flat
                    picture stays flat, and the result is within 1 LSB of EXACT (test_oracle_kat.py). */
          int32_t v = 1 << 19; /* 8192 * 16384 / 256 */
          for (int ky = 0; ky < 6; ky++) {
            /* taps that the edge clamp puts on the same source row act as ONE tap with the summed weight (it is the weight of the row
               that is split into bytes, whoever contributes to it): the run's last tap carries the sum */
            int32_t q = ty[yy].q[ky];
            while (ky < 5 && ty[yy].idx[ky + 1] == ty[yy].idx[ky]) q += ty[yy].q[++ky];
            const uint8_t* r = prow(s, (uint32_t)ty[yy].idx[ky]);
            int32_t h = 0;
            for (int kx = 0; kx < 6; kx++) h += tx[x].q[kx] * (int32_t)r[ch * tx[x].idx[kx] + c];
            const int32_t z = asr32(h + 128, 8) - 8192;
            const int32_t zl = ((z + 128) & 0xff) - 128, ql = ((q + 128) & 0xff) - 128;
            v += asr32(q * z - ql * zl, 8); /* exact: the difference is a multiple of 256 */
          }
          v = asr32(v + (1 << 11), 12);
          o[ch * x + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      }
  }
  free(tx); free(ty);
  return VPFO_OK;
}

/* 32-bit float surfaces (reference: NppResizeSurfacePacked32F3C_Impl Tasks.cpp:1334-1387, nppiResize_32f_C3R;
 * NppResizeSurface32FPlanar_Impl :1390-1445, nppiResize_32f_C1R per plane): the same taps and operation order as the 8-bit
 * paths, on float samples, no rounding and no clamping of the result. */
static int resize_plane_f32(int mode, int interp, int ch, uint32_t sw, uint32_t sh, const vpfo_plane* s, uint32_t dw,
                            uint32_t dh, const vpfo_plane* d) {
  if (interp == 2) {
    if (mode == VPFO_EXACT && g_a10 == 1) return resize_plane_lanczos_wide(1, ch, sw, sh, s, dw, dh, d);
    ltap* tx = (ltap*)malloc(sizeof(ltap) * dw);
    ltap* ty = (ltap*)malloc(sizeof(ltap) * dh);
    if (!tx || !ty) { free(tx); free(ty); return VPFO_BAD_ARG; }
    make_ltaps(mode, sw, dw, tx);
    make_ltaps(mode, sh, dh, ty);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t yy = 0; yy < (int64_t)dh; yy++) {
      float* o = (float*)prow(d, (uint32_t)yy);
      for (uint32_t x = 0; x < dw; x++)
        for (int c = 0; c < ch; c++) {
          if (mode == VPFO_EXACT) {
            double acc = 0;
            for (int ky = 0; ky < 6; ky++) {
              const float* r = (const float*)prow(s, (uint32_t)ty[yy].idx[ky]);
              double ra = 0;
              for (int kx = 0; kx < 6; kx++) ra += tx[x].w[kx] * (double)r[ch * tx[x].idx[kx] + c];
              acc += ty[yy].w[ky] * ra;
            }
            o[ch * x + c] = (float)acc;
          } else {
            float acc = 0.f;
            for (int ky = 0; ky < 6; ky++) {
              const float* r = (const float*)prow(s, (uint32_t)ty[yy].idx[ky]);
              float ra = 0.f;
              for (int kx = 0; kx < 6; kx++) ra = __builtin_fmaf(tx[x].wf[kx], r[ch * tx[x].idx[kx] + c], ra);
              acc = __builtin_fmaf(ty[yy].wf[ky], ra, acc);
            }
            o[ch * x + c] = acc;
          }
        }
    }
    free(tx); free(ty);
    return VPFO_OK;
  }
  tap* tx = (tap*)malloc(sizeof(tap) * dw);
  tap* ty = (tap*)malloc(sizeof(tap) * dh);
  if (!tx || !ty) { free(tx); free(ty); return VPFO_BAD_ARG; }
  make_taps(mode, interp, sw, dw, tx);
  make_taps(mode, interp, sh, dh, ty);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t yy = 0; yy < (int64_t)dh; yy++) {
    const float *r0 = (const float*)prow(s, ty[yy].i0), *r1 = (const float*)prow(s, ty[yy].i1);
    float* o = (float*)prow(d, (uint32_t)yy);
    for (uint32_t x = 0; x < dw; x++)
      for (int k = 0; k < ch; k++) {
        const float p00 = r0[ch * tx[x].i0 + k], p01 = r0[ch * tx[x].i1 + k], p10 = r1[ch * tx[x].i0 + k], p11 = r1[ch * tx[x].i1 + k];
        if (mode == VPFO_EXACT) {
          const double top = p00 + tx[x].f * ((double)p01 - p00), bot = p10 + tx[x].f * ((double)p11 - p10);
          o[ch * x + k] = (float)(top + ty[yy].f * (bot - top));
        } else {
          const float top = __builtin_fmaf(tx[x].ff, p01 - p00, p00), bot = __builtin_fmaf(tx[x].ff, p11 - p10, p10);
          o[ch * x + k] = __builtin_fmaf(ty[yy].ff, bot - top, top);
        }
      }
  }
  free(tx); free(ty);
  return VPFO_OK;
}

int vpfo_resize(int mode, int fmt, int interp, uint32_t sw, uint32_t sh, const vpfo_plane s[3], uint32_t dw,
                uint32_t dh, const vpfo_plane d[3]) {
  if (interp != 0 && interp != 1 && interp != 2) return VPFO_UNSUPPORTED;
  if (mode != VPFO_EXACT && (g_a8 || (g_a10 && interp == 2))) return VPFO_UNSUPPORTED; /* FP32 restates the kernels: default conventions only */
  if (!sw || !sh || !dw || !dh) return VPFO_BAD_ARG;
  if (!check_planes(fmt, sw, s) || !check_planes(fmt, dw, d)) return (nplanes(fmt) ? VPFO_BAD_ARG : VPFO_UNSUPPORTED);
  switch (fmt) {
    case F_RGB: case F_BGR: return resize_plane(mode, interp, 3, sw, sh, &s[0], dw, dh, &d[0]);
    case F_Y: return resize_plane(mode, interp, 1, sw, sh, &s[0], dw, dh, &d[0]);
    case F_YUV444: case F_RGB_PLANAR:
      for (int k = 0; k < 3; k++) { int e = resize_plane(mode, interp, 1, sw, sh, &s[k], dw, dh, &d[k]); if (e) return e; }
      return VPFO_OK;
    case F_YUV420: case F_YCBCR: {
      int e = resize_plane(mode, interp, 1, sw, sh, &s[0], dw, dh, &d[0]);
      for (int k = 1; k < 3 && !e; k++) e = resize_plane(mode, interp, 1, cdiv2(sw), cdiv2(sh), &s[k], cdiv2(dw), cdiv2(dh), &d[k]);
      return e;
    }
    case F_RGB_32F: return resize_plane_f32(mode, interp, 3, sw, sh, &s[0], dw, dh, &d[0]);
    case F_RGB_32F_PLANAR:
      for (int k = 0; k < 3; k++) { int e = resize_plane_f32(mode, interp, 1, sw, sh, &s[k], dw, dh, &d[k]); if (e) return e; }
      return VPFO_OK;
    case F_NV12: { /* = de-interleave, resize U and V planes, re-interleave (Tasks.cpp:1303-1318): ch=2 does exactly that */
      int e = resize_plane(mode, interp, 1, sw, sh, &s[0], dw, dh, &d[0]);
      if (!e) e = resize_plane(mode, interp, 2, cdiv2(sw), cdiv2(sh), &s[1], cdiv2(dw), cdiv2(dh), &d[1]);
      return e;
    }
    default: return VPFO_UNSUPPORTED;
  }
}

/* ------------------------------------------------------------------------------------------
 * Remap.  Reference: NppRemapSurfacePacked3C_Impl::Run Tasks.cpp:1555-1602 — nppiRemap_8u_C3R with
 * NPPI_INTER_LINEAR (:1590), maps are two tight float[h*w] buffers (:1523-1526, step = w*4 :1585),
 * dst size = map size (:1550).  Out-of-range source coordinates leave dst untouched [A9].
 * ------------------------------------------------------------------------------------------ */
int vpfo_remap(int mode, int fmt, uint32_t sw, uint32_t sh, const vpfo_plane* s, const float* xmap, uint32_t xp,
               const float* ymap, uint32_t yp, uint32_t dw, uint32_t dh, const vpfo_plane* d) {
  if (fmt != F_RGB && fmt != F_BGR) return VPFO_UNSUPPORTED;
  if (!sw || !sh || !dw || !dh || !xmap || !ymap || !check_planes(fmt, sw, s) || !check_planes(fmt, dw, d) ||
      xp < 4 * dw || yp < 4 * dw) return VPFO_BAD_ARG;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t yy = 0; yy < (int64_t)dh; yy++) {
    const float* xm = (const float*)((const uint8_t*)xmap + (size_t)yy * xp);
    const float* ym = (const float*)((const uint8_t*)ymap + (size_t)yy * yp);
    uint8_t* o = prow(d, (uint32_t)yy);
    for (uint32_t x = 0; x < dw; x++) {
      float sx = xm[x], sy = ym[x];
      if (!(sx >= 0.f && sx <= (float)(sw - 1) && sy >= 0.f && sy <= (float)(sh - 1))) continue;
      uint32_t x0 = (uint32_t)(int)sx, y0 = (uint32_t)(int)sy;
      uint32_t x1 = (x0 + 1 < sw) ? x0 + 1 : sw - 1, y1 = (y0 + 1 < sh) ? y0 + 1 : sh - 1;
      float fxf = sx - (float)x0, fyf = sy - (float)y0;
      double fx = (double)sx - x0, fy = (double)sy - y0;
      const uint8_t *r0 = prow(s, y0), *r1 = prow(s, y1);
      for (int k = 0; k < 3; k++)
        o[3 * x + k] = bilerp_u8(mode, r0[3 * x0 + k], r0[3 * x1 + k], r1[3 * x0 + k], r1[3 * x1 + k], fx, fy, fxf, fyf);
    }
  }
  return VPFO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Fused NV12/YUV420 -> bilinear -> RGB.  Defined as the two reference steps run back to back
 * (C1 then R1; samples/SampleDecodeMultiThread.py:50-152 builds exactly this chain), so the
 * oracle literally does that through a temporary.
 * ------------------------------------------------------------------------------------------ */
int vpfo_convert_resize(int mode, int sf, int df, int cs, int cr, uint32_t sw, uint32_t sh, const vpfo_plane s[3],
                        uint32_t dw, uint32_t dh, const vpfo_plane d[3]) {
  if (!(sf == F_NV12 || sf == F_YUV420) || !is_rgb3(df)) return VPFO_UNSUPPORTED;
  if (!valid_cscr(cs, cr)) return VPFO_UNSUPPORTED;
  if (!sw || !sh || !dw || !dh) return VPFO_BAD_ARG;
  uint32_t np = (df == F_RGB_PLANAR) ? 3 : 1, rb = (df == F_RGB_PLANAR) ? sw : 3 * sw;
  uint8_t* tmp = (uint8_t*)malloc((size_t)rb * sh * np);
  if (!tmp) return VPFO_BAD_ARG;
  vpfo_plane t[3] = {{tmp, rb, 0}, {tmp + (size_t)rb * sh, rb, 0}, {tmp + 2 * (size_t)rb * sh, rb, 0}};
  int e = vpfo_convert(mode, sf, df, cs, cr, sw, sh, s, t);
  if (!e) e = vpfo_resize(mode, df, 1, sw, sh, t, dw, dh, d);
  free(tmp);
  return e;
}
