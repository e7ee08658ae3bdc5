/* oracle/ref_shim_hip/npp_over_vpf.h — TEST INFRASTRUCTURE: the drop-in boundary, demonstrated.
 *
 * The reference's converter Task layer (src/TC/src/TasksColorCvt.cpp, compiled unmodified from /root/reference by
 * oracle/Makefile `ref_tc_hip`) calls NPP through ~40 `nppi*_Ctx` entry points.  Here each of those names is an inline
 * function with NPP's published argument list whose body forwards to the C ABI of libvpfhip (include/vpf_hip.h): raw device
 * pointers + byte steps + ROI + the stream out of NppStreamContext go straight into vpf_convert().  So the reference's own
 * nv12_rgb::Execute (TasksColorCvt.cpp:122-182) etc. run on an MI355X and produce this repo's pixels — which the -m gpu test
 * tests/test_gpu_reference_caller.py compares with the oracle.  This is the adapter INTEGRATION.md describes, executed.
 *
 * Each function states which (source format, destination format, colour space, colour range) of vpf_convert NPP documents
 * for that name.  Types follow NPP's public headers; nothing here is taken from the reference tree.
 * Not forwarded (return NPP_ERROR, the reference then yields no surface): nppiRGBToYCbCr_8u_C3R — the reference writes a
 * PACKED image into plane 0 of a planar surface with it (TasksColorCvt.cpp:758, a bug this repo does not replicate) — and
 * the two-step 16-bit path nppiDivC_16u_C1RSfs / nppiConvert_16u8u_C1R (one plane at a time through a scratch plane; vpf_convert
 * narrows a whole P10/P12 surface in one pass, and the reference's SurfaceP10 cannot hold 16-bit samples anyway).
 */
#pragma once
#include <stdint.h>

#include "cuda.h"
#include "vpf_hip.h"

typedef unsigned char Npp8u;
typedef unsigned short Npp16u;
typedef float Npp32f;
typedef int Npp32s;
typedef struct { int width, height; } NppiSize;
typedef struct { int x, y, width, height; } NppiRect;
typedef enum { NPP_NO_ERROR = 0, NPP_SUCCESS = 0, NPP_NO_OPERATION_WARNING = 1, NPP_ERROR = -2 } NppStatus;
typedef enum { NPPI_INTER_NN = 1, NPPI_INTER_LINEAR = 2, NPPI_INTER_CUBIC = 4, NPPI_INTER_LANCZOS = 16 } NppiInterpolationMode;
typedef struct {
  CUstream hStream;
  int nCudaDeviceId, nMultiProcessorCount, nMaxThreadsPerMultiProcessor, nMaxThreadsPerBlock;
  size_t nSharedMemPerBlock;
  int nCudaDevAttrComputeCapabilityMajor, nCudaDevAttrComputeCapabilityMinor;
  unsigned int nStreamFlags;
} NppStreamContext;

/* plain 2-D device copy / fill on the context's stream (implemented over the HIP runtime in ref_tc_hip_shim.cpp) */
extern "C" int ref_hip_copy2d(const void* src, int sstep, void* dst, int dstep, int width_bytes, int rows, void* stream);
extern "C" int ref_hip_set2d(void* dst, int dstep, int value, int width_bytes, int rows, void* stream);
extern "C" void ref_hip_note(const char* npp_name, int vpf_status);

namespace npp_over_vpf {
struct P3 { vpf_plane p[3]; };
inline P3 planes(const void* a, int sa, const void* b = nullptr, int sb = 0, const void* c = nullptr, int sc = 0) {
  return P3{{{const_cast<void*>(a), (uint32_t)sa, 0}, {const_cast<void*>(b), (uint32_t)sb, 0}, {const_cast<void*>(c), (uint32_t)sc, 0}}};
}
inline NppStatus convert(const char* name, int sf, int df, int cs, int cr, NppiSize roi, const P3& s, const P3& d, const NppStreamContext& ctx) {
  const vpf_exec ex = {-1, 0, (void*)ctx.hStream};
  const vpf_status st = vpf_convert(&ex, sf, df, cs, cr, vpf_size{(uint32_t)roi.width, (uint32_t)roi.height}, s.p, d.p);
  ref_hip_note(name, (int)st);
  return st == VPF_OK ? NPP_NO_ERROR : NPP_ERROR;
}
}  // namespace npp_over_vpf

#define NOV_FWD(sf, df, cs, cr, S, D) return npp_over_vpf::convert(__func__, sf, df, cs, cr, roi, S, D, ctx)
#define NOV npp_over_vpf::planes

/* ---- NV12 -> packed RGB / BGR: nv12_rgb / nv12_bgr (TasksColorCvt.cpp:53-182).  "709CSC" = BT.709 limited range, "709HDTV" =
 * BT.709 full range, the plain name = NPP's full-range "YUV" model (BT.601 coefficients) */
#define NOV_P2C3R(NAME, DF, CS, CR)                                                                                        \
  inline NppStatus NAME(const Npp8u* const pSrc[2], int rSrcStep, Npp8u* pDst, int nDstStep, NppiSize roi, NppStreamContext ctx) { \
    NOV_FWD(VPF_FMT_NV12, DF, CS, CR, NOV(pSrc[0], rSrcStep, pSrc[1], rSrcStep), NOV(pDst, nDstStep));                     \
  }
NOV_P2C3R(nppiNV12ToRGB_709CSC_8u_P2C3R_Ctx, VPF_FMT_RGB, VPF_BT_709, VPF_MPEG)
NOV_P2C3R(nppiNV12ToRGB_709HDTV_8u_P2C3R_Ctx, VPF_FMT_RGB, VPF_BT_709, VPF_JPEG)
NOV_P2C3R(nppiNV12ToRGB_8u_P2C3R_Ctx, VPF_FMT_RGB, VPF_BT_601, VPF_JPEG)
NOV_P2C3R(nppiNV12ToBGR_709CSC_8u_P2C3R_Ctx, VPF_FMT_BGR, VPF_BT_709, VPF_MPEG)
NOV_P2C3R(nppiNV12ToBGR_709HDTV_8u_P2C3R_Ctx, VPF_FMT_BGR, VPF_BT_709, VPF_JPEG)
NOV_P2C3R(nppiNV12ToBGR_8u_P2C3R_Ctx, VPF_FMT_BGR, VPF_BT_601, VPF_JPEG)

/* ---- NV12 <-> YUV420 re-layout: nv12_yuv420 (:196-240), yuv420_nv12 (:945-975) */
inline NppStatus nppiNV12ToYUV420_8u_P2P3R_Ctx(const Npp8u* const pSrc[2], int nSrcStep, Npp8u* pDst[3], int aDstStep[3], NppiSize roi, NppStreamContext ctx) {
  NOV_FWD(VPF_FMT_NV12, VPF_FMT_YUV420, VPF_BT_601, VPF_JPEG, NOV(pSrc[0], nSrcStep, pSrc[1], nSrcStep), NOV(pDst[0], aDstStep[0], pDst[1], aDstStep[1], pDst[2], aDstStep[2]));
}
inline NppStatus nppiYCbCr420_8u_P2P3R_Ctx(const Npp8u* pSrcY, int nSrcYStep, const Npp8u* pSrcCbCr, int nSrcCbCrStep, Npp8u* pDst[3], int rDstStep[3], NppiSize roi,
                                           NppStreamContext ctx) {
  NOV_FWD(VPF_FMT_NV12, VPF_FMT_YUV420, VPF_BT_601, VPF_MPEG, NOV(pSrcY, nSrcYStep, pSrcCbCr, nSrcCbCrStep), NOV(pDst[0], rDstStep[0], pDst[1], rDstStep[1], pDst[2], rDstStep[2]));
}
inline NppStatus nppiYCbCr420_8u_P3P2R_Ctx(const Npp8u* const pSrc[3], int rSrcStep[3], Npp8u* pDstY, int nDstYStep, Npp8u* pDstCbCr, int nDstCbCrStep, NppiSize roi,
                                           NppStreamContext ctx) {
  NOV_FWD(VPF_FMT_YUV420, VPF_FMT_NV12, VPF_BT_601, VPF_MPEG, NOV(pSrc[0], rSrcStep[0], pSrc[1], rSrcStep[1], pSrc[2], rSrcStep[2]), NOV(pDstY, nDstYStep, pDstCbCr, nDstCbCrStep));
}

/* ---- planar 4:2:0 -> packed: yuv420_rgb / yuv420_bgr (:322-430).  "YUV420" = NPP's full-range YUV model, "YCbCr420" = limited range */
#define NOV_420_C3(NAME, DF, CR)                                                                                                   \
  inline NppStatus NAME(const Npp8u* const pSrc[3], int rSrcStep[3], Npp8u* pDst, int nDstStep, NppiSize roi, NppStreamContext ctx) {       \
    NOV_FWD(VPF_FMT_YUV420, DF, VPF_BT_601, CR, NOV(pSrc[0], rSrcStep[0], pSrc[1], rSrcStep[1], pSrc[2], rSrcStep[2]), NOV(pDst, nDstStep)); \
  }
NOV_420_C3(nppiYUV420ToRGB_8u_P3C3R_Ctx, VPF_FMT_RGB, VPF_JPEG)
NOV_420_C3(nppiYCbCr420ToRGB_8u_P3C3R_Ctx, VPF_FMT_RGB, VPF_MPEG)
NOV_420_C3(nppiYUV420ToBGR_8u_P3C3R_Ctx, VPF_FMT_BGR, VPF_JPEG)
NOV_420_C3(nppiYCbCr420ToBGR_8u_P3C3R_Ctx, VPF_FMT_BGR, VPF_MPEG)

/* ---- planar 4:4:4 -> packed / planar RGB: yuv444_bgr (:444-490), yuv444_rgb (:504-550), yuv444_rgb_planar (:564-612) */
#define NOV_444_C3(NAME, DF, CR)                                                                                              \
  inline NppStatus NAME(const Npp8u* const pSrc[3], int nSrcStep, Npp8u* pDst, int nDstStep, NppiSize roi, NppStreamContext ctx) {     \
    NOV_FWD(VPF_FMT_YUV444, DF, VPF_BT_601, CR, NOV(pSrc[0], nSrcStep, pSrc[1], nSrcStep, pSrc[2], nSrcStep), NOV(pDst, nDstStep));     \
  }
NOV_444_C3(nppiYCbCrToBGR_8u_P3C3R_Ctx, VPF_FMT_BGR, VPF_MPEG)
NOV_444_C3(nppiYUVToBGR_8u_P3C3R_Ctx, VPF_FMT_BGR, VPF_JPEG)
NOV_444_C3(nppiYUVToRGB_8u_P3C3R_Ctx, VPF_FMT_RGB, VPF_JPEG)
inline NppStatus nppiYUVToRGB_8u_P3R_Ctx(const Npp8u* const pSrc[3], int nSrcStep, Npp8u* pDst[3], int nDstStep, NppiSize roi, NppStreamContext ctx) {
  NOV_FWD(VPF_FMT_YUV444, VPF_FMT_RGB_PLANAR, VPF_BT_601, VPF_JPEG, NOV(pSrc[0], nSrcStep, pSrc[1], nSrcStep, pSrc[2], nSrcStep), NOV(pDst[0], nDstStep, pDst[1], nDstStep, pDst[2], nDstStep));
}

/* ---- packed RGB / BGR -> planar 4:4:4: bgr_yuv444 (:626-672), rgb_yuv444 (:731-772) */
#define NOV_C3_444(NAME, SF, CR)                                                                                            \
  inline NppStatus NAME(const Npp8u* pSrc, int nSrcStep, Npp8u* pDst[3], int nDstStep, NppiSize roi, NppStreamContext ctx) {          \
    NOV_FWD(SF, VPF_FMT_YUV444, VPF_BT_601, CR, NOV(pSrc, nSrcStep), NOV(pDst[0], nDstStep, pDst[1], nDstStep, pDst[2], nDstStep));    \
  }
NOV_C3_444(nppiBGRToYCbCr_8u_C3P3R_Ctx, VPF_FMT_BGR, VPF_MPEG)
NOV_C3_444(nppiBGRToYUV_8u_C3P3R_Ctx, VPF_FMT_BGR, VPF_JPEG)
NOV_C3_444(nppiRGBToYUV_8u_C3P3R_Ctx, VPF_FMT_RGB, VPF_JPEG)
inline NppStatus nppiRGBToYCbCr_8u_C3R_Ctx(const Npp8u*, int, Npp8u*, int, NppiSize, NppStreamContext) {  /* see the header comment */
  ref_hip_note(__func__, -1);
  return NPP_ERROR;
}
/* ---- planar RGB -> planar 4:4:4: rgb_planar_yuv444 (:786-830) */
#define NOV_P3_444(NAME, CR)                                                                                                      \
  inline NppStatus NAME(const Npp8u* const pSrc[3], int nSrcStep, Npp8u* pDst[3], int nDstStep, NppiSize roi, NppStreamContext ctx) {      \
    NOV_FWD(VPF_FMT_RGB_PLANAR, VPF_FMT_YUV444, VPF_BT_601, CR, NOV(pSrc[0], nSrcStep, pSrc[1], nSrcStep, pSrc[2], nSrcStep),              \
            NOV(pDst[0], nDstStep, pDst[1], nDstStep, pDst[2], nDstStep));                                                         \
  }
NOV_P3_444(nppiRGBToYUV_8u_P3R_Ctx, VPF_JPEG)
NOV_P3_444(nppiRGBToYCbCr_8u_P3R_Ctx, VPF_MPEG)

/* ---- packed RGB / BGR -> planar 4:2:0: rgb_yuv420 (:887-931), bgr_ycbcr (:686-717) */
#define NOV_C3_420(NAME, SF, DF, CR)                                                                                       \
  inline NppStatus NAME(const Npp8u* pSrc, int nSrcStep, Npp8u* pDst[3], int rDstStep[3], NppiSize roi, NppStreamContext ctx) {      \
    NOV_FWD(SF, DF, VPF_BT_601, CR, NOV(pSrc, nSrcStep), NOV(pDst[0], rDstStep[0], pDst[1], rDstStep[1], pDst[2], rDstStep[2]));      \
  }
NOV_C3_420(nppiRGBToYUV420_8u_C3P3R_Ctx, VPF_FMT_RGB, VPF_FMT_YUV420, VPF_JPEG)
NOV_C3_420(nppiRGBToYCbCr420_8u_C3P3R_Ctx, VPF_FMT_RGB, VPF_FMT_YUV420, VPF_MPEG)
NOV_C3_420(nppiBGRToYCbCr420_8u_C3P3R_Ctx, VPF_FMT_BGR, VPF_FMT_YCBCR, VPF_MPEG)

/* ---- gray, re-layouts, float: rbg8_y (:293-308), rgb8_deinterleave (:1059-1088), rgb8_interleave (:1102-1131), rgb_bgr / bgr_rgb
 * (:1145-1209), rbg8_rgb32f (:1222-1254), rgb32f_deinterleave (:1268-1297) */
inline NppStatus nppiRGBToGray_8u_C3C1R_Ctx(const Npp8u* pSrc, int nSrcStep, Npp8u* pDst, int nDstStep, NppiSize roi, NppStreamContext ctx) {
  NOV_FWD(VPF_FMT_RGB, VPF_FMT_Y, VPF_BT_601, VPF_MPEG, NOV(pSrc, nSrcStep), NOV(pDst, nDstStep));
}
inline NppStatus nppiCopy_8u_C3P3R_Ctx(const Npp8u* pSrc, int nSrcStep, Npp8u* const aDst[3], int nDstStep, NppiSize roi, NppStreamContext ctx) {
  NOV_FWD(VPF_FMT_RGB, VPF_FMT_RGB_PLANAR, VPF_BT_601, VPF_MPEG, NOV(pSrc, nSrcStep), NOV(aDst[0], nDstStep, aDst[1], nDstStep, aDst[2], nDstStep));
}
inline NppStatus nppiCopy_8u_P3C3R_Ctx(const Npp8u* const aSrc[3], int nSrcStep, Npp8u* pDst, int nDstStep, NppiSize roi, NppStreamContext ctx) {
  NOV_FWD(VPF_FMT_RGB_PLANAR, VPF_FMT_RGB, VPF_BT_601, VPF_MPEG, NOV(aSrc[0], nSrcStep, aSrc[1], nSrcStep, aSrc[2], nSrcStep), NOV(pDst, nDstStep));
}
inline NppStatus nppiSwapChannels_8u_C3R_Ctx(const Npp8u* pSrc, int nSrcStep, Npp8u* pDst, int nDstStep, NppiSize roi, const int aDstOrder[3], NppStreamContext ctx) {
  if (aDstOrder[0] != 2 || aDstOrder[1] != 1 || aDstOrder[2] != 0) { ref_hip_note(__func__, -1); return NPP_ERROR; }  /* the only order the reference uses */
  NOV_FWD(VPF_FMT_RGB, VPF_FMT_BGR, VPF_BT_601, VPF_MPEG, NOV(pSrc, nSrcStep), NOV(pDst, nDstStep));
}
inline NppStatus nppiScale_8u32f_C3R_Ctx(const Npp8u* pSrc, int nSrcStep, Npp32f* pDst, int nDstStep, NppiSize roi, Npp32f nMin, Npp32f nMax, NppStreamContext ctx) {
  if (nMin != 0.0f || nMax != 1.0f) { ref_hip_note(__func__, -1); return NPP_ERROR; }
  NOV_FWD(VPF_FMT_RGB, VPF_FMT_RGB_32F, VPF_BT_601, VPF_MPEG, NOV(pSrc, nSrcStep), NOV(pDst, nDstStep));
}
inline NppStatus nppiCopy_32f_C3P3R_Ctx(const Npp32f* pSrc, int nSrcStep, Npp32f* const aDst[3], int nDstStep, NppiSize roi, NppStreamContext ctx) {
  NOV_FWD(VPF_FMT_RGB_32F, VPF_FMT_RGB_32F_PLANAR, VPF_BT_601, VPF_MPEG, NOV(pSrc, nSrcStep), NOV(aDst[0], nDstStep, aDst[1], nDstStep, aDst[2], nDstStep));
}

/* ---- y_yuv444 (:836-880) is two fills and one plane copy: plain 2-D data movement on the stream */
inline NppStatus nppiSet_8u_C1R_Ctx(Npp8u nValue, Npp8u* pDst, int nDstStep, NppiSize roi, NppStreamContext ctx) {
  return ref_hip_set2d(pDst, nDstStep, nValue, roi.width, roi.height, (void*)ctx.hStream) ? NPP_ERROR : NPP_NO_ERROR;
}
inline NppStatus nppiCopy_8u_C1R_Ctx(const Npp8u* pSrc, int nSrcStep, Npp8u* pDst, int nDstStep, NppiSize roi, NppStreamContext ctx) {
  return ref_hip_copy2d(pSrc, nSrcStep, pDst, nDstStep, roi.width, roi.height, (void*)ctx.hStream) ? NPP_ERROR : NPP_NO_ERROR;
}
/* ---- p16_nv12 (:985-1045): not forwarded, see the header comment */
inline NppStatus nppiDivC_16u_C1RSfs_Ctx(const Npp16u*, int, Npp16u, Npp16u*, int, NppiSize, int, NppStreamContext) { ref_hip_note(__func__, -1); return NPP_ERROR; }
inline NppStatus nppiConvert_16u8u_C1R_Ctx(const Npp16u*, int, Npp8u*, int, NppiSize, NppStreamContext) { ref_hip_note(__func__, -1); return NPP_ERROR; }

/* ---- resize / remap (round 3): the reference's ResizeSurface / RemapSurface (src/TC/src/Tasks.cpp:1162-1203 packed 3C, :1217-1261
 * one C1R call per plane, :1285-1324 the NV12 chain C3 -> R2 -> C4, :1555-1602 remap) call these with NPPI_INTER_LANCZOS / NPPI_INTER_LINEAR.
 * NPP's argument lists (nppi_geometry_transforms.h); the ROIs the reference passes are always the whole image (checked). */
namespace npp_over_vpf {
inline int interp_of(int e) { return e == NPPI_INTER_LANCZOS ? VPF_INTERP_LANCZOS3 : e == NPPI_INTER_LINEAR ? VPF_INTERP_LINEAR : e == NPPI_INTER_NN ? VPF_INTERP_NEAREST : -1; }
inline bool whole(NppiSize s, NppiRect r) { return r.x == 0 && r.y == 0 && r.width == s.width && r.height == s.height; }
inline NppStatus resize(const char* name, int fmt, const void* pSrc, int nSrcStep, NppiSize ss, NppiRect sr, void* pDst, int nDstStep, NppiSize ds, NppiRect dr,
                        int eInterpolation, const NppStreamContext& ctx) {
  const int interp = interp_of(eInterpolation);
  if (interp < 0 || !whole(ss, sr) || !whole(ds, dr)) { ref_hip_note(name, -1); return NPP_ERROR; }
  const vpf_exec ex = {-1, 0, (void*)ctx.hStream};
  const P3 s = planes(pSrc, nSrcStep), d = planes(pDst, nDstStep);
  const vpf_status st = vpf_resize(&ex, fmt, interp, vpf_size{(uint32_t)ss.width, (uint32_t)ss.height}, s.p, vpf_size{(uint32_t)ds.width, (uint32_t)ds.height}, d.p);
  ref_hip_note(name, (int)st);
  return st == VPF_OK ? NPP_NO_ERROR : NPP_ERROR;
}
}  // namespace npp_over_vpf
inline NppStatus nppiResize_8u_C3R_Ctx(const Npp8u* pSrc, int nSrcStep, NppiSize oSrcSize, NppiRect oSrcRectROI, Npp8u* pDst, int nDstStep, NppiSize oDstSize,
                                       NppiRect oDstRectROI, int eInterpolation, NppStreamContext ctx) {
  return npp_over_vpf::resize(__func__, VPF_FMT_RGB, pSrc, nSrcStep, oSrcSize, oSrcRectROI, pDst, nDstStep, oDstSize, oDstRectROI, eInterpolation, ctx);
}
inline NppStatus nppiResize_8u_C1R_Ctx(const Npp8u* pSrc, int nSrcStep, NppiSize oSrcSize, NppiRect oSrcRectROI, Npp8u* pDst, int nDstStep, NppiSize oDstSize,
                                       NppiRect oDstRectROI, int eInterpolation, NppStreamContext ctx) {
  return npp_over_vpf::resize(__func__, VPF_FMT_Y, pSrc, nSrcStep, oSrcSize, oSrcRectROI, pDst, nDstStep, oDstSize, oDstRectROI, eInterpolation, ctx);
}
inline NppStatus nppiResize_32f_C3R_Ctx(const Npp32f* pSrc, int nSrcStep, NppiSize oSrcSize, NppiRect oSrcRectROI, Npp32f* pDst, int nDstStep, NppiSize oDstSize,
                                        NppiRect oDstRectROI, int eInterpolation, NppStreamContext ctx) {
  return npp_over_vpf::resize(__func__, VPF_FMT_RGB_32F, pSrc, nSrcStep, oSrcSize, oSrcRectROI, pDst, nDstStep, oDstSize, oDstRectROI, eInterpolation, ctx);
}
/* one float plane at a time: vpf_resize takes a whole RGB_32F_PLANAR surface (three planes) — not forwarded */
inline NppStatus nppiResize_32f_C1R_Ctx(const Npp32f*, int, NppiSize, NppiRect, Npp32f*, int, NppiSize, NppiRect, int, NppStreamContext) { ref_hip_note(__func__, -1); return NPP_ERROR; }
inline NppStatus nppiRemap_8u_C3R_Ctx(const Npp8u* pSrc, NppiSize oSrcSize, int nSrcStep, NppiRect oSrcROI, const Npp32f* pXMap, int nXMapStep, const Npp32f* pYMap,
                                      int nYMapStep, Npp8u* pDst, int nDstStep, NppiSize oDstSizeROI, int eInterpolation, NppStreamContext ctx) {
  if (eInterpolation != NPPI_INTER_LINEAR || !npp_over_vpf::whole(oSrcSize, oSrcROI)) { ref_hip_note(__func__, -1); return NPP_ERROR; }
  const vpf_exec ex = {-1, 0, (void*)ctx.hStream};
  const vpf_plane s = {const_cast<Npp8u*>(pSrc), (uint32_t)nSrcStep, 0}, d = {pDst, (uint32_t)nDstStep, 0};
  const vpf_status st = vpf_remap(&ex, VPF_FMT_RGB, vpf_size{(uint32_t)oSrcSize.width, (uint32_t)oSrcSize.height}, &s, pXMap, (uint32_t)nXMapStep, pYMap, (uint32_t)nYMapStep,
                                  vpf_size{(uint32_t)oDstSizeROI.width, (uint32_t)oDstSizeROI.height}, &d);
  ref_hip_note(__func__, (int)st);
  return st == VPF_OK ? NPP_NO_ERROR : NPP_ERROR;
}

#undef NOV_FWD
#undef NOV
