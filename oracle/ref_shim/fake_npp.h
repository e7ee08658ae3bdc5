/* oracle/ref_shim/fake_npp.h — TEST INFRASTRUCTURE.  Recording stand-ins for the NPP entry points that the reference's
 * TasksColorCvt.cpp calls.  No pixel is computed: every call appends its own name to a per-thread log and returns
 * NPP_NO_ERROR, so a test can ask the REFERENCE'S OWN dispatch code "which NPP function do you select for
 * (src format, dst format, colour space, colour range)?" and compare that with this repo's converter table.
 * Types follow NPP's public headers; the functions are variadic templates so no NPP prototype is restated. */
#pragma once
#include <stdint.h>
#include "cuda.h"
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
extern "C" NppStatus ref_npp_record(const char* name);
#define VPF_FAKE_NPP(name) \
  template <class... A> inline NppStatus name(A&&...) { return ref_npp_record(#name); }
VPF_FAKE_NPP(nppiBGRToYCbCr420_8u_C3P3R_Ctx) VPF_FAKE_NPP(nppiBGRToYCbCr_8u_C3P3R_Ctx) VPF_FAKE_NPP(nppiBGRToYUV_8u_C3P3R_Ctx)
VPF_FAKE_NPP(nppiConvert_16u8u_C1R_Ctx) VPF_FAKE_NPP(nppiCopy_32f_C3P3R_Ctx) VPF_FAKE_NPP(nppiCopy_8u_C1R_Ctx)
VPF_FAKE_NPP(nppiCopy_8u_C3P3R_Ctx) VPF_FAKE_NPP(nppiCopy_8u_P3C3R_Ctx) VPF_FAKE_NPP(nppiDivC_16u_C1RSfs_Ctx)
VPF_FAKE_NPP(nppiNV12ToBGR_709CSC_8u_P2C3R_Ctx) VPF_FAKE_NPP(nppiNV12ToBGR_709HDTV_8u_P2C3R_Ctx) VPF_FAKE_NPP(nppiNV12ToBGR_8u_P2C3R_Ctx)
VPF_FAKE_NPP(nppiNV12ToRGB_709CSC_8u_P2C3R_Ctx) VPF_FAKE_NPP(nppiNV12ToRGB_709HDTV_8u_P2C3R_Ctx) VPF_FAKE_NPP(nppiNV12ToRGB_8u_P2C3R_Ctx)
VPF_FAKE_NPP(nppiNV12ToYUV420_8u_P2P3R_Ctx) VPF_FAKE_NPP(nppiRGBToGray_8u_C3C1R_Ctx) VPF_FAKE_NPP(nppiRGBToYCbCr420_8u_C3P3R_Ctx)
VPF_FAKE_NPP(nppiRGBToYCbCr_8u_C3R_Ctx) VPF_FAKE_NPP(nppiRGBToYCbCr_8u_P3R_Ctx) VPF_FAKE_NPP(nppiRGBToYUV420_8u_C3P3R_Ctx)
VPF_FAKE_NPP(nppiRGBToYUV_8u_C3P3R_Ctx) VPF_FAKE_NPP(nppiRGBToYUV_8u_P3R_Ctx) VPF_FAKE_NPP(nppiScale_8u32f_C3R_Ctx)
VPF_FAKE_NPP(nppiSet_8u_C1R_Ctx) VPF_FAKE_NPP(nppiSwapChannels_8u_C3R_Ctx) VPF_FAKE_NPP(nppiYCbCr420ToBGR_8u_P3C3R_Ctx)
VPF_FAKE_NPP(nppiYCbCr420ToRGB_8u_P3C3R_Ctx) VPF_FAKE_NPP(nppiYCbCr420_8u_P2P3R_Ctx) VPF_FAKE_NPP(nppiYCbCr420_8u_P3P2R_Ctx)
VPF_FAKE_NPP(nppiYCbCrToBGR_8u_P3C3R_Ctx) VPF_FAKE_NPP(nppiYUV420ToBGR_8u_P3C3R_Ctx) VPF_FAKE_NPP(nppiYUV420ToRGB_8u_P3C3R_Ctx)
VPF_FAKE_NPP(nppiYUVToBGR_8u_P3C3R_Ctx) VPF_FAKE_NPP(nppiYUVToRGB_8u_P3C3R_Ctx) VPF_FAKE_NPP(nppiYUVToRGB_8u_P3R_Ctx)
