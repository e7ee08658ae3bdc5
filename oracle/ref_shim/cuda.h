/* oracle/ref_shim/cuda.h — TEST INFRASTRUCTURE.  A host-memory stand-in for the handful of CUDA driver entry points the
 * reference's MemoryInterfaces.cpp / TasksColorCvt.cpp use, so that those reference sources (compiled where they lie,
 * never copied: oracle/Makefile `ref_tc`) run on a machine with no CUDA: "device" memory is malloc'd host memory.
 * Written from the CUDA driver API's public signatures; nothing here comes from the reference tree. */
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifndef CUDAAPI
#define CUDAAPI
#endif
#ifdef __cplusplus
extern "C" {
#endif
typedef unsigned long long CUdeviceptr;
typedef struct CUctx_st* CUcontext;
typedef struct CUstream_st* CUstream;
typedef int CUdevice;
typedef enum { CUDA_SUCCESS = 0, CUDA_ERROR_INVALID_VALUE = 1, CUDA_ERROR_OUT_OF_MEMORY = 2 } CUresult;
typedef enum { CU_MEMORYTYPE_HOST = 1, CU_MEMORYTYPE_DEVICE = 2, CU_MEMORYTYPE_ARRAY = 3, CU_MEMORYTYPE_UNIFIED = 4 } CUmemorytype;
typedef enum { CU_POINTER_ATTRIBUTE_CONTEXT = 1 } CUpointer_attribute;
typedef struct CUarray_st* CUarray;
typedef struct {
  size_t srcXInBytes, srcY;
  CUmemorytype srcMemoryType;
  const void* srcHost;
  CUdeviceptr srcDevice;
  CUarray srcArray;
  size_t srcPitch;
  size_t dstXInBytes, dstY;
  CUmemorytype dstMemoryType;
  void* dstHost;
  CUdeviceptr dstDevice;
  CUarray dstArray;
  size_t dstPitch;
  size_t WidthInBytes, Height;
} CUDA_MEMCPY2D;
CUresult cuGetErrorName(CUresult, const char**);
CUresult cuGetErrorString(CUresult, const char**);
CUresult cuMemAllocHost(void**, size_t);
CUresult cuMemFreeHost(void*);
CUresult cuMemAlloc(CUdeviceptr*, size_t);
CUresult cuMemAllocPitch(CUdeviceptr*, size_t* pitch, size_t width_bytes, size_t height, unsigned int elem_size);
CUresult cuMemFree(CUdeviceptr);
CUresult cuMemcpyDtoD(CUdeviceptr, CUdeviceptr, size_t);
CUresult cuMemcpyHtoDAsync(CUdeviceptr, const void*, size_t, CUstream);
CUresult cuMemcpyDtoHAsync(void*, CUdeviceptr, size_t, CUstream);
CUresult cuMemcpy2DAsync(const CUDA_MEMCPY2D*, CUstream);
CUresult cuStreamSynchronize(CUstream);
CUresult cuCtxPushCurrent(CUcontext);
CUresult cuCtxPopCurrent(CUcontext*);
CUresult cuPointerGetAttribute(void*, CUpointer_attribute, CUdeviceptr);
#ifdef __cplusplus
}
#endif
