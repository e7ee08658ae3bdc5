// oracle/ref_tc_hip_shim.cpp — TEST INFRASTRUCTURE.  The reference's own converter Task layer on an MI355X.
//
// oracle/Makefile `ref_tc_hip` compiles the reference's MemoryInterfaces.cpp (Surface classes) and TasksColorCvt.cpp (all 24
// converter impls + the ConvertSurface dispatch) UNMODIFIED from /root/reference, where they lie, against
//   * ref_shim/cuda.h, whose handful of CUDA driver entry points are implemented below over the HIP runtime (device memory is
//     real HBM: cuMemAllocPitch -> hipMalloc with a 256-B pitch, cuMemcpy2DAsync -> hipMemcpy2DAsync, ...), and
//   * ref_shim_hip/npp_over_vpf.h, where every nppi*_Ctx the reference calls forwards to libvpfhip's C ABI (vpf_convert).
// ref_hip_convert() below then drives the reference's ConvertSurface exactly as PySurfaceConverter::Execute does
// (src/PyNvCodec/src/PySurfaceConverter.cpp:50-74).  tests/test_gpu_reference_caller.py compares its pixels with the oracle.
// Nothing in the product links or loads this file.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <string>

#include "MemoryInterfaces.hpp"
#include "NppCommon.hpp"
#include "Tasks.hpp"

using namespace VPF;

static thread_local std::string g_log;
static thread_local int g_vpf_calls = 0;

extern "C" {
// ---------------------------------------------------------------------------------------- CUDA driver API over HIP
CUresult cuGetErrorName(CUresult, const char** s) { *s = "CUDA_ERROR(hip shim)"; return CUDA_SUCCESS; }
CUresult cuGetErrorString(CUresult, const char** s) { *s = "hip shim"; return CUDA_SUCCESS; }
CUresult cuMemAllocHost(void** p, size_t n) { return hipHostMalloc(p, n ? n : 1, hipHostMallocDefault) == hipSuccess ? CUDA_SUCCESS : CUDA_ERROR_OUT_OF_MEMORY; }
CUresult cuMemFreeHost(void* p) { (void)hipHostFree(p); return CUDA_SUCCESS; }
CUresult cuMemAlloc(CUdeviceptr* p, size_t n) {
  void* d = nullptr;
  if (hipMalloc(&d, n ? n : 1) != hipSuccess) return CUDA_ERROR_OUT_OF_MEMORY;
  *p = (CUdeviceptr)(uintptr_t)d;
  return CUDA_SUCCESS;
}
CUresult cuMemAllocPitch(CUdeviceptr* p, size_t* pitch, size_t wb, size_t h, unsigned int) {
  *pitch = (wb + 255) / 256 * 256;
  return cuMemAlloc(p, *pitch * (h ? h : 1));
}
CUresult cuMemFree(CUdeviceptr p) { (void)hipFree((void*)(uintptr_t)p); return CUDA_SUCCESS; }
CUresult cuMemcpyDtoD(CUdeviceptr d, CUdeviceptr s, size_t n) {
  return hipMemcpy((void*)(uintptr_t)d, (const void*)(uintptr_t)s, n, hipMemcpyDeviceToDevice) == hipSuccess ? CUDA_SUCCESS : CUDA_ERROR_INVALID_VALUE;
}
CUresult cuMemcpyHtoDAsync(CUdeviceptr d, const void* s, size_t n, CUstream st) {
  return hipMemcpyAsync((void*)(uintptr_t)d, s, n, hipMemcpyHostToDevice, (hipStream_t)st) == hipSuccess ? CUDA_SUCCESS : CUDA_ERROR_INVALID_VALUE;
}
CUresult cuMemcpy2DAsync(const CUDA_MEMCPY2D* m, CUstream st) {
  const bool sh = m->srcMemoryType == CU_MEMORYTYPE_HOST, dh = m->dstMemoryType == CU_MEMORYTYPE_HOST;
  const uint8_t* s = (sh ? (const uint8_t*)m->srcHost : (const uint8_t*)(uintptr_t)m->srcDevice) + m->srcY * m->srcPitch + m->srcXInBytes;
  uint8_t* d = (dh ? (uint8_t*)m->dstHost : (uint8_t*)(uintptr_t)m->dstDevice) + m->dstY * m->dstPitch + m->dstXInBytes;
  const hipMemcpyKind k = sh ? (dh ? hipMemcpyHostToHost : hipMemcpyHostToDevice) : (dh ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
  return hipMemcpy2DAsync(d, m->dstPitch, s, m->srcPitch, m->WidthInBytes, m->Height, k, (hipStream_t)st) == hipSuccess ? CUDA_SUCCESS : CUDA_ERROR_INVALID_VALUE;
}
CUresult cuStreamSynchronize(CUstream st) { return hipStreamSynchronize((hipStream_t)st) == hipSuccess ? CUDA_SUCCESS : CUDA_ERROR_INVALID_VALUE; }
CUresult cuCtxPushCurrent(CUcontext) { return CUDA_SUCCESS; }  // no contexts on ROCm: the current device is the context
CUresult cuCtxPopCurrent(CUcontext*) { return CUDA_SUCCESS; }
CUresult cuPointerGetAttribute(void* out, CUpointer_attribute, CUdeviceptr) { *(CUcontext*)out = nullptr; return CUDA_SUCCESS; }

int ref_hip_copy2d(const void* src, int sstep, void* dst, int dstep, int wb, int rows, void* stream) {
  return hipMemcpy2DAsync(dst, dstep, src, sstep, wb, rows, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : 1;
}
int ref_hip_set2d(void* dst, int dstep, int value, int wb, int rows, void* stream) {
  return hipMemset2DAsync(dst, dstep, value, wb, rows, (hipStream_t)stream) == hipSuccess ? 0 : 1;
}
void ref_hip_note(const char* npp_name, int vpf_status) {
  if (!g_log.empty()) g_log += ",";
  g_log += npp_name;
  g_log += vpf_status == 0 ? ":ok" : (vpf_status < 0 ? ":not-forwarded" : ":vpf_status=" + std::to_string(vpf_status));
  if (vpf_status >= 0) g_vpf_calls++;
}
}

void SetupNppContext(CUcontext, CUstream stream, NppStreamContext& ctx) {  // the reference's lives in NppCommon.cpp (cudaGetDeviceProperties ...)
  std::memset(&ctx, 0, sizeof(ctx));
  ctx.hStream = stream;
}

extern "C" {
// One conversion through the REFERENCE'S ConvertSurface on the GPU.  `src` / `dst` are tight host frames (planes concatenated at
// tight width, the layout of CudaUploadFrame / CudaDownloadSurface, Tasks.cpp:643-658,815-854).  cs / cr < 0: no context token.
// Returns -1 ctor threw (unsupported pair), 0 refused / failed (no output surface), 1 ok, -3 host buffers too small, -4 HIP error.
// `log` receives "nppiName:ok,..." for every NPP-named adapter the reference called; *vpf_calls = how many reached vpf_convert.
int ref_hip_convert(int in_fmt, int out_fmt, uint32_t w, uint32_t h, int cs, int cr, const uint8_t* src, size_t src_bytes, uint8_t* dst,
                    size_t dst_cap, size_t* dst_bytes, char* log, int cap, int* vpf_calls) {
  g_log.clear();
  g_vpf_calls = 0;
  if (log && cap) log[0] = 0;
  int rc = 0;
  try {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -4;
    {
      std::unique_ptr<ConvertSurface> conv(ConvertSurface::Make(w, h, (Pixel_Format)in_fmt, (Pixel_Format)out_fmt, nullptr, (CUstream)st));
      std::unique_ptr<Surface> in(Surface::Make((Pixel_Format)in_fmt, w, h, nullptr));
      if (!conv || !in) { (void)hipStreamDestroy(st); return -1; }
      if (in->HostMemSize() > src_bytes) { (void)hipStreamDestroy(st); return -3; }
      size_t off = 0;
      for (uint32_t p = 0; p < in->NumPlanes(); p++) {  // upload plane by plane, like CudaUploadFrame::Run
        const size_t wb = in->WidthInBytes(p), rows = in->Height(p);
        if (hipMemcpy2DAsync((void*)(uintptr_t)in->PlanePtr(p), in->Pitch(p), src + off, wb, wb, rows, hipMemcpyHostToDevice, st) != hipSuccess) rc = -4;
        off += wb * rows;
      }
      std::unique_ptr<Buffer> ctx_buf(Buffer::MakeOwnMem(sizeof(ColorspaceConversionContext)));
      conv->ClearInputs();
      conv->SetInput(in.get(), 0U);
      if (cs >= 0 && cr >= 0) {
        ColorspaceConversionContext cc((ColorSpace)cs, (ColorRange)cr);
        ctx_buf->CopyFrom(sizeof(cc), &cc);
        conv->SetInput((Token*)ctx_buf.get(), 1U);
      }
      const auto status = conv->Execute();
      auto* out = (Surface*)conv->GetOutput(0U);
      if (rc == 0 && status == TaskExecStatus::TASK_EXEC_SUCCESS && out) {
        if (out->HostMemSize() > dst_cap) {
          rc = -3;
        } else {
          off = 0;
          for (uint32_t p = 0; p < out->NumPlanes(); p++) {
            const size_t wb = out->WidthInBytes(p), rows = out->Height(p);
            if (hipMemcpy2DAsync(dst + off, wb, (const void*)(uintptr_t)out->PlanePtr(p), out->Pitch(p), wb, rows, hipMemcpyDeviceToHost, st) != hipSuccess) rc = -4;
            off += wb * rows;
          }
          if (dst_bytes) *dst_bytes = off;
          if (rc == 0) rc = 1;
        }
      }
      if (hipStreamSynchronize(st) != hipSuccess) rc = -4;
    }
    (void)hipStreamDestroy(st);
  } catch (std::exception& e) {
    if (log && cap) std::snprintf(log, cap, "EXC:%s", e.what());
    return -1;
  }
  if (log && cap) std::snprintf(log, cap, "%s", g_log.c_str());
  if (vpf_calls) *vpf_calls = g_vpf_calls;
  return rc;
}
}
