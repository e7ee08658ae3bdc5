// oracle/ref_tc_shim.cpp — TEST INFRASTRUCTURE.  C entry points over the REFERENCE'S OWN Surface classes
// (src/TC/src/MemoryInterfaces.cpp) and converter dispatch (src/TC/src/TasksColorCvt.cpp), which oracle/Makefile
// (`ref_tc`) compiles from /root/reference where they lie against the stand-in headers in oracle/ref_shim/:
//   * "device" memory is host memory (ref_shim/cuda.h, implemented below),
//   * NPP entry points record their names instead of computing pixels (ref_shim/fake_npp.h).
// What this pins: surface geometry per pixel format, and WHICH NPP function (= which colour model) the reference selects
// or refuses for every (src format, dst format, colour space, colour range).  What it cannot pin: NPP's arithmetic.
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

// ---------------------------------------------------------------------------------------- CUDA stand-ins (host memory)
extern "C" {
CUresult cuGetErrorName(CUresult, const char** s) { *s = "CUDA_ERROR(stub)"; return CUDA_SUCCESS; }
CUresult cuGetErrorString(CUresult, const char** s) { *s = "stub"; return CUDA_SUCCESS; }
CUresult cuMemAllocHost(void** p, size_t n) { *p = std::calloc(n ? n : 1, 1); return *p ? CUDA_SUCCESS : CUDA_ERROR_OUT_OF_MEMORY; }
CUresult cuMemFreeHost(void* p) { std::free(p); return CUDA_SUCCESS; }
CUresult cuMemAlloc(CUdeviceptr* p, size_t n) { *p = (CUdeviceptr)(uintptr_t)std::calloc(n ? n : 1, 1); return *p ? CUDA_SUCCESS : CUDA_ERROR_OUT_OF_MEMORY; }
CUresult cuMemAllocPitch(CUdeviceptr* p, size_t* pitch, size_t wb, size_t h, unsigned int) {
  *pitch = (wb + 511) / 512 * 512;  // what current NVIDIA drivers return; only ratios of it are compared
  return cuMemAlloc(p, *pitch * (h ? h : 1));
}
CUresult cuMemFree(CUdeviceptr p) { std::free((void*)(uintptr_t)p); return CUDA_SUCCESS; }
CUresult cuMemcpyDtoD(CUdeviceptr d, CUdeviceptr s, size_t n) { std::memcpy((void*)(uintptr_t)d, (const void*)(uintptr_t)s, n); return CUDA_SUCCESS; }
CUresult cuMemcpyHtoDAsync(CUdeviceptr d, const void* s, size_t n, CUstream) { std::memcpy((void*)(uintptr_t)d, s, n); return CUDA_SUCCESS; }
CUresult cuMemcpy2DAsync(const CUDA_MEMCPY2D* m, CUstream) {
  const uint8_t* s = m->srcMemoryType == CU_MEMORYTYPE_HOST ? (const uint8_t*)m->srcHost : (const uint8_t*)(uintptr_t)m->srcDevice;
  uint8_t* d = m->dstMemoryType == CU_MEMORYTYPE_HOST ? (uint8_t*)m->dstHost : (uint8_t*)(uintptr_t)m->dstDevice;
  for (size_t r = 0; r < m->Height; r++)
    std::memcpy(d + (m->dstY + r) * m->dstPitch + m->dstXInBytes, s + (m->srcY + r) * m->srcPitch + m->srcXInBytes, m->WidthInBytes);
  return CUDA_SUCCESS;
}
CUresult cuStreamSynchronize(CUstream) { return CUDA_SUCCESS; }
CUresult cuCtxPushCurrent(CUcontext) { return CUDA_SUCCESS; }
CUresult cuCtxPopCurrent(CUcontext*) { return CUDA_SUCCESS; }
CUresult cuPointerGetAttribute(void* out, CUpointer_attribute, CUdeviceptr) { *(CUcontext*)out = nullptr; return CUDA_SUCCESS; }
}

// ---------------------------------------------------------------------------------------- NPP stand-ins
static thread_local std::string g_npp_log;
extern "C" NppStatus ref_npp_record(const char* name) {
  if (!g_npp_log.empty()) g_npp_log += ",";
  g_npp_log += name;
  return NPP_NO_ERROR;
}
void SetupNppContext(CUcontext, CUstream stream, NppStreamContext& ctx) {  // the reference's lives in NppCommon.cpp (cudaGetDeviceProperties ...)
  std::memset(&ctx, 0, sizeof(ctx));
  ctx.hStream = stream;
}

// ---------------------------------------------------------------------------------------- probes
extern "C" {

// out[0] = NumPlanes, out[1] = HostMemSize, out[2] = ElemSize, then per plane p: Width(p), Height(p), Pitch(p),
// WidthInBytes(p), PlanePtr(p) - PlanePtr(0), GetSurfacePlane(p)->Width(), ->Height().  Returns 0, or -1 if Make() yields nothing / throws.
int ref_surface_geometry(int fmt, uint32_t w, uint32_t h, int64_t* out, int cap) {
  try {
    std::unique_ptr<Surface> s(Surface::Make((Pixel_Format)fmt, w, h, nullptr));
    if (!s) return -1;
    const int n = (int)s->NumPlanes();
    if (cap < 3 + 7 * n) return -2;
    out[0] = n; out[1] = s->HostMemSize(); out[2] = s->ElemSize();
    for (int p = 0; p < n; p++) {
      int64_t* o = out + 3 + 7 * p;
      o[0] = s->Width(p); o[1] = s->Height(p); o[2] = s->Pitch(p); o[3] = s->WidthInBytes(p);
      o[4] = (int64_t)(s->PlanePtr(p) - s->PlanePtr(0));
      SurfacePlane* sp = s->GetSurfacePlane(p);
      o[5] = sp ? sp->Width() : -1; o[6] = sp ? sp->Height() : -1;
    }
    return 0;
  } catch (std::exception&) {
    return -1;
  }
}

// Runs the reference's ConvertSurface for one configuration.  cs / cr < 0: no ColorspaceConversionContext input at all.
// Returns -1 when the constructor throws (unsupported pair), 0 when Execute yields no output surface (refused
// combination), 1 on success; `log` receives the comma-separated NPP entry points that were called.
int ref_convert_probe(int in_fmt, int out_fmt, uint32_t w, uint32_t h, int cs, int cr, char* log, int cap, int* out_fmt_seen) {
  g_npp_log.clear();
  if (log && cap) log[0] = 0;
  try {
    std::unique_ptr<ConvertSurface> conv(ConvertSurface::Make(w, h, (Pixel_Format)in_fmt, (Pixel_Format)out_fmt, nullptr, nullptr));
    std::unique_ptr<Surface> src(Surface::Make((Pixel_Format)in_fmt, w, h, nullptr));
    if (!conv || !src) return -1;
    std::unique_ptr<Buffer> ctx_buf(Buffer::MakeOwnMem(sizeof(ColorspaceConversionContext)));
    conv->ClearInputs();
    conv->SetInput(src.get(), 0U);
    if (cs >= 0 && cr >= 0) {
      ColorspaceConversionContext cc((ColorSpace)cs, (ColorRange)cr);
      ctx_buf->CopyFrom(sizeof(cc), &cc);
      conv->SetInput((Token*)ctx_buf.get(), 1U);
    }
    const auto st = conv->Execute();
    auto* out = (Surface*)conv->GetOutput(0U);
    if (log && cap) std::snprintf(log, cap, "%s", g_npp_log.c_str());
    if (out && out_fmt_seen) *out_fmt_seen = (int)out->PixelFormat();
    return (st == TaskExecStatus::TASK_EXEC_SUCCESS && out) ? 1 : 0;
  } catch (std::exception& e) {
    if (log && cap) std::snprintf(log, cap, "EXC:%s", e.what());
    return -1;
  }
}
}
