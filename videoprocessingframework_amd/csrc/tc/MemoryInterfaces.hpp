// MemoryInterfaces.hpp — host buffers, device buffers, pitched planes and Surfaces on HIP.
//
// Keeps the class surface of the reference's src/TC/inc/MemoryInterfaces.hpp (Pixel_Format values
// :30-49, ColorSpace/ColorRange :51-61, ColorspaceConversionContext :63-71, Buffer :76-116,
// CudaBuffer :118-150, SurfacePlane :175-295, Surface :300-384) because that is the drop-in
// contract of the Task layer, but is a different design underneath:
//   * one data-driven Surface implementation over a per-format geometry table instead of one
//     hand-written class per pixel format (the reference has 17);
//   * no contexts: ROCm has none worth the name.  `HipContext` is an opaque cookie that only encodes
//     a device ordinal (HipResMgr::GetCtx), so the reference's (context, stream) signatures survive;
//   * device memory comes from a pluggable allocator (default hipMalloc; a PyTorch caching-allocator
//     adapter or, in CPU-only unit tests, plain host memory can be plugged in);
//   * planes are allocated with a pitch that is a multiple of 256 B so every row start is
//     dwordx4-aligned for the gfx950 kernels (the reference takes cuMemAllocPitch's 512 B/arch default).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>

#include "TC_CORE.hpp"

namespace VPF {

enum Pixel_Format {
  UNDEFINED = 0, Y = 1, RGB = 2, NV12 = 3, YUV420 = 4, RGB_PLANAR = 5, BGR = 6, YCBCR = 7, YUV444 = 8,
  RGB_32F = 9, RGB_32F_PLANAR = 10, YUV422 = 11, P10 = 12, P12 = 13, YUV444_10bit = 14,
  YUV420_10bit = 15, NV12_Planar = 16, GRAY12 = 17,
};
enum ColorSpace { BT_601 = 0, BT_709 = 1, UNSPEC = 2 };
enum ColorRange { MPEG = 0, JPEG = 1, UDEF = 2 };

struct ColorspaceConversionContext {
  ColorSpace color_space;
  ColorRange color_range;
  ColorspaceConversionContext() : color_space(UNSPEC), color_range(UDEF) {}
  ColorspaceConversionContext(ColorSpace cs, ColorRange cr) : color_space(cs), color_range(cr) {}
};

typedef uintptr_t DevicePtr;    // the reference's CUdeviceptr
typedef uintptr_t HipContext;   // opaque: 0 = "current device", d + 1 = device ordinal d (HipResMgr::GetCtx)
typedef void* HipStream;        // hipStream_t

const char* PixelFormatName(Pixel_Format f);
int DeviceOfContext(HipContext ctx);  // -1 = current device

// RAII: make the context's device current for the scope (replaces CudaCtxPush)
class DeviceScope {
public:
  explicit DeviceScope(HipContext ctx);
  ~DeviceScope();
private:
  int prev_ = -1;
  bool switched_ = false;
};

// pluggable device allocator (process-wide).  nullptr restores the hipMalloc/hipFree default.
struct DeviceAllocator {
  void* (*alloc)(size_t bytes, int device, void* user);
  void (*free)(void* ptr, int device, void* user);
  void* user;
};
void SetDeviceAllocator(const DeviceAllocator* a);

// host memory; pinned (hipHostMalloc) when made with a context, like the reference's Buffer
class Buffer final : public Token {
public:
  ~Buffer() final;
  void* GetRawMemPtr() { return data_; }
  const void* GetRawMemPtr() const { return data_; }
  size_t GetRawMemSize() const { return size_; }
  void Update(size_t newSize, void* newPtr = nullptr);
  bool CopyFrom(size_t size, void const* ptr);
  template <typename T> T* GetDataAs() { return (T*)data_; }
  template <typename T> T const* GetDataAs() const { return (T const*)data_; }

  static Buffer* Make(size_t bufferSize);                     // owns pageable memory
  static Buffer* Make(size_t bufferSize, void* pCopyFrom);    // wraps, does not own
  static Buffer* MakeOwnMem(size_t bufferSize, HipContext ctx = 0);
  static Buffer* MakeOwnMem(size_t bufferSize, const void* pCopyFrom, HipContext ctx = 0);
  bool Pinned() const { return pinned_; }

private:
  Buffer(size_t size, void* wrap, bool own, bool pinned);
  void release();
  bool own_ = true, pinned_ = false;
  size_t size_ = 0;
  void* data_ = nullptr;
};

// linear device memory (kept under the reference's name: it is part of the Python API)
class CudaBuffer final : public Token {
public:
  static CudaBuffer* Make(size_t elemSize, size_t numElems, HipContext ctx);
  static CudaBuffer* Make(const void* hostPtr, size_t elemSize, size_t numElems, HipContext ctx, HipStream str);
  CudaBuffer* Clone();  // deep copy
  size_t GetRawMemSize() const { return elem_size_ * num_elems_; }
  size_t GetNumElems() const { return num_elems_; }
  size_t GetElemSize() const { return elem_size_; }
  DevicePtr GpuMem() { return mem_; }
  HipContext Context() const { return ctx_; }
  ~CudaBuffer();

private:
  CudaBuffer(size_t elemSize, size_t numElems, HipContext ctx);
  DevicePtr mem_ = 0;
  HipContext ctx_ = 0;
  int device_ = -1;
  size_t elem_size_ = 0, num_elems_ = 0;
};

// 2-D device memory without a format; sizes are raw (an RGB image is one plane 3x as wide)
struct SurfacePlane {
  DevicePtr gpuMem = 0;
  HipContext ctx = 0;
  uint32_t width = 0, height = 0, pitch = 0, elemSize = 0;
  bool ownMem = false;
  int allocDevice = -1;  // device ordinal the allocator was given in Allocate(): handed back unchanged to its free()

  SurfacePlane() = default;
  SurfacePlane(const SurfacePlane& other);             // non-owning alias
  SurfacePlane& operator=(const SurfacePlane& other);  // non-owning alias (frees what it owned)
  SurfacePlane(uint32_t w, uint32_t h, uint32_t pitch, uint32_t elemSize, DevicePtr ptr);  // wrap
  SurfacePlane(uint32_t w, uint32_t h, uint32_t elemSize, HipContext ctx);                 // allocate + own
  SurfacePlane(uint32_t w, uint32_t h, uint32_t elemSize, uint32_t srcPitch, DevicePtr src, HipContext ctx, HipStream str);
  ~SurfacePlane();

  void Allocate();
  void Deallocate();

  void Export(DevicePtr dst, uint32_t dst_pitch, HipContext ctx, HipStream str);
  void Export(DevicePtr dst, uint32_t dst_pitch, HipContext ctx, HipStream str, uint32_t roi_x, uint32_t roi_y,
              uint32_t roi_w, uint32_t roi_h, uint32_t pos_x, uint32_t pos_y);
  void Import(DevicePtr src, uint32_t src_pitch, HipContext ctx, HipStream str);
  void Import(DevicePtr src, uint32_t src_pitch, HipContext ctx, HipStream str, uint32_t roi_x, uint32_t roi_y,
              uint32_t roi_w, uint32_t roi_h, uint32_t pos_x, uint32_t pos_y);
  void Export(SurfacePlane& dst, HipContext ctx, HipStream str);
  void Import(SurfacePlane& src, HipContext ctx, HipStream str);

  bool OwnMemory() const { return ownMem; }
  DevicePtr GpuMem() const { return gpuMem; }
  uint32_t Width() const { return width; }
  uint32_t Height() const { return height; }
  uint32_t Pitch() const { return pitch; }
  uint32_t ElemSize() const { return elemSize; }
  uint32_t GetHostMemSize() const { return width * height * elemSize; }
  HipContext GetContext() const { return ctx; }
};

// A picture in device memory: 1-3 logical planes over 1-3 pitched allocations, geometry per format.
class Surface : public Token {
public:
  ~Surface() override;

  uint32_t Width(uint32_t plane = 0) const;         // pixels (chroma planes: their own width)
  uint32_t WidthInBytes(uint32_t plane = 0) const;  // bytes of one row of the plane
  uint32_t Height(uint32_t plane = 0) const;
  uint32_t Pitch(uint32_t plane = 0) const;
  uint32_t ElemSize() const;
  uint32_t HostMemSize() const;                      // tightly packed size of all planes
  uint32_t NumPlanes() const;
  DevicePtr PlanePtr(uint32_t plane = 0);
  Pixel_Format PixelFormat() const { return format_; }
  bool Empty() const { return alloc_[0].GpuMem() == 0; }
  // plane 0 (and, for three-allocation formats, planes 1-2) is the raw allocation, as in the reference;
  // other logical planes (NV12's UV, the G/B planes of RGB_PLANAR ...) are non-owning views of their region
  SurfacePlane* GetSurfacePlane(uint32_t plane = 0);
  bool Update(SurfacePlane* planes, size_t n);       // re-point at external memory (never when owning)
  Surface* Clone();                                  // non-owning alias (reference: "virtual copy ctor")
  Surface* Create();                                 // empty, same format
  HipContext Context() { return alloc_[0].GetContext(); }
  bool OwnMemory();

  // ROI copies; coordinates in pixels of plane 0, scaled per plane (chroma halves)
  void Import(Surface& src, HipContext ctx, HipStream str, uint32_t roi_x, uint32_t roi_y, uint32_t roi_w,
              uint32_t roi_h, uint32_t pos_x, uint32_t pos_y);
  void Export(Surface& dst, HipContext ctx, HipStream str, uint32_t roi_x, uint32_t roi_y, uint32_t roi_w,
              uint32_t roi_h, uint32_t pos_x, uint32_t pos_y);

  static Surface* Make(Pixel_Format format);                                                  // empty
  static Surface* Make(Pixel_Format format, uint32_t width, uint32_t height, HipContext ctx);  // allocates
  // wrap one existing pitched allocation (single-allocation formats: Y, NV12, P10/P12, RGB, BGR, planar)
  static Surface* Make(Pixel_Format format, uint32_t width, uint32_t height, uint32_t pitch, DevicePtr ptr);
  static bool Supported(Pixel_Format format);
  static uint32_t HostMemSizeOf(Pixel_Format format, uint32_t width, uint32_t height);  // tight size of a w x h picture

private:
  Surface(Pixel_Format f, uint32_t w, uint32_t h);
  void refresh_views();
  Pixel_Format format_;
  uint32_t w_ = 0, h_ = 0;  // picture size in pixels
  SurfacePlane alloc_[3];
  SurfacePlane view_[3];
};

// process-wide per-device resources (replaces CudaResMgr, src/PyNvCodec/src/PyNvCodec.cpp:57-162)
class HipResMgr {
public:
  static HipResMgr& Instance();
  size_t GetNumGpus();
  HipContext GetCtx(size_t gpu_id);  // cookie for device gpu_id; throws on a bad ordinal
  HipStream GetStream(size_t gpu_id);  // one non-blocking stream per device, created lazily
private:
  HipResMgr() = default;
  ~HipResMgr();
  struct Impl;
  Impl* impl();
  Impl* impl_ = nullptr;
};

}  // namespace VPF
