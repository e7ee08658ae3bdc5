// MemoryInterfaces.cpp — see MemoryInterfaces.hpp.  Geometry follows the reference's per-format classes
// (src/TC/src/MemoryInterfaces.cpp: SurfaceY :760-810, SurfaceNV12 :811-913, SurfaceYUV420 :915-1062,
// SurfaceRGB/BGR :1361-1519, SurfaceRGBPlanar/YUV444 :1521-1637, RGB32F :1639-1860) through one table.
#include "MemoryInterfaces.hpp"

#include <atomic>

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <vector>

namespace VPF {

namespace {
void ThrowOnHipError(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    std::stringstream ss;
    ss << what << ": HIP error " << hipGetErrorName(e) << " (" << hipGetErrorString(e) << ")";
    throw std::runtime_error(ss.str());
  }
}
inline uint32_t cdiv2(uint32_t v) { return (v + 1) / 2; }
inline uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

void* default_alloc(size_t bytes, int, void*) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  return p;
}
void default_free(void* p, int, void*) { (void)hipFree(p); }
DeviceAllocator g_alloc = {default_alloc, default_free, nullptr};
std::mutex g_alloc_mutex;

enum Kind { K_NONE, K_Y, K_SEMI, K_420, K_422, K_PACKED3, K_PLANAR3 };
struct Traits {
  Kind kind;
  uint32_t elem;
  const char* name;
};
Traits traits(Pixel_Format f) {
  switch (f) {
    case Y: return {K_Y, 1, "Y"};
    case NV12: return {K_SEMI, 1, "NV12"};
    case P10: return {K_SEMI, 2, "P10"};
    case P12: return {K_SEMI, 2, "P12"};
    case YUV420: return {K_420, 1, "YUV420"};
    case YCBCR: return {K_420, 1, "YCBCR"};
    case YUV420_10bit: return {K_SEMI, 2, "YUV420_10bit"};  // the reference factory builds a SurfaceP12 for it (MemoryInterfaces.cpp:662-664)
    case YUV422: return {K_422, 1, "YUV422"};
    case RGB: return {K_PACKED3, 1, "RGB"};
    case BGR: return {K_PACKED3, 1, "BGR"};
    case RGB_32F: return {K_PACKED3, 4, "RGB_32F"};
    case RGB_PLANAR: return {K_PLANAR3, 1, "RGB_PLANAR"};
    case YUV444: return {K_PLANAR3, 1, "YUV444"};
    case YUV444_10bit: return {K_PLANAR3, 2, "YUV444_10bit"};
    case RGB_32F_PLANAR: return {K_PLANAR3, 4, "RGB_32F_PLANAR"};
    default: return {K_NONE, 0, "UNDEFINED"};
  }
}
uint32_t num_allocs(Kind k) { return (k == K_420 || k == K_422) ? 3 : (k == K_NONE ? 0 : 1); }
uint32_t num_planes(Kind k) {
  switch (k) {
    case K_Y: case K_PACKED3: return 1;
    case K_SEMI: return 2;
    case K_420: case K_422: case K_PLANAR3: return 3;
    default: return 0;
  }
}
// allocation a of a w x h picture, in elements / rows
void alloc_dims(Kind k, uint32_t a, uint32_t w, uint32_t h, uint32_t* aw, uint32_t* ah) {
  switch (k) {
    case K_Y: *aw = w; *ah = h; break;
    case K_SEMI: *aw = 2 * cdiv2(w); *ah = h + cdiv2(h); break;  // luma rows then interleaved-chroma rows, one pitch
    case K_420: *aw = a ? cdiv2(w) : w; *ah = a ? cdiv2(h) : h; break;
    case K_422: *aw = a ? cdiv2(w) : w; *ah = h; break;
    case K_PACKED3: *aw = 3 * w; *ah = h; break;
    case K_PLANAR3: *aw = w; *ah = 3 * h; break;
    default: *aw = *ah = 0;
  }
}
struct PlaneGeo {
  uint32_t alloc, row_off, width_px, width_elems, height;
};
PlaneGeo plane_geo(Kind k, uint32_t p, uint32_t w, uint32_t h) {
  switch (k) {
    case K_Y: return {0, 0, w, w, h};
    case K_SEMI: return p == 0 ? PlaneGeo{0, 0, w, w, h} : PlaneGeo{0, h, w, 2 * cdiv2(w), cdiv2(h)};
    case K_420: return p == 0 ? PlaneGeo{0, 0, w, w, h} : PlaneGeo{p, 0, cdiv2(w), cdiv2(w), cdiv2(h)};
    case K_422: return p == 0 ? PlaneGeo{0, 0, w, w, h} : PlaneGeo{p, 0, cdiv2(w), cdiv2(w), h};
    case K_PACKED3: return {0, 0, w, 3 * w, h};
    case K_PLANAR3: return {0, p * h, w, w, h};
    default: return {0, 0, 0, 0, 0};
  }
}
void copy2d(DevicePtr dst, uint32_t dpitch, DevicePtr src, uint32_t spitch, size_t wbytes, size_t rows, HipContext ctx,
            HipStream str, const char* what) {
  if (!dst || !src || !wbytes || !rows) return;
  DeviceScope scope(ctx);
  ThrowOnHipError(hipMemcpy2DAsync((void*)dst, dpitch, (const void*)src, spitch, wbytes, rows, hipMemcpyDeviceToDevice,
                                   (hipStream_t)str), what);
  ThrowOnHipError(hipStreamSynchronize((hipStream_t)str), what);
}
}  // namespace

const char* PixelFormatName(Pixel_Format f) { return traits(f).name; }

int DeviceOfContext(HipContext ctx) {
  int n = 0;
  if (ctx == 0 || hipGetDeviceCount(&n) != hipSuccess) return -1;
  return (ctx <= (HipContext)n) ? (int)ctx - 1 : -1;  // foreign handles (e.g. a stale CUcontext value) mean "current"
}

DeviceScope::DeviceScope(HipContext ctx) {
  const int dev = DeviceOfContext(ctx);
  if (dev < 0) return;
  if (hipGetDevice(&prev_) == hipSuccess && prev_ != dev && hipSetDevice(dev) == hipSuccess) switched_ = true;
}
DeviceScope::~DeviceScope() {
  if (switched_) (void)hipSetDevice(prev_);
}

void SetDeviceAllocator(const DeviceAllocator* a) {
  std::lock_guard<std::mutex> lock(g_alloc_mutex);
  g_alloc = a ? *a : DeviceAllocator{default_alloc, default_free, nullptr};
}

// ---------------------------------------------------------------------------------------------- Buffer
// A staging buffer that was asked to be page-locked but is not (hipHostMalloc failed: no device / pinned-memory limit) still
// works, but every upload / download through it loses its asynchronous, overlapped DMA.  Said once per process on stderr.
static void note_unpinned(size_t size) {
  static std::atomic<bool> said{false};
  (void)hipGetLastError();
  if (!said.exchange(true))
    std::cerr << "libvpf (MemoryInterfaces): hipHostMalloc(" << size << " B) failed; staging falls back to pageable host memory — uploads / "
                 "downloads through it are no longer asynchronous or overlapped (reported once)" << std::endl;
}
Buffer::Buffer(size_t size, void* wrap, bool own, bool pinned) : own_(own), pinned_(false), size_(size) {
  if (!own) {
    data_ = wrap;
    return;
  }
  if (pinned && hipHostMalloc(&data_, size ? size : 1, hipHostMallocDefault) == hipSuccess) {
    pinned_ = true;
  } else {
    if (pinned) note_unpinned(size);
    data_ = std::calloc(size ? size : 1, 1);
    if (!data_) throw std::bad_alloc();
  }
}
void Buffer::release() {
  if (own_ && data_) {
    if (pinned_) (void)hipHostFree(data_);
    else std::free(data_);
  }
  data_ = nullptr;
}
Buffer::~Buffer() { release(); }
Buffer* Buffer::Make(size_t n) { return new Buffer(n, nullptr, true, false); }
Buffer* Buffer::Make(size_t n, void* p) { return new Buffer(n, p, false, false); }
Buffer* Buffer::MakeOwnMem(size_t n, HipContext ctx) { return new Buffer(n, nullptr, true, ctx != 0); }
Buffer* Buffer::MakeOwnMem(size_t n, const void* src, HipContext ctx) {
  Buffer* b = new Buffer(n, nullptr, true, ctx != 0);
  if (src) std::memcpy(b->data_, src, n);
  return b;
}
void Buffer::Update(size_t newSize, void* newPtr) {
  // reference semantics (MemoryInterfaces.cpp:240-262): a wrapper re-points, an owner re-allocates
  if (own_) {
    const bool pin = pinned_;
    release();
    size_ = newSize;
    if (pin && hipHostMalloc(&data_, newSize ? newSize : 1, hipHostMallocDefault) == hipSuccess) pinned_ = true;
    else { if (pin) note_unpinned(newSize); pinned_ = false; data_ = std::calloc(newSize ? newSize : 1, 1); }
    if (newPtr) std::memcpy(data_, newPtr, newSize);
  } else {
    size_ = newSize;
    data_ = newPtr;
  }
}
bool Buffer::CopyFrom(size_t size, void const* ptr) {
  if (!ptr || size > size_ || !data_) return false;
  std::memcpy(data_, ptr, size);
  return true;
}

// ---------------------------------------------------------------------------------------------- CudaBuffer
CudaBuffer::CudaBuffer(size_t e, size_t n, HipContext ctx) : ctx_(ctx), elem_size_(e), num_elems_(n) {
  DeviceScope scope(ctx);
  (void)hipGetDevice(&device_);
  DeviceAllocator a;
  { std::lock_guard<std::mutex> lock(g_alloc_mutex); a = g_alloc; }
  mem_ = (DevicePtr)a.alloc(GetRawMemSize() ? GetRawMemSize() : 1, device_, a.user);
  if (!mem_) throw std::runtime_error("CudaBuffer: device allocation failed");
}
CudaBuffer::~CudaBuffer() {
  if (mem_) {
    DeviceAllocator a;
    { std::lock_guard<std::mutex> lock(g_alloc_mutex); a = g_alloc; }
    a.free((void*)mem_, device_, a.user);
  }
}
CudaBuffer* CudaBuffer::Make(size_t e, size_t n, HipContext ctx) { return new CudaBuffer(e, n, ctx); }
CudaBuffer* CudaBuffer::Make(const void* host, size_t e, size_t n, HipContext ctx, HipStream str) {
  // synchronous H2D like the reference (MemoryInterfaces.cpp:336-353)
  std::unique_ptr<CudaBuffer> b(new CudaBuffer(e, n, ctx));
  DeviceScope scope(ctx);
  ThrowOnHipError(hipMemcpyAsync((void*)b->mem_, host, b->GetRawMemSize(), hipMemcpyHostToDevice, (hipStream_t)str), "CudaBuffer::Make");
  ThrowOnHipError(hipStreamSynchronize((hipStream_t)str), "CudaBuffer::Make");
  return b.release();
}
CudaBuffer* CudaBuffer::Clone() {
  std::unique_ptr<CudaBuffer> b(new CudaBuffer(elem_size_, num_elems_, ctx_));
  DeviceScope scope(ctx_);
  // D2D hipMemcpy may return before the copy has finished; the clone must be complete for users on other
  // (non-blocking) streams, so drain the null stream explicitly
  ThrowOnHipError(hipMemcpyAsync((void*)b->mem_, (const void*)mem_, GetRawMemSize(), hipMemcpyDeviceToDevice, nullptr), "CudaBuffer::Clone");
  ThrowOnHipError(hipStreamSynchronize(nullptr), "CudaBuffer::Clone");
  return b.release();
}

// ---------------------------------------------------------------------------------------------- SurfacePlane
SurfacePlane::SurfacePlane(const SurfacePlane& o)
    : gpuMem(o.gpuMem), ctx(o.ctx), width(o.width), height(o.height), pitch(o.pitch), elemSize(o.elemSize), ownMem(false) {}
SurfacePlane& SurfacePlane::operator=(const SurfacePlane& o) {
  if (this == &o) return *this;
  Deallocate();
  gpuMem = o.gpuMem; ctx = o.ctx; width = o.width; height = o.height; pitch = o.pitch; elemSize = o.elemSize;
  ownMem = false;
  return *this;
}
SurfacePlane::SurfacePlane(uint32_t w, uint32_t h, uint32_t p, uint32_t e, DevicePtr ptr)
    : gpuMem(ptr), width(w), height(h), pitch(p), elemSize(e), ownMem(false) {}
SurfacePlane::SurfacePlane(uint32_t w, uint32_t h, uint32_t e, HipContext c) : ctx(c), width(w), height(h), elemSize(e), ownMem(true) {
  Allocate();
}
SurfacePlane::SurfacePlane(uint32_t w, uint32_t h, uint32_t e, uint32_t srcPitch, DevicePtr src, HipContext c, HipStream str)
    : SurfacePlane(w, h, e, c) {
  Import(src, srcPitch, c, str);
}
SurfacePlane::~SurfacePlane() { Deallocate(); }

void SurfacePlane::Allocate() {
  if (!ownMem || gpuMem) return;
  pitch = round_up(width * elemSize, 256);  // every row start 256-B aligned: dwordx4 everywhere, whole cache lines
  const size_t bytes = (size_t)pitch * height;
  DeviceScope scope(ctx);
  int dev = 0;
  (void)hipGetDevice(&dev);
  DeviceAllocator a;
  { std::lock_guard<std::mutex> lock(g_alloc_mutex); a = g_alloc; }
  gpuMem = (DevicePtr)a.alloc(bytes ? bytes : 1, dev, a.user);
  if (!gpuMem) throw std::runtime_error("SurfacePlane: device allocation failed");
  allocDevice = dev;
}
void SurfacePlane::Deallocate() {
  if (ownMem && gpuMem) {
    DeviceAllocator a;
    { std::lock_guard<std::mutex> lock(g_alloc_mutex); a = g_alloc; }
    DeviceScope scope(ctx);  // a per-device allocator sees the same (device, ordinal) pair on free as on alloc
    a.free((void*)gpuMem, allocDevice, a.user);
  }
  gpuMem = 0;
  ownMem = false;
}
void SurfacePlane::Import(DevicePtr src, uint32_t sp, HipContext c, HipStream str, uint32_t rx, uint32_t ry, uint32_t rw,
                          uint32_t rh, uint32_t px, uint32_t py) {
  if (rx + rw > Width() || ry + rh > Height()) throw std::runtime_error("ROI isn't enclosed within a Surface plane");
  if (!src || !gpuMem) return;
  copy2d(gpuMem + (size_t)px * elemSize + (size_t)py * pitch, pitch, src + (size_t)rx * elemSize + (size_t)ry * sp, sp,
         (size_t)rw * elemSize, rh, c, str, "SurfacePlane::Import");
}
void SurfacePlane::Import(DevicePtr src, uint32_t sp, HipContext c, HipStream str) { Import(src, sp, c, str, 0, 0, Width(), Height(), 0, 0); }
void SurfacePlane::Export(DevicePtr dst, uint32_t dp, HipContext c, HipStream str, uint32_t rx, uint32_t ry, uint32_t rw,
                          uint32_t rh, uint32_t px, uint32_t py) {
  if (rx + rw > Width() || ry + rh > Height()) throw std::runtime_error("ROI isn't enclosed within a Surface plane");
  if (!dst || !gpuMem) return;
  copy2d(dst + (size_t)px * elemSize + (size_t)py * dp, dp, gpuMem + (size_t)rx * elemSize + (size_t)ry * pitch, pitch,
         (size_t)rw * elemSize, rh, c, str, "SurfacePlane::Export");
}
void SurfacePlane::Export(DevicePtr dst, uint32_t dp, HipContext c, HipStream str) { Export(dst, dp, c, str, 0, 0, Width(), Height(), 0, 0); }
void SurfacePlane::Export(SurfacePlane& dst, HipContext c, HipStream str) {
  if (Width() != dst.Width() || Height() != dst.Height() || ElemSize() != dst.ElemSize()) return;
  Export(dst.GpuMem(), dst.Pitch(), c, str);
}
void SurfacePlane::Import(SurfacePlane& src, HipContext c, HipStream str) {
  if (Width() != src.Width() || Height() != src.Height() || ElemSize() != src.ElemSize()) return;
  Import(src.GpuMem(), src.Pitch(), c, str);
}

// ---------------------------------------------------------------------------------------------- Surface
Surface::Surface(Pixel_Format f, uint32_t w, uint32_t h) : format_(f), w_(w), h_(h) {}
Surface::~Surface() = default;

bool Surface::Supported(Pixel_Format f) { return traits(f).kind != K_NONE; }
uint32_t Surface::HostMemSizeOf(Pixel_Format f, uint32_t w, uint32_t h) {
  const Traits t = traits(f);
  uint32_t n = 0;
  for (uint32_t p = 0; p < num_planes(t.kind); p++) {
    const PlaneGeo g = plane_geo(t.kind, p, w, h);
    n += g.width_elems * t.elem * g.height;
  }
  return n;
}

Surface* Surface::Make(Pixel_Format f) {
  if (!Supported(f)) {
    std::cerr << "Surface::Make: unsupported pixel format: " << (int)f << std::endl;
    return nullptr;
  }
  return new Surface(f, 0, 0);
}
Surface* Surface::Make(Pixel_Format f, uint32_t w, uint32_t h, HipContext ctx) {
  const Traits t = traits(f);
  if (t.kind == K_NONE) {
    std::cerr << "Surface::Make: unsupported pixel format: " << (int)f << std::endl;
    return nullptr;
  }
  std::unique_ptr<Surface> s(new Surface(f, w, h));
  for (uint32_t a = 0; a < num_allocs(t.kind); a++) {
    uint32_t aw, ah;
    alloc_dims(t.kind, a, w, h, &aw, &ah);
    SurfacePlane owned(aw, ah, t.elem, ctx);
    // move ownership into the member (operator= makes aliases, so hand the pointer over explicitly)
    s->alloc_[a].gpuMem = owned.gpuMem; s->alloc_[a].ctx = ctx; s->alloc_[a].width = aw; s->alloc_[a].height = ah;
    s->alloc_[a].pitch = owned.pitch; s->alloc_[a].elemSize = t.elem; s->alloc_[a].ownMem = true;
    s->alloc_[a].allocDevice = owned.allocDevice;
    owned.ownMem = false; owned.gpuMem = 0;
  }
  s->refresh_views();
  return s.release();
}
Surface* Surface::Make(Pixel_Format f, uint32_t w, uint32_t h, uint32_t pitch, DevicePtr ptr) {
  const Traits t = traits(f);
  if (t.kind == K_NONE || num_allocs(t.kind) != 1) return nullptr;
  uint32_t aw, ah;
  alloc_dims(t.kind, 0, w, h, &aw, &ah);
  Surface* s = new Surface(f, w, h);
  s->alloc_[0] = SurfacePlane(aw, ah, pitch, t.elem, ptr);
  s->refresh_views();
  return s;
}

void Surface::refresh_views() {
  const Traits t = traits(format_);
  for (uint32_t p = 0; p < 3; p++) view_[p] = SurfacePlane();
  if (Empty()) return;
  for (uint32_t p = 0; p < num_planes(t.kind); p++) {
    const PlaneGeo g = plane_geo(t.kind, p, w_, h_);
    const SurfacePlane& a = alloc_[g.alloc];
    view_[p] = SurfacePlane(g.width_elems, g.height, a.pitch, t.elem, a.gpuMem + (size_t)g.row_off * a.pitch);
    view_[p].ctx = a.ctx;
  }
}

uint32_t Surface::NumPlanes() const { return num_planes(traits(format_).kind); }
uint32_t Surface::ElemSize() const { return traits(format_).elem; }
static void check_plane(uint32_t p, uint32_t n) {
  if (p >= n) throw std::invalid_argument("Invalid plane number");
}
uint32_t Surface::Width(uint32_t p) const { check_plane(p, NumPlanes()); return plane_geo(traits(format_).kind, p, w_, h_).width_px; }
uint32_t Surface::WidthInBytes(uint32_t p) const {
  check_plane(p, NumPlanes());
  return plane_geo(traits(format_).kind, p, w_, h_).width_elems * ElemSize();
}
uint32_t Surface::Height(uint32_t p) const { check_plane(p, NumPlanes()); return plane_geo(traits(format_).kind, p, w_, h_).height; }
uint32_t Surface::Pitch(uint32_t p) const {
  check_plane(p, NumPlanes());
  return alloc_[plane_geo(traits(format_).kind, p, w_, h_).alloc].pitch;
}
uint32_t Surface::HostMemSize() const {
  uint32_t n = 0;
  for (uint32_t p = 0; p < NumPlanes(); p++) n += WidthInBytes(p) * Height(p);
  return n;
}
DevicePtr Surface::PlanePtr(uint32_t p) {
  check_plane(p, NumPlanes());
  if (Empty()) return 0;
  const PlaneGeo g = plane_geo(traits(format_).kind, p, w_, h_);
  return alloc_[g.alloc].gpuMem + (size_t)g.row_off * alloc_[g.alloc].pitch;
}
SurfacePlane* Surface::GetSurfacePlane(uint32_t p) {
  if (p >= NumPlanes()) return nullptr;
  const Kind k = traits(format_).kind;
  if (num_allocs(k) == 3) return &alloc_[p];
  return p == 0 ? &alloc_[0] : &view_[p];
}
bool Surface::Update(SurfacePlane* planes, size_t n) {
  const Kind k = traits(format_).kind;
  if (!planes || n != num_allocs(k)) return false;
  for (uint32_t a = 0; a < n; a++)
    if (alloc_[a].OwnMemory()) return false;
  for (uint32_t a = 0; a < n; a++) alloc_[a] = planes[a];
  // picture size from the new allocation(s)
  switch (k) {
    case K_SEMI: w_ = alloc_[0].width; h_ = alloc_[0].height * 2 / 3; break;
    case K_PACKED3: w_ = alloc_[0].width / 3; h_ = alloc_[0].height; break;
    case K_PLANAR3: w_ = alloc_[0].width; h_ = alloc_[0].height / 3; break;
    default: w_ = alloc_[0].width; h_ = alloc_[0].height;
  }
  refresh_views();
  return true;
}
Surface* Surface::Clone() {
  Surface* s = new Surface(format_, w_, h_);
  for (int a = 0; a < 3; a++) s->alloc_[a] = alloc_[a];  // aliases
  s->refresh_views();
  return s;
}
Surface* Surface::Create() { return new Surface(format_, 0, 0); }
bool Surface::OwnMemory() {
  const uint32_t n = num_allocs(traits(format_).kind);
  for (uint32_t a = 0; a < n; a++)
    if (!alloc_[a].OwnMemory()) return false;
  return n > 0;
}

// ROI copy between same-format surfaces.  Coordinates are pixels of plane 0; each logical plane scales them by
// its own subsampling, and the byte offset uses the plane's bytes per pixel (so packed RGB crops by pixels —
// the reference scales RGB ROIs by 1 and therefore crops bytes, src/TC/src/MemoryInterfaces.cpp:683-700).
void Surface::Export(Surface& dst, HipContext ctx, HipStream str, uint32_t rx, uint32_t ry, uint32_t rw, uint32_t rh,
                     uint32_t px, uint32_t py) {
  if (PixelFormat() != dst.PixelFormat()) throw std::runtime_error("Pixel format mismatch.");
  if (rx + rw > Width() || ry + rh > Height() || px + rw > dst.Width() || py + rh > dst.Height())
    throw std::runtime_error("ROI isn't enclosed within a Surface plane");
  for (uint32_t p = 0; p < NumPlanes(); p++) {
    // subsampling of this plane relative to plane 0 (NV12's interleaved UV plane reports the luma width, so it
    // is addressed like luma horizontally: byte x of a UV row belongs to pixel x for even x)
    const uint32_t sx = Width() > Width(p) ? 2 : 1;
    const uint32_t sy = Height() > Height(p) ? 2 : 1;
    const uint32_t bpp = WidthInBytes(p) / Width(p);
    const uint32_t x0 = rx / sx, y0 = ry / sy, x1 = (rx + rw + sx - 1) / sx, y1 = (ry + rh + sy - 1) / sy;
    const uint32_t dx = px / sx, dy = py / sy;
    copy2d(dst.PlanePtr(p) + (size_t)dx * bpp + (size_t)dy * dst.Pitch(p), dst.Pitch(p),
           PlanePtr(p) + (size_t)x0 * bpp + (size_t)y0 * Pitch(p), Pitch(p), (size_t)(x1 - x0) * bpp, y1 - y0, ctx, str,
           "Surface::Export");
  }
}
void Surface::Import(Surface& src, HipContext ctx, HipStream str, uint32_t rx, uint32_t ry, uint32_t rw, uint32_t rh,
                     uint32_t px, uint32_t py) {
  src.Export(*this, ctx, str, rx, ry, rw, rh, px, py);
}

// ---------------------------------------------------------------------------------------------- HipResMgr
struct HipResMgr::Impl {
  std::mutex m;
  std::vector<hipStream_t> streams;
};
HipResMgr& HipResMgr::Instance() {
  static HipResMgr inst;
  return inst;
}
HipResMgr::Impl* HipResMgr::impl() {
  static std::mutex m;
  std::lock_guard<std::mutex> lock(m);
  if (!impl_) {
    impl_ = new Impl;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    impl_->streams.assign((size_t)n, nullptr);
  }
  return impl_;
}
HipResMgr::~HipResMgr() {
  // streams are left to process teardown: destroying them after the HIP runtime has shut down is an error
  delete impl_;
}
size_t HipResMgr::GetNumGpus() { return impl()->streams.size(); }
HipContext HipResMgr::GetCtx(size_t gpu_id) {
  if (gpu_id >= GetNumGpus()) throw std::runtime_error("HipResMgr: GPU ordinal out of range (no such device)");
  return (HipContext)gpu_id + 1;
}
HipStream HipResMgr::GetStream(size_t gpu_id) {
  Impl* im = impl();
  if (gpu_id >= im->streams.size()) throw std::runtime_error("HipResMgr: GPU ordinal out of range (no such device)");
  std::lock_guard<std::mutex> lock(im->m);
  if (!im->streams[gpu_id]) {
    DeviceScope scope((HipContext)gpu_id + 1);
    ThrowOnHipError(hipStreamCreateWithFlags(&im->streams[gpu_id], hipStreamNonBlocking), "HipResMgr::GetStream");
  }
  return (HipStream)im->streams[gpu_id];
}

}  // namespace VPF
