// PyNvCodec.cpp — pybind11 module `_PyNvCodec`: the Python API of the surface path, name for name with
// the reference's bindings (src/PyNvCodec/src/PyNvCodec.cpp:208-461 module + enums, PySurface.cpp:163-492,
// PySurfaceConverter.cpp:80-120, PySurfaceResizer.cpp:66-102, PySurfaceRemaper.cpp:70-107,
// PyFrameUploader.cpp:104-161, PySurfaceDownloader.cpp:119-189, PyBufferUploader.cpp:48-96,
// PyCudaBufferDownloader.cpp:50-104) over the HIP Task layer in ../tc.
//
// `context` arguments are opaque device cookies on ROCm (0 = current device, see MemoryInterfaces.hpp);
// `stream` arguments are hipStream_t values (torch.cuda.Stream().cuda_stream works).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "Tasks.hpp"
#include "vpf_hip.h"
#ifdef VPF_WITH_LIBAV
#include <map>

#include "FfmpegFeeder.hpp"
#endif

namespace py = pybind11;
using namespace VPF;

namespace {
constexpr auto TASK_EXEC_SUCCESS = TaskExecStatus::TASK_EXEC_SUCCESS;
constexpr auto TASK_EXEC_FAIL = TaskExecStatus::TASK_EXEC_FAIL;

HipContext ctx_of(int gpu) { return HipResMgr::Instance().GetCtx((size_t)gpu); }
HipStream str_of(int gpu) { return HipResMgr::Instance().GetStream((size_t)gpu); }

std::string plane_repr(SurfacePlane* p, int indent = 0) {
  if (!p) return {};
  const std::string sp(indent, ' ');
  std::stringstream ss;
  ss << sp << "Owns mem:  " << p->OwnMemory() << "\n" << sp << "Width:     " << p->Width() << "\n"
     << sp << "Height:    " << p->Height() << "\n" << sp << "Pitch:     " << p->Pitch() << "\n"
     << sp << "Elem size: " << p->ElemSize() << "\n" << sp << "HIP ctx:   " << p->GetContext() << "\n"
     << sp << "HIP ptr:   " << p->GpuMem() << "\n";
  return ss.str();
}
std::string surface_repr(Surface* s) {
  if (!s) return {};
  std::stringstream ss;
  if (s->Empty()) {
    ss << "Empty surface\nFormat:           " << PixelFormatName(s->PixelFormat()) << "\n";
    return ss.str();
  }
  ss << "Width:            " << s->Width() << "\nHeight:           " << s->Height()
     << "\nFormat:           " << PixelFormatName(s->PixelFormat()) << "\nPitch:            " << s->Pitch()
     << "\nElem size(bytes): " << s->ElemSize() << "\n";
  for (uint32_t i = 0; i < s->NumPlanes() && s->GetSurfacePlane(i); i++) ss << "Plane " << i << "\n" << plane_repr(s->GetSurfacePlane(i), 2) << "\n";
  return ss.str();
}

// deep copy src -> dst, plane by plane, then wait (PySurface.cpp:54-81)
void copy_surface(Surface& src, Surface& dst, HipContext ctx, HipStream str) {
  if (src.Empty() || dst.Empty()) return;
  src.Export(dst, ctx, str, 0, 0, src.Width(), src.Height(), 0, 0);
}
void check_same(Surface& a, Surface& b) {
  if (a.PixelFormat() != b.PixelFormat()) throw std::runtime_error("Surfaces have different pixel formats");
  if (a.Width() != b.Width() || a.Height() != b.Height()) throw std::runtime_error("Surfaces have different size");
}
std::shared_ptr<Surface> make_surface(Pixel_Format f, uint32_t w, uint32_t h, HipContext ctx) {
  Surface* s = Surface::Make(f, w, h, ctx);
  if (!s) throw std::invalid_argument("Surface.Make: unsupported pixel format");
  return std::shared_ptr<Surface>(s);
}
std::shared_ptr<Surface> empty_surface(Pixel_Format f) { return std::shared_ptr<Surface>(Surface::Make(f)); }

// host-memory "device" allocator: lets CPU-only unit tests build Surfaces (geometry, dispatch, error paths)
void* host_alloc(size_t n, int, void*) { return std::calloc(n ? n : 1, 1); }
void host_free(void* p, int, void*) { std::free(p); }

// ------------------------------------------------------------------------------------------------
class PySurfaceConverter {
  std::unique_ptr<ConvertSurface> conv_;
  std::unique_ptr<Buffer> ctx_buf_;
  Pixel_Format out_fmt_;

public:
  PySurfaceConverter(uint32_t w, uint32_t h, Pixel_Format in, Pixel_Format out, HipContext ctx, HipStream str) : out_fmt_(out) {
    conv_.reset(ConvertSurface::Make(w, h, in, out, ctx, str));
    ctx_buf_.reset(Buffer::MakeOwnMem(sizeof(ColorspaceConversionContext)));
  }
  Pixel_Format GetFormat() const { return out_fmt_; }
  void SetOutputReuseHint(bool on) { conv_->SetOutputReused(on); }
  bool GetOutputReuseHint() const { return conv_->GetOutputReused(); }
  // PySurfaceConverter.cpp:50-74: returns a NON-OWNING alias of the task's single output surface (overwritten by
  // the next Execute); failure of any kind = an Empty() surface
  std::shared_ptr<Surface> Execute(std::shared_ptr<Surface> src, std::shared_ptr<ColorspaceConversionContext> cc) {
    if (!src) return empty_surface(out_fmt_);
    conv_->ClearInputs();
    conv_->SetInput(src.get(), 0U);
    if (cc) {
      ctx_buf_->CopyFrom(sizeof(ColorspaceConversionContext), cc.get());
      conv_->SetInput(ctx_buf_.get(), 1U);
    }
    if (TASK_EXEC_SUCCESS != conv_->Execute()) return empty_surface(out_fmt_);
    auto* out = static_cast<Surface*>(conv_->GetOutput(0U));
    return std::shared_ptr<Surface>(out ? out->Clone() : Surface::Make(out_fmt_));
  }
  // additive: n surfaces in, n caller-owned surfaces out, one dispatch per 32 frames
  bool ExecuteBatch(const std::vector<std::shared_ptr<Surface>>& src, const std::vector<std::shared_ptr<Surface>>& dst,
                    std::shared_ptr<ColorspaceConversionContext> cc) {
    if (src.size() != dst.size() || src.empty()) return false;
    std::vector<Surface*> a, b;
    for (auto& s : src) a.push_back(s.get());
    for (auto& d : dst) b.push_back(d.get());
    return TASK_EXEC_SUCCESS == conv_->RunBatch(a.data(), b.data(), (uint32_t)a.size(), cc.get());
  }
};

// additive: fused conversion + bilinear resize (ConvertResizeSurface), same calling conventions as PySurfaceConverter
class PySurfaceConvertResizer {
  std::unique_ptr<ConvertResizeSurface> task_;
  std::unique_ptr<Buffer> ctx_buf_;
  Pixel_Format out_fmt_;

public:
  PySurfaceConvertResizer(uint32_t sw, uint32_t sh, Pixel_Format in, uint32_t dw, uint32_t dh, Pixel_Format out, HipContext ctx, HipStream str)
      : out_fmt_(out) {
    task_.reset(ConvertResizeSurface::Make(sw, sh, in, dw, dh, out, ctx, str));
    ctx_buf_.reset(Buffer::MakeOwnMem(sizeof(ColorspaceConversionContext)));
  }
  Pixel_Format GetFormat() const { return out_fmt_; }
  std::shared_ptr<Surface> Execute(std::shared_ptr<Surface> src, std::shared_ptr<ColorspaceConversionContext> cc) {
    if (!src) return empty_surface(out_fmt_);
    task_->ClearInputs();
    task_->SetInput(src.get(), 0U);
    if (cc) {
      ctx_buf_->CopyFrom(sizeof(ColorspaceConversionContext), cc.get());
      task_->SetInput(ctx_buf_.get(), 1U);
    }
    if (TASK_EXEC_SUCCESS != task_->Execute()) return empty_surface(out_fmt_);
    auto* out = static_cast<Surface*>(task_->GetOutput(0U));
    return std::shared_ptr<Surface>(out ? out->Clone() : Surface::Make(out_fmt_));
  }
  bool ExecuteBatch(const std::vector<std::shared_ptr<Surface>>& src, const std::vector<std::shared_ptr<Surface>>& dst,
                    std::shared_ptr<ColorspaceConversionContext> cc) {
    if (src.size() != dst.size() || src.empty()) return false;
    std::vector<Surface*> a, b;
    for (auto& s : src) a.push_back(s.get());
    for (auto& d : dst) b.push_back(d.get());
    return TASK_EXEC_SUCCESS == task_->RunBatch(a.data(), b.data(), (uint32_t)a.size(), cc.get());
  }
};

class PySurfaceResizer {
  std::unique_ptr<ResizeSurface> rs_;
  Pixel_Format fmt_;

public:
  PySurfaceResizer(uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str) : fmt_(f) {
    rs_.reset(ResizeSurface::Make(w, h, f, ctx, str));
  }
  Pixel_Format GetFormat() const { return fmt_; }
  void SetInterpolation(int i) { rs_->SetInterpolation(i); }
  void SetAsync(bool on) { rs_->SetAsync(on); }
  bool GetAsync() const { return rs_->GetAsync(); }
  int GetInterpolation() const { return rs_->GetInterpolation(); }
  std::shared_ptr<Surface> Execute(std::shared_ptr<Surface> src) {
    if (!src) return empty_surface(fmt_);
    rs_->SetInput(src.get(), 0U);
    if (TASK_EXEC_SUCCESS != rs_->Execute()) return empty_surface(fmt_);
    auto* out = static_cast<Surface*>(rs_->GetOutput(0U));
    return std::shared_ptr<Surface>(out ? out->Clone() : Surface::Make(fmt_));
  }
  // additive: n surfaces in, n caller-owned surfaces out, every plane of every frame in as few dispatches as possible
  bool ExecuteBatch(const std::vector<std::shared_ptr<Surface>>& src, const std::vector<std::shared_ptr<Surface>>& dst) {
    if (src.size() != dst.size() || src.empty()) return false;
    std::vector<Surface*> a, b;
    for (auto& s : src) a.push_back(s.get());
    for (auto& d : dst) b.push_back(d.get());
    return TASK_EXEC_SUCCESS == rs_->RunBatch(a.data(), b.data(), (uint32_t)a.size());
  }
};

class PySurfaceRemaper {
  std::unique_ptr<RemapSurface> rm_;
  Pixel_Format fmt_;

public:
  PySurfaceRemaper(py::array_t<float, py::array::c_style | py::array::forcecast>& x, py::array_t<float, py::array::c_style | py::array::forcecast>& y,
                   Pixel_Format f, HipContext ctx, HipStream str)
      : fmt_(f) {
    if (x.ndim() != 2 || y.ndim() != 2 || x.shape(0) != y.shape(0) || x.shape(1) != y.shape(1))
      throw std::runtime_error("x_map and y_map must be 2-D float32 arrays of the same shape");
    rm_.reset(RemapSurface::Make(x.data(), y.data(), (uint32_t)x.shape(1), (uint32_t)x.shape(0), f, ctx, str));
  }
  Pixel_Format GetFormat() const { return fmt_; }
  void SetAsync(bool on) { rm_->SetAsync(on); }
  bool GetAsync() const { return rm_->GetAsync(); }
  std::shared_ptr<Surface> Execute(std::shared_ptr<Surface> src) {
    if (!src) return empty_surface(fmt_);
    rm_->SetInput(src.get(), 0U);
    if (TASK_EXEC_SUCCESS != rm_->Execute()) return empty_surface(fmt_);
    auto* out = static_cast<Surface*>(rm_->GetOutput(0U));
    return std::shared_ptr<Surface>(out ? out->Clone() : Surface::Make(fmt_));
  }
  // additive: the same maps applied to n surfaces into n caller-owned surfaces, one dispatch per 32 frames
  bool ExecuteBatch(const std::vector<std::shared_ptr<Surface>>& src, const std::vector<std::shared_ptr<Surface>>& dst) {
    if (src.size() != dst.size() || src.empty()) return false;
    std::vector<Surface*> a, b;
    for (auto& s : src) a.push_back(s.get());
    for (auto& d : dst) b.push_back(d.get());
    return TASK_EXEC_SUCCESS == rm_->RunBatch(a.data(), b.data(), (uint32_t)a.size());
  }
};

// The numpy array that OWNS a frame's memory (the end of the .base chain), if the frame is a plain view of it: the object HostPinCache may
// key a page-locked registration on — its death (a weak-reference callback) is when the memory can go away.  nullptr: nobody to vouch for the
// memory (a bytes / bytearray / mmap / foreign buffer at the root, an array that does not own its data): the upload stages a copy as before.
static PyObject* frame_owner(const py::array& f) {
  py::handle o = f;
  for (int depth = 0; depth < 8; depth++) {
    if (!py::isinstance<py::array>(o)) return nullptr;
    py::object base = py::reinterpret_borrow<py::array>(o).base();
    if (!base || base.is_none()) return py::reinterpret_borrow<py::array>(o).owndata() ? o.ptr() : nullptr;
    o = base;  // (borrowed: the chain is kept alive by `f` for the duration of the call)
    if (py::isinstance<py::capsule>(o)) return nullptr;  // AllocPinned(): page-locked already
  }
  return nullptr;
}
// Does the buffer own its PAGES?  hipHostRegister / hipHostUnregister work on whole pages, and unregistering tears the GPU mapping of every page of
// the range down — also for whatever else lives in the range's first and last page.  The HIP runtime page-locks pageable memory on the fly for its
// own copies (torch's .cpu(), hipMemcpy from a numpy array) and keeps track of what it has locked: when a buffer this cache releases shares a page
// with memory the runtime believes locked, the runtime's next copy there faults on the GPU ("Memory access fault by GPU ... on address <host page>":
// seen in 3 of 19 long single-process test runs, profiles/r06_pin_cache_fault.txt).  So only buffers that own their pages are taken: glibc serves
// large requests by mmap — a chunk of whole pages of its own, user pointer 16 bytes in, IS_MMAPPED set in the size word in front of it (read only
// when the pointer sits 16 bytes into a page: the word is then on the same, mapped, page).  Heap-cut buffers keep the staged copy.
static bool owns_its_pages(const void* data, size_t bytes) {
  const uintptr_t p = reinterpret_cast<uintptr_t>(data);
  if ((p & 4095u) != 16u) return false;
  const size_t word = reinterpret_cast<const size_t*>(p)[-1], chunk = word & ~(size_t)7;
  return (word & 2u) != 0 && (chunk & 4095u) == 0 && chunk >= bytes + 16u;
}
static void vouch_for_frame(const py::array& f, const void* data, size_t bytes, int device) {
  if (!owns_its_pages(data, bytes)) return;
  PyObject* owner = frame_owner(f);
  if (!owner) return;
  const uint64_t id = (uint64_t)(uintptr_t)owner;
  if (HostPinCache::note_use(data, bytes, id, device) != HostPinCache::kFirstSight) return;
  // the owner's first buffer: from now on its death unregisters whatever was page-locked on its behalf (before numpy frees the memory:
  // weak-reference callbacks run before tp_dealloc releases the data)
  py::weakref(py::handle(owner), py::cpp_function([id](py::handle wr) {
                HostPinCache::owner_gone(id);
                wr.dec_ref();
              })).release();
}

class PyFrameUploader {
  std::unique_ptr<CudaUploadFrame> up_;
  Pixel_Format fmt_;
  int dev_ = 0;

public:
  PyFrameUploader(uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str) : fmt_(f) {
    up_.reset(CudaUploadFrame::Make(str, ctx, w, h, f));
    dev_ = DeviceOfContext(ctx);  // (-1: whatever device is current)
  }
  Pixel_Format GetFormat() const { return fmt_; }
  int Device() const { return dev_; }
  void SetAsync(bool on, bool in_place) { up_->SetAsync(on); up_->SetAsyncInPlace(on && in_place); }
  bool GetAsync() const { return up_->GetAsync(); }
  bool Vouches() const { return !up_->GetAsync() || up_->GetAsyncInPlace(); }  // uploads that may read a caller frame in place: blocking ones, asynchronous ones on the caller's promise
  std::shared_ptr<Surface> Upload(void* data, size_t bytes) {
    py::gil_scoped_release nogil;  // the reference releases the GIL around uploads too (PyFrameUploader.cpp:118-160)
    std::unique_ptr<Buffer> raw(Buffer::Make(bytes, data));
    up_->SetInput(raw.get(), 0U);
    const auto res = up_->Execute();
    up_->ClearInputs();
    auto* s = static_cast<Surface*>(up_->GetOutput(0U));
    if (TASK_EXEC_FAIL == res || !s) throw std::runtime_error("Error uploading frame to GPU");
    return std::shared_ptr<Surface>(s->Clone());
  }
};

class PySurfaceDownloader {
  std::unique_ptr<CudaDownloadSurface> dl_;
  Pixel_Format fmt_;

public:
  PySurfaceDownloader(uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str) : fmt_(f) {
    dl_.reset(CudaDownloadSurface::Make(str, ctx, w, h, f));
  }
  Pixel_Format GetFormat() const { return fmt_; }
  template <typename T>
  bool Download(std::shared_ptr<Surface> s, py::array_t<T>& frame) {
    if (!s || s->Empty()) return false;
    const size_t bytes = s->HostMemSize();
    if (bytes != (size_t)frame.size() * sizeof(T)) frame.resize({(py::ssize_t)(bytes / sizeof(T))}, false);
    T* dst = frame.mutable_data();
    // the DMA and (for pageable arrays) the single host copy run without the GIL; arrays from AllocPinned() receive
    // the DMA directly
    py::gil_scoped_release nogil;
    return TASK_EXEC_SUCCESS == dl_->DownloadInto(s.get(), dst, bytes);
  }
};

class PyBufferUploader {
  std::unique_ptr<UploadBuffer> up_;

public:
  PyBufferUploader(uint32_t e, uint32_t n, HipContext ctx, HipStream str) { up_.reset(UploadBuffer::Make(str, ctx, e, n)); }
  std::shared_ptr<CudaBuffer> Upload(py::array_t<uint8_t>& a) {
    std::unique_ptr<Buffer> raw(Buffer::Make((size_t)a.size(), a.mutable_data()));
    up_->SetInput(raw.get(), 0U);
    const auto res = up_->Execute();
    up_->ClearInputs();
    auto* b = static_cast<CudaBuffer*>(up_->GetOutput(0U));
    if (TASK_EXEC_FAIL == res || !b) throw std::runtime_error("Error uploading frame to GPU");
    return std::shared_ptr<CudaBuffer>(b->Clone());
  }
};

class PyCudaBufferDownloader {
  std::unique_ptr<DownloadCudaBuffer> dl_;

public:
  PyCudaBufferDownloader(uint32_t e, uint32_t n, HipContext ctx, HipStream str) { dl_.reset(DownloadCudaBuffer::Make(str, ctx, e, n)); }
  bool Download(std::shared_ptr<CudaBuffer> b, py::array_t<uint8_t>& a) {
    if (!b) return false;
    dl_->SetInput(b.get(), 0U);
    if (TASK_EXEC_FAIL == dl_->Execute()) return false;
    auto* raw = static_cast<Buffer*>(dl_->GetOutput(0U));
    if (!raw) return false;
    const size_t bytes = b->GetRawMemSize();
    if (bytes != (size_t)a.size()) a.resize({(py::ssize_t)bytes}, false);
    std::memcpy(a.mutable_data(), raw->GetRawMemPtr(), bytes);
    return true;
  }
};
#ifdef VPF_WITH_LIBAV
// PyFfmpegDecoder (reference: src/PyNvCodec/src/PyFFMpegDecoder.cpp:37-70,220-268): software decode on the host, frames
// enter the GPU through the pinned-staging uploader.  Only compiled where libav exists.
class PyFfmpegDecoder {
  std::unique_ptr<FfmpegFeeder> dec_;
  std::unique_ptr<PyFrameUploader> up_;
  std::vector<uint8_t> host_;
  int gpu_;

public:
  PyFfmpegDecoder(const std::string& url, const std::map<std::string, std::string>& opts, int gpu) : gpu_(gpu) {
    dec_.reset(new FfmpegFeeder(url, opts));
  }
  uint32_t Width() const { return dec_->Width(); }
  uint32_t Height() const { return dec_->Height(); }
  double Framerate() const { return dec_->Framerate(); }
  ColorSpace GetColorSpace() const { return dec_->GetColorSpace(); }
  ColorRange GetColorRange() const { return dec_->GetColorRange(); }
  Pixel_Format GetPixelFormat() const { return dec_->GetPixelFormat(); }
  uint32_t up_w_ = 0, up_h_ = 0;
  // uploaders of earlier resolutions, keyed by size: surfaces already handed out alias their memory, and a stream that alternates between
  // resolutions gets its old uploader back instead of a new pair of pinned + device buffers per switch
  std::map<std::pair<uint32_t, uint32_t>, std::unique_ptr<PyFrameUploader>> parked_;

  bool DecodeSingleFrame(py::array_t<uint8_t>& frame) {
    if (!dec_->NextFrame()) return false;
    const size_t n = dec_->PendingFrameBytes();  // sized from the decoded picture: a stream may change resolution mid-way
    if ((size_t)frame.size() != n) frame.resize({(py::ssize_t)n}, false);
    return dec_->CopyFrameNV12(frame.mutable_data(), n);
  }
  std::shared_ptr<Surface> DecodeSingleSurface() {
    if (!dec_->NextFrame()) return empty_surface(NV12);
    const uint32_t w = dec_->FrameWidth(), h = dec_->FrameHeight();
    host_.resize(dec_->PendingFrameBytes());
    if (!dec_->CopyFrameNV12(host_.data(), host_.size())) return empty_surface(NV12);
    if (!up_ || w != up_w_ || h != up_h_) {
      if (up_) parked_[{up_w_, up_h_}] = std::move(up_);
      auto it = parked_.find({w, h});
      if (it != parked_.end()) { up_ = std::move(it->second); parked_.erase(it); }
      else up_.reset(new PyFrameUploader(w, h, NV12, ctx_of(gpu_), str_of(gpu_)));
      up_w_ = w; up_h_ = h;
    }
    return up_->Upload(host_.data(), host_.size());
  }
};
#endif

}  // namespace

PYBIND11_MODULE(_PyNvCodec, m) {
  m.doc() = "MI355X-native surface conversion behind VPF's Python API (hand-written HIP kernels via libvpfhip)";

  py::enum_<Pixel_Format>(m, "PixelFormat")
      .value("Y", Y).value("RGB", RGB).value("NV12", NV12).value("YUV420", YUV420).value("RGB_PLANAR", RGB_PLANAR)
      .value("BGR", BGR).value("YCBCR", YCBCR).value("YUV444", YUV444).value("YUV444_10bit", YUV444_10bit)
      .value("YUV420_10bit", YUV420_10bit).value("UNDEFINED", UNDEFINED).value("RGB_32F", RGB_32F)
      .value("RGB_32F_PLANAR", RGB_32F_PLANAR).value("YUV422", YUV422).value("P10", P10).value("P12", P12)
      .export_values();
  py::enum_<ColorSpace>(m, "ColorSpace").value("BT_601", BT_601).value("BT_709", BT_709).value("UNSPEC", UNSPEC).export_values();
  py::enum_<ColorRange>(m, "ColorRange").value("MPEG", MPEG).value("JPEG", JPEG).value("UDEF", UDEF).export_values();

  py::class_<ColorspaceConversionContext, std::shared_ptr<ColorspaceConversionContext>>(m, "ColorspaceConversionContext")
      .def(py::init<>())
      .def(py::init<ColorSpace, ColorRange>(), py::arg("color_space"), py::arg("color_range"))
      .def_readwrite("color_space", &ColorspaceConversionContext::color_space)
      .def_readwrite("color_range", &ColorspaceConversionContext::color_range);

  py::class_<CudaBuffer, std::shared_ptr<CudaBuffer>>(m, "CudaBuffer")
      .def("GetRawMemSize", &CudaBuffer::GetRawMemSize)
      .def("GetNumElems", &CudaBuffer::GetNumElems)
      .def("GetElemSize", &CudaBuffer::GetElemSize)
      .def("GpuMem", &CudaBuffer::GpuMem)
      .def("Clone", [](std::shared_ptr<CudaBuffer> self) { return std::shared_ptr<CudaBuffer>(self->Clone()); })
      .def("CopyFrom",
           [](std::shared_ptr<CudaBuffer> self, std::shared_ptr<CudaBuffer> other, size_t ctx, size_t str) {
             if (self->GetRawMemSize() != other->GetRawMemSize()) throw std::runtime_error("Buffers have different size.");
             std::unique_ptr<SurfacePlane> d(new SurfacePlane((uint32_t)self->GetRawMemSize(), 1, (uint32_t)self->GetRawMemSize(), 1, self->GpuMem()));
             d->Import(other->GpuMem(), (uint32_t)other->GetRawMemSize(), (HipContext)ctx, (HipStream)str);
           },
           py::arg("other"), py::arg("context"), py::arg("stream"))
      .def("CopyFrom",
           [](std::shared_ptr<CudaBuffer> self, std::shared_ptr<CudaBuffer> other, int gpu) {
             if (self->GetRawMemSize() != other->GetRawMemSize()) throw std::runtime_error("Buffers have different size.");
             std::unique_ptr<SurfacePlane> d(new SurfacePlane((uint32_t)self->GetRawMemSize(), 1, (uint32_t)self->GetRawMemSize(), 1, self->GpuMem()));
             d->Import(other->GpuMem(), (uint32_t)other->GetRawMemSize(), ctx_of(gpu), str_of(gpu));
           },
           py::arg("other"), py::arg("gpu_id"))
      .def_static("Make", [](uint32_t e, uint32_t n, int gpu) { return std::shared_ptr<CudaBuffer>(CudaBuffer::Make(e, n, ctx_of(gpu))); },
                  py::arg("elem_size"), py::arg("num_elems"), py::arg("gpu_id"));

  py::class_<SurfacePlane, std::shared_ptr<SurfacePlane>>(m, "SurfacePlane")
      .def("Width", &SurfacePlane::Width)
      .def("Height", &SurfacePlane::Height)
      .def("Pitch", &SurfacePlane::Pitch)
      .def("GpuMem", &SurfacePlane::GpuMem)
      .def("ElemSize", &SurfacePlane::ElemSize)
      .def("HostFrameSize", &SurfacePlane::GetHostMemSize)
      .def("Import", [](std::shared_ptr<SurfacePlane> self, size_t src, uint32_t pitch, int gpu) { self->Import((DevicePtr)src, pitch, ctx_of(gpu), str_of(gpu)); },
           py::arg("src"), py::arg("src_pitch"), py::arg("gpu_id"))
      .def("Import", [](std::shared_ptr<SurfacePlane> self, size_t src, uint32_t pitch, size_t ctx, size_t str) { self->Import((DevicePtr)src, pitch, (HipContext)ctx, (HipStream)str); },
           py::arg("src"), py::arg("src_pitch"), py::arg("context"), py::arg("stream"))
      .def("Export", [](std::shared_ptr<SurfacePlane> self, size_t dst, uint32_t pitch, int gpu) { self->Export((DevicePtr)dst, pitch, ctx_of(gpu), str_of(gpu)); },
           py::arg("dst"), py::arg("dst_pitch"), py::arg("gpu_id"))
      .def("Export", [](std::shared_ptr<SurfacePlane> self, size_t dst, uint32_t pitch, size_t ctx, size_t str) { self->Export((DevicePtr)dst, pitch, (HipContext)ctx, (HipStream)str); },
           py::arg("dst"), py::arg("dst_pitch"), py::arg("context"), py::arg("stream"))
      .def("__repr__", [](std::shared_ptr<SurfacePlane> self) { return plane_repr(self.get()); });

  py::class_<Surface, std::shared_ptr<Surface>>(m, "Surface", py::dynamic_attr())
      // additive: non-owning Surface over memory someone else owns (a torch tensor, a decoder's frame pool); the
      // reference has the equivalent only in C++ (SurfaceNV12(width, height, pitch, ptr), MemoryInterfaces.cpp:822-826)
      .def_static("Wrap",
                  [](Pixel_Format f, uint32_t w, uint32_t h, uint32_t pitch, size_t ptr) {
                    Surface* s = Surface::Make(f, w, h, pitch, (DevicePtr)ptr);
                    if (!s) throw std::invalid_argument("Surface.Wrap: format must live in ONE pitched allocation (Y, NV12, P10, P12, RGB, BGR, RGB_PLANAR, YUV444, RGB_32F, RGB_32F_PLANAR)");
                    return std::shared_ptr<Surface>(s);
                  },
                  py::arg("format"), py::arg("width"), py::arg("height"), py::arg("pitch"), py::arg("ptr"))
      .def("Width", &Surface::Width, py::arg("plane") = 0U)
      .def("Height", &Surface::Height, py::arg("plane") = 0U)
      .def("Pitch", &Surface::Pitch, py::arg("plane") = 0U)
      .def("Format", &Surface::PixelFormat)
      .def("Empty", &Surface::Empty)
      .def("NumPlanes", &Surface::NumPlanes)
      .def("HostSize", &Surface::HostMemSize)
      .def("OwnMemory", &Surface::OwnMemory)
      .def_static("Make", [](Pixel_Format f, uint32_t w, uint32_t h, int gpu) { return make_surface(f, w, h, ctx_of(gpu)); },
                  py::arg("format"), py::arg("width"), py::arg("height"), py::arg("gpu_id"))
      .def_static("Make", [](Pixel_Format f, uint32_t w, uint32_t h, size_t ctx) { return make_surface(f, w, h, (HipContext)ctx); },
                  py::arg("format"), py::arg("width"), py::arg("height"), py::arg("context"))
      .def("PlanePtr",
           [](std::shared_ptr<Surface> self, int plane) {
             SurfacePlane* p = self->GetSurfacePlane((uint32_t)plane);
             if (!p) throw std::invalid_argument("Invalid plane number");
             return std::make_shared<SurfacePlane>(*p);  // non-owning copy
           },
           py::arg("plane") = 0U)
      // Drop-in semantics: the reference's binding copies SELF -> OTHER (PySurface.cpp:54-81 takes (self, other) as (source,
      // destination) and :361,382 pass them in that order) — the opposite of what the name suggests, but code written against the
      // reference relies on it, so it is kept.  UpdateFrom (additive) is the unambiguous other -> self copy.
      .def("CopyFrom",
           [](std::shared_ptr<Surface> self, std::shared_ptr<Surface> other, int gpu) {
             check_same(*self, *other);
             copy_surface(*self, *other, ctx_of(gpu), str_of(gpu));
           },
           py::arg("other"), py::arg("gpu_id"), "DtoD copy of THIS surface INTO `other` (the reference's direction, PySurface.cpp:361)")
      .def("CopyFrom",
           [](std::shared_ptr<Surface> self, std::shared_ptr<Surface> other, size_t ctx, size_t str) {
             check_same(*self, *other);
             copy_surface(*self, *other, (HipContext)ctx, (HipStream)str);
           },
           py::arg("other"), py::arg("context"), py::arg("stream"), "DtoD copy of THIS surface INTO `other` (the reference's direction, PySurface.cpp:382)")
      .def("UpdateFrom",
           [](std::shared_ptr<Surface> self, std::shared_ptr<Surface> src, int gpu) {
             check_same(*self, *src);
             copy_surface(*src, *self, ctx_of(gpu), str_of(gpu));
           },
           py::arg("src"), py::arg("gpu_id"), "additive: DtoD copy of `src` into THIS surface")
      .def("UpdateFrom",
           [](std::shared_ptr<Surface> self, std::shared_ptr<Surface> src, size_t ctx, size_t str) {
             check_same(*self, *src);
             copy_surface(*src, *self, (HipContext)ctx, (HipStream)str);
           },
           py::arg("src"), py::arg("context"), py::arg("stream"), "additive: DtoD copy of `src` into THIS surface")
      .def("Clone",
           [](std::shared_ptr<Surface> self) {
             auto n = make_surface(self->PixelFormat(), self->Width(), self->Height(), self->Context());
             copy_surface(*self, *n, self->Context(), nullptr);
             return n;
           },
           py::call_guard<py::gil_scoped_release>())
      .def("Clone",
           [](std::shared_ptr<Surface> self, int gpu) {
             auto n = make_surface(self->PixelFormat(), self->Width(), self->Height(), ctx_of(gpu));
             copy_surface(*self, *n, ctx_of(gpu), str_of(gpu));
             return n;
           },
           py::arg("gpu_id"), py::call_guard<py::gil_scoped_release>())
      .def("Clone",
           [](std::shared_ptr<Surface> self, size_t ctx, size_t str) {
             auto n = make_surface(self->PixelFormat(), self->Width(), self->Height(), (HipContext)ctx);
             copy_surface(*self, *n, (HipContext)ctx, (HipStream)str);
             return n;
           },
           py::arg("context"), py::arg("stream"), py::call_guard<py::gil_scoped_release>())
      .def("Crop",
           [](std::shared_ptr<Surface> self, uint32_t x, uint32_t y, uint32_t w, uint32_t h, int gpu) {
             auto n = make_surface(self->PixelFormat(), w, h, ctx_of(gpu));
             self->Export(*n, ctx_of(gpu), str_of(gpu), x, y, w, h, 0U, 0U);
             return n;
           },
           py::arg("x"), py::arg("y"), py::arg("w"), py::arg("h"), py::arg("gpu_id"), py::call_guard<py::gil_scoped_release>())
      .def("Crop",
           [](std::shared_ptr<Surface> self, uint32_t x, uint32_t y, uint32_t w, uint32_t h, size_t ctx, size_t str) {
             auto n = make_surface(self->PixelFormat(), w, h, (HipContext)ctx);
             self->Export(*n, (HipContext)ctx, (HipStream)str, x, y, w, h, 0U, 0U);
             return n;
           },
           py::arg("x"), py::arg("y"), py::arg("w"), py::arg("h"), py::arg("context"), py::arg("stream"),
           py::call_guard<py::gil_scoped_release>())
      .def("__repr__", [](std::shared_ptr<Surface> self) { return surface_repr(self.get()); });

  py::class_<PySurfaceConverter>(m, "PySurfaceConverter")
      .def(py::init([](uint32_t w, uint32_t h, Pixel_Format in, Pixel_Format out, uint32_t gpu) {
             return new PySurfaceConverter(w, h, in, out, ctx_of((int)gpu), str_of((int)gpu));
           }),
           py::arg("width"), py::arg("height"), py::arg("src_format"), py::arg("dst_format"), py::arg("gpu_id"))
      .def(py::init([](uint32_t w, uint32_t h, Pixel_Format in, Pixel_Format out, size_t ctx, size_t str) {
             return new PySurfaceConverter(w, h, in, out, (HipContext)ctx, (HipStream)str);
           }),
           py::arg("width"), py::arg("height"), py::arg("src_format"), py::arg("dst_format"), py::arg("context"), py::arg("stream"))
      .def("Format", &PySurfaceConverter::GetFormat)
      .def("SetOutputReuseHint", &PySurfaceConverter::SetOutputReuseHint, py::arg("on"),
           "additive: the result is consumed by the next kernel of a per-frame chain -> keep it in the Infinity Cache (identical pixels)")
      .def("GetOutputReuseHint", &PySurfaceConverter::GetOutputReuseHint)
      // the returned Surface aliases memory owned by the converter: keep the converter alive as long as it lives
      .def("Execute", &PySurfaceConverter::Execute, py::arg("src"), py::arg("cc_ctx") = nullptr, py::keep_alive<0, 1>(),
           py::call_guard<py::gil_scoped_release>())
      .def("ExecuteBatch", &PySurfaceConverter::ExecuteBatch, py::arg("src"), py::arg("dst"), py::arg("cc_ctx") = nullptr,
           py::call_guard<py::gil_scoped_release>());

  py::class_<PySurfaceConvertResizer>(m, "PySurfaceConvertResizer",
                                      "Additive: NV12 / YUV420 -> bilinear resize -> RGB / BGR / RGB_PLANAR in one pass; bit-identical to "
                                      "PySurfaceConverter followed by PySurfaceResizer (bilinear).")
      .def(py::init([](uint32_t sw, uint32_t sh, Pixel_Format in, uint32_t dw, uint32_t dh, Pixel_Format out, uint32_t gpu) {
             return new PySurfaceConvertResizer(sw, sh, in, dw, dh, out, ctx_of((int)gpu), str_of((int)gpu));
           }),
           py::arg("src_width"), py::arg("src_height"), py::arg("src_format"), py::arg("dst_width"), py::arg("dst_height"),
           py::arg("dst_format"), py::arg("gpu_id"))
      .def(py::init([](uint32_t sw, uint32_t sh, Pixel_Format in, uint32_t dw, uint32_t dh, Pixel_Format out, size_t ctx, size_t str) {
             return new PySurfaceConvertResizer(sw, sh, in, dw, dh, out, (HipContext)ctx, (HipStream)str);
           }),
           py::arg("src_width"), py::arg("src_height"), py::arg("src_format"), py::arg("dst_width"), py::arg("dst_height"),
           py::arg("dst_format"), py::arg("context"), py::arg("stream"))
      .def("Format", &PySurfaceConvertResizer::GetFormat)
      .def("Execute", &PySurfaceConvertResizer::Execute, py::arg("src"), py::arg("cc_ctx") = nullptr, py::keep_alive<0, 1>(),
           py::call_guard<py::gil_scoped_release>())
      .def("ExecuteBatch", &PySurfaceConvertResizer::ExecuteBatch, py::arg("src"), py::arg("dst"), py::arg("cc_ctx") = nullptr,
           py::call_guard<py::gil_scoped_release>());

  py::class_<PySurfaceResizer>(m, "PySurfaceResizer")
      .def(py::init([](uint32_t w, uint32_t h, Pixel_Format f, uint32_t gpu) { return new PySurfaceResizer(w, h, f, ctx_of((int)gpu), str_of((int)gpu)); }),
           py::arg("width"), py::arg("height"), py::arg("format"), py::arg("gpu_id"))
      .def(py::init([](uint32_t w, uint32_t h, Pixel_Format f, size_t ctx, size_t str) { return new PySurfaceResizer(w, h, f, (HipContext)ctx, (HipStream)str); }),
           py::arg("width"), py::arg("height"), py::arg("format"), py::arg("context"), py::arg("stream"))
      .def("Format", &PySurfaceResizer::GetFormat)
      .def("SetInterpolation", &PySurfaceResizer::SetInterpolation, py::arg("interp"),
           "additive: 0 nearest, 1 bilinear, 2 Lanczos-3 (default: the filter the reference requests from NPP)")
      .def("GetInterpolation", &PySurfaceResizer::GetInterpolation)
      .def("SetAsync", &PySurfaceResizer::SetAsync, py::arg("on"),
           "additive: Execute() no longer waits for the stream (like PySurfaceConverter); default False = the reference's blocking behaviour")
      .def("GetAsync", &PySurfaceResizer::GetAsync)
      .def("Execute", &PySurfaceResizer::Execute, py::arg("src"), py::keep_alive<0, 1>(), py::call_guard<py::gil_scoped_release>())
      .def("ExecuteBatch", &PySurfaceResizer::ExecuteBatch, py::arg("src"), py::arg("dst"), py::call_guard<py::gil_scoped_release>(),
           "additive: n same-shape surfaces into n caller-owned surfaces, all planes of all frames in as few dispatches as possible; asynchronous on the resizer's stream");

  using FMap = py::array_t<float, py::array::c_style | py::array::forcecast>;
  py::class_<PySurfaceRemaper>(m, "PySurfaceRemaper")
      .def(py::init([](FMap& x, FMap& y, Pixel_Format f, uint32_t gpu) { return new PySurfaceRemaper(x, y, f, ctx_of((int)gpu), str_of((int)gpu)); }),
           py::arg("x_map"), py::arg("y_map"), py::arg("format"), py::arg("gpu_id"))
      .def(py::init([](FMap& x, FMap& y, Pixel_Format f, size_t ctx, size_t str) { return new PySurfaceRemaper(x, y, f, (HipContext)ctx, (HipStream)str); }),
           py::arg("x_map"), py::arg("y_map"), py::arg("format"), py::arg("context"), py::arg("stream"))
      .def("Format", &PySurfaceRemaper::GetFormat)
      .def("SetAsync", &PySurfaceRemaper::SetAsync, py::arg("on"), "additive: Execute() no longer waits for the stream; default False = the reference's blocking behaviour")
      .def("GetAsync", &PySurfaceRemaper::GetAsync)
      .def("Execute", &PySurfaceRemaper::Execute, py::arg("src"), py::keep_alive<0, 1>(), py::call_guard<py::gil_scoped_release>())
      .def("ExecuteBatch", &PySurfaceRemaper::ExecuteBatch, py::arg("src"), py::arg("dst"), py::call_guard<py::gil_scoped_release>(),
           "additive: the remaper's maps applied to n surfaces into n caller-owned surfaces, one dispatch per 32 frames; asynchronous on the remaper's stream");

  py::class_<PyFrameUploader>(m, "PyFrameUploader")
      .def(py::init([](uint32_t w, uint32_t h, Pixel_Format f, uint32_t gpu) { return new PyFrameUploader(w, h, f, ctx_of((int)gpu), str_of((int)gpu)); }),
           py::arg("width"), py::arg("height"), py::arg("format"), py::arg("gpu_id"))
      .def(py::init([](uint32_t w, uint32_t h, Pixel_Format f, size_t ctx, size_t str) { return new PyFrameUploader(w, h, f, (HipContext)ctx, (HipStream)str); }),
           py::arg("width"), py::arg("height"), py::arg("format"), py::arg("context"), py::arg("stream"))
      .def("Format", &PyFrameUploader::GetFormat)
      .def("SetAsync", &PyFrameUploader::SetAsync, py::arg("on"), py::arg("in_place") = false,
           "additive: True = UploadSingleFrame returns once the copy is queued (the surface is valid in stream order on the uploader's stream; a "
           "page-locked source frame must stay untouched until that stream is synchronised).  Default False = wait for the copy, like the reference. "
           "VPF_HIP_UPLOAD_ASYNC=1 makes True the default.  in_place=True: ordinary numpy frames that come back may be page-locked and read where they "
           "lie as well (the caller promises for them what it promises for AllocPinned() frames); default: they are copied out before the call returns")
      .def("GetAsync", &PyFrameUploader::GetAsync)
      .def("UploadSingleFrame", [](PyFrameUploader& self, py::array_t<uint8_t>& f) { if (self.Vouches()) vouch_for_frame(f, f.data(), (size_t)f.size(), self.Device()); return self.Upload(f.mutable_data(), (size_t)f.size()); },
           py::arg("frame").noconvert(true), py::keep_alive<0, 1>())
      .def("UploadSingleFrame", [](PyFrameUploader& self, py::array_t<float>& f) { if (self.Vouches()) vouch_for_frame(f, f.data(), (size_t)f.size() * sizeof(float), self.Device()); return self.Upload(f.mutable_data(), (size_t)f.size() * sizeof(float)); },
           py::arg("frame").noconvert(true), py::keep_alive<0, 1>())
      .def("UploadSingleFrame", [](PyFrameUploader& self, py::array_t<uint16_t>& f) { if (self.Vouches()) vouch_for_frame(f, f.data(), (size_t)f.size() * sizeof(uint16_t), self.Device()); return self.Upload(f.mutable_data(), (size_t)f.size() * sizeof(uint16_t)); },
           py::arg("frame").noconvert(true), py::keep_alive<0, 1>());

  py::class_<PySurfaceDownloader>(m, "PySurfaceDownloader")
      .def(py::init([](uint32_t w, uint32_t h, Pixel_Format f, uint32_t gpu) { return new PySurfaceDownloader(w, h, f, ctx_of((int)gpu), str_of((int)gpu)); }),
           py::arg("width"), py::arg("height"), py::arg("format"), py::arg("gpu_id"))
      .def(py::init([](uint32_t w, uint32_t h, Pixel_Format f, size_t ctx, size_t str) { return new PySurfaceDownloader(w, h, f, (HipContext)ctx, (HipStream)str); }),
           py::arg("width"), py::arg("height"), py::arg("format"), py::arg("context"), py::arg("stream"))
      .def("Format", &PySurfaceDownloader::GetFormat)
      .def("DownloadSingleSurface", &PySurfaceDownloader::Download<uint8_t>, py::arg("surface"), py::arg("frame").noconvert(true))
      .def("DownloadSingleSurface", &PySurfaceDownloader::Download<float>, py::arg("surface"), py::arg("frame").noconvert(true))
      .def("DownloadSingleSurface", &PySurfaceDownloader::Download<uint16_t>, py::arg("surface"), py::arg("frame").noconvert(true));

  py::class_<PyBufferUploader>(m, "PyBufferUploader")
      .def(py::init([](uint32_t e, uint32_t n, uint32_t gpu) { return new PyBufferUploader(e, n, ctx_of((int)gpu), str_of((int)gpu)); }),
           py::arg("elem_size"), py::arg("num_elems"), py::arg("gpu_id"))
      .def(py::init([](uint32_t e, uint32_t n, size_t ctx, size_t str) { return new PyBufferUploader(e, n, (HipContext)ctx, (HipStream)str); }),
           py::arg("elem_size"), py::arg("num_elems"), py::arg("context"), py::arg("stream"))
      .def("UploadSingleBuffer", &PyBufferUploader::Upload, py::arg("array"));

  py::class_<PyCudaBufferDownloader>(m, "PyCudaBufferDownloader")
      .def(py::init([](uint32_t e, uint32_t n, uint32_t gpu) { return new PyCudaBufferDownloader(e, n, ctx_of((int)gpu), str_of((int)gpu)); }),
           py::arg("elem_size"), py::arg("num_elems"), py::arg("gpu_id"))
      .def(py::init([](uint32_t e, uint32_t n, size_t ctx, size_t str) { return new PyCudaBufferDownloader(e, n, (HipContext)ctx, (HipStream)str); }),
           py::arg("elem_size"), py::arg("num_elems"), py::arg("context"), py::arg("stream"))
      .def("DownloadSingleCudaBuffer", &PyCudaBufferDownloader::Download, py::arg("buffer"), py::arg("array"));

#ifdef VPF_WITH_LIBAV
  py::class_<PyFfmpegDecoder>(m, "PyFfmpegDecoder")
      .def(py::init<const std::string&, const std::map<std::string, std::string>&, int>(), py::arg("input"), py::arg("opts"),
           py::arg("gpu_id") = 0)
      .def("Width", &PyFfmpegDecoder::Width)
      .def("Height", &PyFfmpegDecoder::Height)
      .def("Framerate", &PyFfmpegDecoder::Framerate)
      .def("ColorSpace", &PyFfmpegDecoder::GetColorSpace)
      .def("ColorRange", &PyFfmpegDecoder::GetColorRange)
      .def("Format", &PyFfmpegDecoder::GetPixelFormat)
      .def("DecodeSingleFrame", &PyFfmpegDecoder::DecodeSingleFrame, py::arg("frame"))
      .def("DecodeSingleSurface", &PyFfmpegDecoder::DecodeSingleSurface, py::keep_alive<0, 1>());
  m.attr("HAVE_LIBAV") = true;
#else
  m.attr("HAVE_LIBAV") = false;
#endif
  m.def("GetNumGpus", []() { return (int)HipResMgr::Instance().GetNumGpus(); });
  // --- additive helpers (not in the reference) ---
  m.def("GetContext", [](int gpu) { return (size_t)ctx_of(gpu); }, py::arg("gpu_id"), "opaque context cookie of a GPU ordinal");
  m.def("GetStream", [](int gpu) { return (size_t)str_of(gpu); }, py::arg("gpu_id"), "the per-GPU non-blocking hipStream_t used by the gpu_id overloads");
  m.def("SetExtendedColorspaces", &SetExtendedColorspaces, py::arg("on"),
        "also accept colour-space/range combinations the reference's converters reject although the kernels implement them");
  m.def("ConverterPairSupport", &ConvertSurface::PairSupport, py::arg("src_format"), py::arg("dst_format"),
        "1: pair exists in the reference's ConvertSurface, 2: additive pair, 0: unsupported");
  m.def("ConverterResolve",
        [](Pixel_Format in, Pixel_Format out, std::shared_ptr<ColorspaceConversionContext> cc) -> py::object {
          int cs = 0, cr = 0;
          if (!ConvertSurface::ResolveContext(in, out, cc.get(), &cs, &cr)) return py::none();
          return py::make_tuple(cs, cr);
        },
        py::arg("src_format"), py::arg("dst_format"), py::arg("cc_ctx") = nullptr,
        "(color_space, color_range) the converter would use for this pair and context, or None if it refuses the combination");
  m.def("KernelLibraryVersion", []() { return std::string(vpf_version()); });
  m.def("AllocPinned",
        [](size_t nbytes) {
          // numpy uint8 array over hipHostMalloc memory: PyFrameUploader DMAs from it directly (no staging memcpy)
          Buffer* b = Buffer::MakeOwnMem(nbytes, (HipContext)-1);
          py::capsule owner(b, [](void* p) { delete static_cast<Buffer*>(p); });
          return py::array_t<uint8_t>({(py::ssize_t)nbytes}, {(py::ssize_t)1}, b->GetDataAs<uint8_t>(), owner);
        },
        py::arg("nbytes"), "additive: page-locked host buffer as a numpy uint8 array (decode straight into it)");
  m.def("PinCacheStats", []() {
    const HostPinCache::Stats t = HostPinCache::stats();
    py::dict d;
    d["registered"] = t.registered; d["bytes"] = t.bytes; d["in_place"] = t.hits; d["staged"] = t.staged; d["evictions"] = t.evictions; d["failures"] = t.failures; d["budget"] = t.budget;
    return d;
  }, "additive: the cache of page-locked caller frame buffers (Tasks.hpp HostPinCache): buffers registered now, their bytes, uploads DMA'd in place / "
     "staged through a copy since start, registrations given up, registrations that failed");
  m.def("PinCacheDrop", []() { HostPinCache::drop_all(); }, "additive: unregister every page-locked caller buffer");
  m.def("PinCacheSetBudgetMB", [](size_t mb) { HostPinCache::set_budget_mb(mb); }, py::arg("mb"),
        "additive: switch the cache of page-locked caller frame buffers on (mb > 0: its byte budget) or off (0, the default unless VPF_HIP_PIN_CACHE_MB is set)");
  m.def("_UseHostAllocator", [](bool on) {
    static const DeviceAllocator host = {host_alloc, host_free, nullptr};
    SetDeviceAllocator(on ? &host : nullptr);
  }, py::arg("on"), "TEST HOOK: back Surfaces with host memory so geometry/dispatch can be unit-tested without a GPU");
}
