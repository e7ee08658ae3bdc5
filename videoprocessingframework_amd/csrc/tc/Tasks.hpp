// Tasks.hpp — the Task layer of the surface path: ConvertSurface, ResizeSurface, RemapSurface and the
// upload / download tasks either side of it.  Class surface of the reference's src/TC/inc/Tasks.hpp
// (:267-338 for the three hot tasks; CudaUploadFrame / CudaDownloadSurface / UploadBuffer /
// DownloadCudaBuffer :100-200), with (HipContext, HipStream) where the reference has (CUcontext, CUstream).
// Every pixel is produced by libvpfhip (include/vpf_hip.h); there is no CPU path.
#pragma once
#include "MemoryInterfaces.hpp"

namespace VPF {

class ConvertSurface final : public Task {
public:
  // throws std::invalid_argument for an unsupported (inFormat, outFormat) pair, like the reference
  static ConvertSurface* Make(uint32_t width, uint32_t height, Pixel_Format inFormat, Pixel_Format outFormat,
                              HipContext ctx, HipStream str);
  ~ConvertSurface() override;
  // input 0: Surface, input 1: Buffer holding a ColorspaceConversionContext (optional).  Asynchronous on the
  // task's stream.  Always returns TASK_EXEC_SUCCESS; failure = null output (TasksColorCvt.cpp:1378-1391).
  TaskExecStatus Run() final;
  // Additive: convert n same-shape surfaces into n caller-provided surfaces with as few dispatches as possible.
  TaskExecStatus RunBatch(Surface* const* inputs, Surface* const* outputs, uint32_t n, const ColorspaceConversionContext* ctx);
  // 1 if the reference's ConvertSurface ctor accepts the pair (TasksColorCvt.cpp:1313-1360), 2 if it is one of
  // our additive pairs (e.g. the fused NV12 -> RGB_PLANAR), 0 otherwise
  static int PairSupport(Pixel_Format inFormat, Pixel_Format outFormat);
  // Which colour model Run() would use for this pair under `ctx` (nullptr = no context input): true + (*cs, *cr) in
  // vpf_hip.h values, or false when the combination is refused like the reference refuses it.  Host logic only
  // (tests/test_reference_tc_pin.py compares it with the reference's own dispatch code).
  static bool ResolveContext(Pixel_Format inFormat, Pixel_Format outFormat, const ColorspaceConversionContext* ctx, int* cs, int* cr);
  // Additive hint (VPF_EXEC_DST_REUSED): the output surface is consumed by the next kernel of a per-frame chain, keep
  // it in the 256 MiB Infinity Cache instead of streaming it past.  Pixels are identical either way.
  void SetOutputReused(bool reused);
  bool GetOutputReused() const;

private:
  static const uint32_t numInputs = 2U, numOutputs = 1U;
  struct Impl;
  std::unique_ptr<Impl> pImpl;  // unique_ptr: a constructor that throws after allocating it (device OOM) frees it
  ConvertSurface(uint32_t w, uint32_t h, Pixel_Format in, Pixel_Format out, HipContext ctx, HipStream str);
};

// Additive (no reference counterpart; SURVEY §8 R1 "(ii) fused NV12 -> bilinear -> RGB"): convert and bilinear-resize in
// one pass, bit-identical to ConvertSurface followed by ResizeSurface but without the full-size RGB intermediate.
// Sources NV12 / YUV420, destinations RGB / BGR / RGB_PLANAR; colour-context rules are those of the unfused pair.
class ConvertResizeSurface final : public Task {
public:
  static ConvertResizeSurface* Make(uint32_t src_width, uint32_t src_height, Pixel_Format inFormat, uint32_t dst_width,
                                    uint32_t dst_height, Pixel_Format outFormat, HipContext ctx, HipStream str);
  ~ConvertResizeSurface() override;
  TaskExecStatus Run() final;  // asynchronous like ConvertSurface; null output = failure
  TaskExecStatus RunBatch(Surface* const* inputs, Surface* const* outputs, uint32_t n, const ColorspaceConversionContext* ctx);

private:
  static const uint32_t numInputs = 2U, numOutputs = 1U;
  struct Impl;
  std::unique_ptr<Impl> pImpl;  // unique_ptr: a constructor that throws after allocating it (device OOM) frees it
  ConvertResizeSurface(uint32_t sw, uint32_t sh, Pixel_Format in, uint32_t dw, uint32_t dh, Pixel_Format out, HipContext ctx, HipStream str);
};

class ResizeSurface final : public Task {
public:
  static ResizeSurface* Make(uint32_t width, uint32_t height, Pixel_Format format, HipContext ctx, HipStream str);
  ~ResizeSurface() override;
  TaskExecStatus Run() final;  // blocking: the task registers a stream-sync callback (Tasks.cpp:1455-1456)
  // additive: n same-shape surfaces -> n caller-owned surfaces of the task's size, every plane of every frame in as few dispatches
  // as possible (vpf_resize_batch); asynchronous on the task's stream
  TaskExecStatus RunBatch(Surface* const* inputs, Surface* const* outputs, uint32_t n);
  // 0 nearest, 1 bilinear (default: BASELINE.json north_star), 2 Lanczos-3 (what the reference asks NPP for, :1190)
  void SetInterpolation(int interp);
  int GetInterpolation() const;
  // additive: Run() stops waiting for the stream (it then behaves like ConvertSurface: asynchronous on the task's stream, the caller
  // orders later work on that stream or synchronises itself).  Default false = the reference's blocking behaviour.
  void SetAsync(bool on);
  bool GetAsync() const;

private:
  static const uint32_t numInputs = 1U, numOutputs = 1U;
  struct Impl;
  std::unique_ptr<Impl> pImpl;  // unique_ptr: a constructor that throws after allocating it (device OOM) frees it
  ResizeSurface(uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str);
};

class RemapSurface final : public Task {
public:
  static RemapSurface* Make(const float* x_map, const float* y_map, uint32_t remap_w, uint32_t remap_h,
                            Pixel_Format format, HipContext ctx, HipStream str);
  ~RemapSurface() override;
  TaskExecStatus Run() final;
  // additive: the task's maps applied to n same-shape surfaces -> n caller-owned surfaces of the map's size, one dispatch per 32 frames
  TaskExecStatus RunBatch(Surface* const* inputs, Surface* const* outputs, uint32_t n);
  void SetAsync(bool on);  // additive: see ResizeSurface::SetAsync
  bool GetAsync() const;

private:
  static const uint32_t numInputs = 1U, numOutputs = 1U;
  struct Impl;
  std::unique_ptr<Impl> pImpl;  // unique_ptr: a constructor that throws after allocating it (device OOM) frees it
  RemapSurface(const float* x_map, const float* y_map, uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str);
};

// Page-locking of CALLER frame buffers that are seen again and again (round 6).  A frame in pageable memory costs the host three times its
// bytes of DRAM traffic per upload (host copy read, pinned-slot write, DMA read) and a core's time for the copy; a decoder's frame pool is a
// small fixed set of buffers, so the second time a buffer shows up it is registered (hipHostRegister) and from then on DMA'd from where it
// lies, like an AllocPinned() buffer.  The cache cannot know when a caller frees memory, so it only takes buffers somebody VOUCHES for:
// `owner` names whatever keeps the memory alive (the Python binding passes the numpy array that owns the data and calls owner_gone() from a
// weak-reference callback when that array dies; a C++ caller passes any id and calls owner_gone() before it frees).  A range that overlaps a
// registered one under another owner or other bounds evicts it first: a freed-and-reallocated buffer at an old address is a NEW buffer.
// Least recently used entries leave when the cache is full (at most 64 buffers).  OPT-IN: the budget is 0 until VPF_HIP_PIN_CACHE_MB=N or
// set_budget_mb(N) says otherwise — with the cache on, whole-suite runs on this image's ROCm aborted inside LATER pageable hipMemcpy calls of
// torch / the test harness (2 of 9 single-process runs; 0 of 16 with it off or on the round-5 tree), not root-caused: DESIGN.md §5.
// One-shot buffers keep the staged copy — and so does every ASYNCHRONOUS upload (SetAsync(true)): its contract lets the caller reuse an ordinary
// frame the moment the call returns, which only holds while the frame is copied out before that.  Process-wide, thread-safe.
class HostPinCache {
public:
  enum Use { kStaged = 0, kInPlace = 1, kFirstSight = 2 };  // kFirstSight: `owner` has no buffer here yet (the binding installs its weak reference now)
  static Use note_use(const void* p, size_t bytes, uint64_t owner, int device);
  static void owner_gone(uint64_t owner);
  static bool covers(const void* p);  // p lies in a range this cache has page-locked (not in an AllocPinned() buffer, which is the caller's own)
  static void drop_all();
  static void set_budget_mb(size_t mb);  // 0: off (and everything registered is released)
  struct Stats { uint64_t registered, bytes, hits, staged, evictions, failures, budget; };
  static Stats stats();
};

// host frame (planes concatenated at tight width) -> device Surface
class CudaUploadFrame final : public Task {
public:
  static CudaUploadFrame* Make(HipStream str, HipContext ctx, uint32_t width, uint32_t height, Pixel_Format format);
  ~CudaUploadFrame() override;
  TaskExecStatus Run() final;
  // Additive.  Default: Run() waits for the host-to-device copy, like the reference's task (src/TC/src/Tasks.cpp:617-618) — the returned
  // surface is complete for consumers on ANY stream.  SetAsync(true) (or VPF_HIP_UPLOAD_ASYNC=1): Run() returns once the copy is queued; the
  // surface is valid in stream order on the task's stream only, and a page-locked source frame must stay untouched until the caller's
  // next synchronisation of that stream.
  void SetAsync(bool on);
  bool GetAsync() const;
  // Additive, asynchronous uploads only: frames that HostPinCache has page-locked may be read IN PLACE — the caller promises what it already promises
  // for AllocPinned() frames: not to rewrite a frame before the stream has consumed it.  Default false: such frames are copied out first.
  void SetAsyncInPlace(bool on);
  bool GetAsyncInPlace() const;

private:
  static const uint32_t numInputs = 1U, numOutputs = 1U;
  struct Impl;
  std::unique_ptr<Impl> pImpl;  // unique_ptr: a constructor that throws after allocating it (device OOM) frees it
  CudaUploadFrame(HipStream str, HipContext ctx, uint32_t w, uint32_t h, Pixel_Format f);
};

// device Surface -> pinned host Buffer (planes concatenated at tight width)
class CudaDownloadSurface final : public Task {
public:
  static CudaDownloadSurface* Make(HipStream str, HipContext ctx, uint32_t width, uint32_t height, Pixel_Format format);
  ~CudaDownloadSurface() override;
  TaskExecStatus Run() final;
  // Additive: deliver the frame into caller memory (planes concatenated at tight width, like Run's output buffer).
  // Page-locked destinations (AllocPinned) receive the DMA directly; pageable ones go through the task's pinned
  // staging buffer and ONE host copy (the reference copies twice: PySurfaceDownloader.cpp:70).  Blocking.
  TaskExecStatus DownloadInto(Surface* surface, void* dst, size_t dst_bytes);

private:
  static const uint32_t numInputs = 1U, numOutputs = 1U;
  struct Impl;
  std::unique_ptr<Impl> pImpl;  // unique_ptr: a constructor that throws after allocating it (device OOM) frees it
  CudaDownloadSurface(HipStream str, HipContext ctx, uint32_t w, uint32_t h, Pixel_Format f);
};

class UploadBuffer final : public Task {
public:
  static UploadBuffer* Make(HipStream str, HipContext ctx, uint32_t elem_size, uint32_t num_elems);
  ~UploadBuffer() override;
  TaskExecStatus Run() final;

private:
  static const uint32_t numInputs = 1U, numOutputs = 1U;
  struct Impl;
  std::unique_ptr<Impl> pImpl;  // unique_ptr: a constructor that throws after allocating it (device OOM) frees it
  UploadBuffer(HipStream str, HipContext ctx, uint32_t elem_size, uint32_t num_elems);
};

class DownloadCudaBuffer final : public Task {
public:
  static DownloadCudaBuffer* Make(HipStream str, HipContext ctx, uint32_t elem_size, uint32_t num_elems);
  ~DownloadCudaBuffer() override;
  TaskExecStatus Run() final;

private:
  static const uint32_t numInputs = 1U, numOutputs = 1U;
  struct Impl;
  std::unique_ptr<Impl> pImpl;  // unique_ptr: a constructor that throws after allocating it (device OOM) frees it
  DownloadCudaBuffer(HipStream str, HipContext ctx, uint32_t elem_size, uint32_t num_elems);
};

// When true (default false, or env VPF_HIP_EXTENDED=1) ConvertSurface also accepts the colour-space /
// range combinations the reference's *_Impl::Execute reject although the kernels implement them
// (e.g. BT.601 + MPEG straight from NV12, BT.709 from YUV420).
void SetExtendedColorspaces(bool on);
bool ExtendedColorspaces();

}  // namespace VPF
