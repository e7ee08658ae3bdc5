// Tasks.cpp — see Tasks.hpp.  The reference's 24 NPP-backed *_Impl structs (src/TC/src/TasksColorCvt.cpp)
// become ONE implementation driven by a pair table: the table restates, per (in,out) pair, the default
// colour context and the combinations each reference impl accepts or rejects; the pixels come from
// vpf_convert (libvpfhip).
#include "Tasks.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "vpf_hip.h"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace VPF {

namespace {
// the reference brackets every Task::Run with an NvtxMark (src/TC/inc/Tasks.hpp:27-52, e.g. TasksColorCvt.cpp:124); here the same
// ranges are roctx ranges, switched on at run time with VPF_HIP_ROCTX=1 (libvpfhip dlopen()s roctx; nothing is linked)
struct HipMark {
  int opened;
  explicit HipMark(const char* name) : opened(vpf_trace_push(name)) {}
  ~HipMark() { vpf_trace_pop(opened); }
  HipMark(const HipMark&) = delete;
  HipMark& operator=(const HipMark&) = delete;
};
constexpr auto TASK_EXEC_SUCCESS = TaskExecStatus::TASK_EXEC_SUCCESS;
constexpr auto TASK_EXEC_FAIL = TaskExecStatus::TASK_EXEC_FAIL;

std::atomic<int> g_extended{-1};

// Host copy of a large block that nobody reads soon (pinned staging <-> a caller's frame): non-temporal stores.  A plain memcpy of 2 MB
// pieces stays under glibc's non-temporal threshold, so every destination line is first READ (write-allocate) and then evicts something
// useful: 3 bytes of memory traffic per byte copied instead of 2.  VPF_HIP_NT_COPY=0 keeps memcpy (A/B: tools/download_bench.py).
#if defined(__x86_64__)
__attribute__((target("avx2"))) void copy_nt_avx2(uint8_t* d, const uint8_t* s, size_t n) {
  size_t head = (32 - ((uintptr_t)d & 31)) & 31;
  if (head > n) head = n;
  if (head) { std::memcpy(d, s, head); d += head; s += head; n -= head; }
  size_t i = 0;
  for (; i + 128 <= n; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i)), b = _mm256_loadu_si256((const __m256i*)(s + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(s + i + 64)), e = _mm256_loadu_si256((const __m256i*)(s + i + 96));
    _mm256_stream_si256((__m256i*)(d + i), a); _mm256_stream_si256((__m256i*)(d + i + 32), b);
    _mm256_stream_si256((__m256i*)(d + i + 64), c); _mm256_stream_si256((__m256i*)(d + i + 96), e);
  }
  _mm_sfence();
  if (i < n) std::memcpy(d + i, s + i, n - i);
}
#endif
void host_copy_large(void* dst, const void* src, size_t n) {
#if defined(__x86_64__)
  static const bool nt = [] {
    const char* e = std::getenv("VPF_HIP_NT_COPY");
    return !(e && e[0] == '0') && __builtin_cpu_supports("avx2");
  }();
  if (nt && n >= (256u << 10)) { copy_nt_avx2(static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), n); return; }
#endif
  std::memcpy(dst, src, n);
}

// The reference's resizer / remaper / down- and uploader tasks block until their stream has drained (their cuda_stream_sync callback:
// Tasks.cpp:1630-1640).  What they wait for here is mostly a kernel of 2 - 20 us, and hipStreamSynchronize puts the thread to sleep on an
// interrupt: ~15 - 20 us per call, several times the kernel (profiles/r03_sample_chain.txt: 36 us per frame as the reference's sample
// chain is written, 16 with the wait dropped).  Polling hipStreamQuery is no better (measured: slower).  What is cheap is a completion FLAG in
// page-locked host memory: the wait queues a 4-byte write of a sequence number behind the work (hipStreamWriteValue32) and spins on the
// flag in its own cache — no runtime call in the loop.  After kSpinUs without the value (a long queue, a big copy) it falls back to
// hipStreamSynchronize; a task whose flag could not be set up always takes that path.  VPF_HIP_SYNC_SPIN_US changes the budget (0: never spin).
struct StreamRef {  // argument of the stream-sync callback (always non-null, also for the NULL stream)
  HipContext ctx;
  HipStream str;
  volatile uint32_t* flag = nullptr;  // page-locked, mapped: written by the stream, read by the host
  void* flag_dev = nullptr;
  uint32_t seq = 0;
  bool flag_failed = false;
  StreamRef() = default;
  StreamRef(HipContext c, HipStream s) : ctx(c), str(s) {}
  StreamRef(const StreamRef& o) : ctx(o.ctx), str(o.str) {}  // the flag belongs to one owner: copies start without one
  StreamRef& operator=(const StreamRef& o) {
    if (this != &o) { release(); ctx = o.ctx; str = o.str; seq = 0; flag_failed = false; }
    return *this;
  }
  ~StreamRef() { release(); }
  void release() {
    if (flag) { (void)hipHostFree(const_cast<uint32_t*>(flag)); flag = nullptr; flag_dev = nullptr; }
  }
};
int64_t sync_spin_ns() {
  static const int64_t ns = [] {
    const char* e = std::getenv("VPF_HIP_SYNC_SPIN_US");
    return (int64_t)1000 * (e ? std::strtol(e, nullptr, 10) : 200);
  }();
  return ns;
}
bool flag_ready(StreamRef* s) {  // the completion flag of this task, set up on first use
  if (s->flag) return true;
  if (s->flag_failed || sync_spin_ns() <= 0) return false;
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&s->flag_dev, h, 0) == hipSuccess) {
    s->flag = static_cast<volatile uint32_t*>(h);
    *s->flag = 0;
    return true;
  }
  if (h) (void)hipHostFree(h);
  (void)hipGetLastError();
  s->flag_failed = true;
  return false;
}
// spin until the flag has reached `want` (sequence numbers only grow; wrap-around safe); false after the budget
bool flag_wait(StreamRef* s, uint32_t want, int64_t budget_ns) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0;; i++) {
    if ((int32_t)(*s->flag - want) >= 0) {
      std::atomic_thread_fence(std::memory_order_acquire);  // what the stream wrote before the flag (a staged download's bytes) is read after it
      return true;
    }
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
    if ((i & 255u) == 255u && std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > budget_ns) return false;
  }
}
void hip_stream_sync(void* p) {
  auto* s = static_cast<StreamRef*>(p);
  DeviceScope scope(s->ctx);
  const int64_t spin_ns = sync_spin_ns();
  if (spin_ns > 0 && !s->flag_failed) {
    (void)flag_ready(s);
    if (s->flag) {
      const uint32_t want = ++s->seq;
      if (hipStreamWriteValue32((hipStream_t)s->str, s->flag_dev, want, 0) == hipSuccess) {
        if (flag_wait(s, want, spin_ns)) return;
      } else {
        (void)hipGetLastError();
        s->flag_failed = true;
      }
    }
  }
  (void)hipStreamSynchronize((hipStream_t)s->str);
}

// Before device memory that queued kernels may still read is freed: the stream is drained; a stream the caller has destroyed meanwhile
// (the Python side owns it) cannot be asked — then the whole device is, and the failure is said out loud.
// This leans on the HIP runtime validating stream handles (a destroyed handle returns hipErrorContextIsDestroyed / InvalidHandle /
// InvalidValue instead of being dereferenced): if the address has meanwhile been RECYCLED for a new stream, the synchronize succeeds on that
// stream — benign (a wait on somebody else's work, and the allocation's own users were drained when their stream was destroyed:
// hipStreamDestroy completes queued work first), but then nothing is printed.  LzmHipSync::stream_state and persist_stream_idle rely on the
// same validation.
static void drain_before_free(StreamRef* s, const char* what) {
  DeviceScope scope(s->ctx);
  const hipError_t e = hipStreamSynchronize((hipStream_t)s->str);
  if (e == hipSuccess) return;
  (void)hipGetLastError();
  std::cerr << what << ": hipStreamSynchronize failed (" << hipGetErrorName(e) << "); waiting for the device instead" << std::endl;
  if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
}

// every HIP failure is reported on stderr (the reference prints its CUDA/NPP error codes the same way)
bool hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  std::cerr << what << " failed: " << hipGetErrorName(e) << " (" << hipGetErrorString(e) << ")" << std::endl;
  return false;
}

vpf_exec make_exec(HipContext ctx, HipStream str) {
  vpf_exec e;
  e.device = DeviceOfContext(ctx);
  e.flags = 0;
  e.stream = str;
  return e;
}
void fill_planes(Surface* s, vpf_plane out[3]) {
  std::memset(out, 0, 3 * sizeof(vpf_plane));
  for (uint32_t p = 0; p < s->NumPlanes() && p < 3; p++) {
    out[p].ptr = (void*)s->PlanePtr(p);
    out[p].pitch = s->Pitch(p);
  }
}

// How each reference impl turns the optional ColorspaceConversionContext into a matrix, and what it rejects.
enum CtxRule {
  RULE_NONE,         // context ignored (pure re-layout)
  RULE_NV12_RGB,     // nv12_rgb / nv12_bgr        TasksColorCvt.cpp:67-100,136-169
  RULE_RANGE_ONLY,   // nv12_yuv420                :218-232
  RULE_YUV420_RGB,   // yuv420_rgb / yuv420_bgr    :329-361,390-422
  RULE_601_BOTH,     // yuv444_bgr, bgr_yuv444     :452-483,634-665
  RULE_601_JPEG,     // yuv444_rgb (+ rgb planar)  :514-543,574-605
  RULE_RGB_YUV,      // rgb_yuv444, rgb_planar_yuv444, rgb_yuv420 :734-765,789-823,891-924
  RULE_FIXED_MPEG,   // bgr_ycbcr                  :686-712 (context ignored, YCbCr model)
};
struct PairInfo {
  Pixel_Format in, out;
  CtxRule rule;
  int level;  // 1 = pair exists in the reference ctor (:1313-1360), 2 = additive
  const char* name;
};
const PairInfo kPairs[] = {
    {NV12, YUV420, RULE_RANGE_ONLY, 1, "nv12_yuv420"},   {YUV420, NV12, RULE_NONE, 1, "yuv420_nv12"},
    {P10, NV12, RULE_NONE, 1, "p16_nv12"},               {P12, NV12, RULE_NONE, 1, "p16_nv12"},
    {NV12, RGB, RULE_NV12_RGB, 1, "nv12_rgb"},           {NV12, BGR, RULE_NV12_RGB, 1, "nv12_bgr"},
    {RGB, RGB_PLANAR, RULE_NONE, 1, "rgb8_deinterleave"}, {RGB_PLANAR, RGB, RULE_NONE, 1, "rgb8_interleave"},
    {RGB_PLANAR, YUV444, RULE_RGB_YUV, 1, "rgb_planar_yuv444"}, {Y, YUV444, RULE_NONE, 1, "y_yuv444"},
    {YUV420, RGB, RULE_YUV420_RGB, 1, "yuv420_rgb"},     {RGB, YUV420, RULE_RGB_YUV, 1, "rgb_yuv420"},
    {RGB, YUV444, RULE_RGB_YUV, 1, "rgb_yuv444"},        {BGR, YCBCR, RULE_FIXED_MPEG, 1, "bgr_ycbcr"},
    {RGB, BGR, RULE_NONE, 1, "rgb_bgr"},                 {BGR, RGB, RULE_NONE, 1, "bgr_rgb"},
    {YUV420, BGR, RULE_YUV420_RGB, 1, "yuv420_bgr"},     {YUV444, BGR, RULE_601_BOTH, 1, "yuv444_bgr"},
    {YUV444, RGB, RULE_601_JPEG, 1, "yuv444_rgb"},       {BGR, YUV444, RULE_601_BOTH, 1, "bgr_yuv444"},
    {NV12, Y, RULE_NONE, 1, "nv12_y"},                   {RGB, RGB_32F, RULE_NONE, 1, "rbg8_rgb32f"},
    {RGB, Y, RULE_NONE, 1, "rbg8_y"},                    {RGB_32F, RGB_32F_PLANAR, RULE_NONE, 1, "rgb32f_deinterleave"},
    // additive pairs (single-pass fusions and symmetric completions the kernels provide)
    {NV12, RGB_PLANAR, RULE_NV12_RGB, 2, "nv12_rgb_planar"},      // = nv12_rgb + rgb8_deinterleave in one pass
    {YUV420, RGB_PLANAR, RULE_YUV420_RGB, 2, "yuv420_rgb_planar"},
    {YUV444, RGB_PLANAR, RULE_601_JPEG, 2, "yuv444_rgb_planar"},  // defined but never dispatched in the reference (:555-615)
    {BGR, RGB_PLANAR, RULE_NONE, 2, "bgr8_deinterleave"},         {RGB_PLANAR, BGR, RULE_NONE, 2, "bgr8_interleave"},
    {BGR, Y, RULE_NONE, 2, "bgr8_y"},                             {RGB_PLANAR, Y, RULE_NONE, 2, "rgb8_planar_y"},
    {BGR, YUV420, RULE_RGB_YUV, 2, "bgr_yuv420"},                 {RGB_PLANAR, YUV420, RULE_RGB_YUV, 2, "rgb_planar_yuv420"},
};
const PairInfo* find_pair(Pixel_Format in, Pixel_Format out) {
  for (const auto& p : kPairs)
    if (p.in == in && p.out == out) return &p;
  return nullptr;
}

// -> true and (cs, cr) to hand to vpf_convert, or false after printing the reference's diagnostic
bool resolve_ctx(const PairInfo& pi, const ColorspaceConversionContext* c, int* cs, int* cr) {
  const bool ext = ExtendedColorspaces();
  auto unsupported_space = [&]() { std::cerr << pi.name << ": unsupported color space." << std::endl; return false; };
  auto unsupported_range = [&]() { std::cerr << pi.name << ": unsupported color range." << std::endl; return false; };
  switch (pi.rule) {
    case RULE_NONE: *cs = VPF_BT_601; *cr = VPF_MPEG; return true;
    case RULE_FIXED_MPEG: *cs = VPF_BT_601; *cr = VPF_MPEG; return true;
    case RULE_RANGE_ONLY: {
      const ColorRange r = c ? c->color_range : MPEG;
      if (r != JPEG && r != MPEG) return unsupported_range();
      *cs = VPF_BT_601; *cr = r; return true;
    }
    case RULE_NV12_RGB: {
      const ColorRange r = c ? c->color_range : MPEG;
      const ColorSpace s = c ? c->color_space : BT_601;
      if (s == BT_709) { *cs = VPF_BT_709; *cr = (r == JPEG) ? VPF_JPEG : VPF_MPEG; return true; }
      if (s == BT_601) {
        if (r == JPEG) { *cs = VPF_BT_601; *cr = VPF_JPEG; return true; }
        if (ext && r == MPEG) { *cs = VPF_BT_601; *cr = VPF_MPEG; return true; }
        std::cerr << "Rec. 601 NV12 -> RGB MPEG range conversion isn't supported yet." << std::endl
                  << "Convert NV12 -> YUV first and then do Rec. 601 YUV -> RGB MPEG range conversion." << std::endl;
        return false;
      }
      return unsupported_space();
    }
    case RULE_YUV420_RGB: {
      const ColorRange r = c ? c->color_range : MPEG;
      const ColorSpace s = c ? c->color_space : BT_601;
      if (s == BT_709) {
        if (ext) { *cs = VPF_BT_709; *cr = (r == JPEG) ? VPF_JPEG : VPF_MPEG; return true; }
        std::cerr << "Rec.709 YUV -> RGB conversion isn't supported yet." << std::endl;
        return false;
      }
      if (s != BT_601) return unsupported_space();
      *cs = VPF_BT_601; *cr = (r == JPEG) ? VPF_JPEG : VPF_MPEG; return true;
    }
    case RULE_601_BOTH: {
      const ColorRange r = c ? c->color_range : MPEG;
      const ColorSpace s = c ? c->color_space : BT_601;
      if (s != BT_601 && !(ext && s == BT_709 && pi.in == YUV444)) return unsupported_space();
      if (r != JPEG && r != MPEG) return unsupported_range();
      *cs = s; *cr = r; return true;
    }
    case RULE_601_JPEG: {
      const ColorRange r = c ? c->color_range : MPEG;
      const ColorSpace s = c ? c->color_space : BT_601;
      if (s != BT_601 && !(ext && s == BT_709)) return unsupported_space();
      if (r != JPEG && !(ext && r == MPEG)) return unsupported_range();
      *cs = s; *cr = r; return true;
    }
    case RULE_RGB_YUV: {
      const ColorRange r = c ? c->color_range : JPEG;
      const ColorSpace s = c ? c->color_space : BT_601;
      if (s != BT_601) return unsupported_space();
      if (r != JPEG && r != MPEG) return unsupported_range();
      *cs = VPF_BT_601; *cr = r; return true;
    }
  }
  return false;
}
}  // namespace

void SetExtendedColorspaces(bool on) { g_extended.store(on ? 1 : 0); }
bool ExtendedColorspaces() {
  int v = g_extended.load();
  if (v < 0) {
    const char* e = std::getenv("VPF_HIP_EXTENDED");
    v = (e && e[0] && e[0] != '0') ? 1 : 0;
    g_extended.store(v);
  }
  return v == 1;
}

// ------------------------------------------------------------------------------------------ ConvertSurface
struct ConvertSurface::Impl {
  const PairInfo* pair;
  uint32_t w, h;
  HipContext ctx;
  HipStream str;
  std::unique_ptr<Surface> out;  // allocated once, reused by every Execute (reference: nv12_rgb ctor :114-118)
  bool out_reused = false;
};
void ConvertSurface::SetOutputReused(bool reused) { pImpl->out_reused = reused; }
bool ConvertSurface::GetOutputReused() const { return pImpl->out_reused; }

int ConvertSurface::PairSupport(Pixel_Format in, Pixel_Format out) {
  const PairInfo* p = find_pair(in, out);
  return p ? p->level : 0;
}

bool ConvertSurface::ResolveContext(Pixel_Format in, Pixel_Format out, const ColorspaceConversionContext* c, int* cs, int* cr) {
  const PairInfo* p = find_pair(in, out);
  int a = 0, b = 0;
  const bool ok = p && resolve_ctx(*p, c, &a, &b);
  if (ok && cs) *cs = a;
  if (ok && cr) *cr = b;
  return ok;
}

ConvertSurface::ConvertSurface(uint32_t w, uint32_t h, Pixel_Format in, Pixel_Format out, HipContext ctx, HipStream str)
    : Task("HipConvertSurface", numInputs, numOutputs, nullptr, nullptr), pImpl() {
  const PairInfo* p = find_pair(in, out);
  if (!p) {
    std::stringstream ss;
    ss << "Unsupported pixel format conversion: " << in << " to " << out;
    throw std::invalid_argument(ss.str());
  }
  pImpl.reset(new Impl{p, w, h, ctx, str, nullptr, false});
  pImpl->out.reset(Surface::Make(out, w, h, ctx));
}
ConvertSurface::~ConvertSurface() {}
ConvertSurface* ConvertSurface::Make(uint32_t w, uint32_t h, Pixel_Format in, Pixel_Format out, HipContext ctx, HipStream str) {
  return new ConvertSurface(w, h, in, out, ctx, str);
}

TaskExecStatus ConvertSurface::Run() {
  const HipMark tick("ConvertSurface::Run");
  ClearOutputs();
  auto* in = static_cast<Surface*>(GetInput(0));
  const ColorspaceConversionContext* cc = nullptr;
  if (auto* b = static_cast<Buffer*>(GetInput(1))) cc = b->GetDataAs<ColorspaceConversionContext>();
  if (!in || in->Empty() || !pImpl->out || pImpl->out->Empty()) return TASK_EXEC_SUCCESS;  // null output = failure
  if (in->PixelFormat() != pImpl->pair->in || in->Width() != pImpl->w || in->Height() != pImpl->h) {
    std::cerr << pImpl->pair->name << ": input surface is " << PixelFormatName(in->PixelFormat()) << " " << in->Width()
              << "x" << in->Height() << ", converter was built for " << PixelFormatName(pImpl->pair->in) << " "
              << pImpl->w << "x" << pImpl->h << std::endl;
    return TASK_EXEC_SUCCESS;
  }
  int cs, cr;
  if (!resolve_ctx(*pImpl->pair, cc, &cs, &cr)) return TASK_EXEC_SUCCESS;
  vpf_plane src[3], dst[3];
  fill_planes(in, src);
  fill_planes(pImpl->out.get(), dst);
  vpf_exec ex = make_exec(pImpl->ctx, pImpl->str);
  if (pImpl->out_reused) ex.flags |= VPF_EXEC_DST_REUSED;
  const vpf_status st = vpf_convert(&ex, pImpl->pair->in, pImpl->pair->out, cs, cr, vpf_size{pImpl->w, pImpl->h}, src, dst);
  if (st != VPF_OK) {
    std::cerr << "Failed to convert surface. Error code: " << st << " (" << vpf_status_string(st) << ")" << std::endl;
    return TASK_EXEC_SUCCESS;
  }
  SetOutput(pImpl->out.get(), 0U);
  return TASK_EXEC_SUCCESS;
}

TaskExecStatus ConvertSurface::RunBatch(Surface* const* ins, Surface* const* outs, uint32_t n, const ColorspaceConversionContext* cc) {
  if (!ins || !outs || !n) return TASK_EXEC_FAIL;
  int cs, cr;
  if (!resolve_ctx(*pImpl->pair, cc, &cs, &cr)) return TASK_EXEC_FAIL;
  std::vector<vpf_frame_io> io(n);
  for (uint32_t i = 0; i < n; i++) {
    Surface *s = ins[i], *d = outs[i];
    if (!s || !d || s->Empty() || d->Empty() || s->PixelFormat() != pImpl->pair->in || d->PixelFormat() != pImpl->pair->out ||
        s->Width() != pImpl->w || s->Height() != pImpl->h || d->Width() != pImpl->w || d->Height() != pImpl->h)
      return TASK_EXEC_FAIL;
    fill_planes(s, io[i].src);
    fill_planes(d, io[i].dst);
  }
  const vpf_exec ex = make_exec(pImpl->ctx, pImpl->str);
  const vpf_status st = vpf_convert_batch(&ex, pImpl->pair->in, pImpl->pair->out, cs, cr, vpf_size{pImpl->w, pImpl->h}, n, io.data());
  if (st != VPF_OK) {
    std::cerr << "Failed to convert surfaces. Error code: " << st << " (" << vpf_status_string(st) << ")" << std::endl;
    return TASK_EXEC_FAIL;
  }
  return TASK_EXEC_SUCCESS;
}

// ------------------------------------------------------------------------------------------ ConvertResizeSurface
struct ConvertResizeSurface::Impl {
  const PairInfo* pair;
  uint32_t sw, sh, dw, dh;
  HipContext ctx;
  HipStream str;
  std::unique_ptr<Surface> out;
};
ConvertResizeSurface::ConvertResizeSurface(uint32_t sw, uint32_t sh, Pixel_Format in, uint32_t dw, uint32_t dh, Pixel_Format out,
                                           HipContext ctx, HipStream str)
    : Task("HipConvertResizeSurface", numInputs, numOutputs, nullptr, nullptr), pImpl() {
  const PairInfo* p = find_pair(in, out);
  const bool fusable = p && (in == NV12 || in == YUV420) && (out == RGB || out == BGR || out == RGB_PLANAR);
  if (!fusable || !sw || !sh || !dw || !dh) {
    std::stringstream ss;
    ss << "Unsupported fused conversion + resize: " << in << " to " << out;
    throw std::invalid_argument(ss.str());
  }
  pImpl.reset(new Impl{p, sw, sh, dw, dh, ctx, str, nullptr});
  pImpl->out.reset(Surface::Make(out, dw, dh, ctx));
}
ConvertResizeSurface::~ConvertResizeSurface() {}
ConvertResizeSurface* ConvertResizeSurface::Make(uint32_t sw, uint32_t sh, Pixel_Format in, uint32_t dw, uint32_t dh, Pixel_Format out,
                                                 HipContext ctx, HipStream str) {
  return new ConvertResizeSurface(sw, sh, in, dw, dh, out, ctx, str);
}
TaskExecStatus ConvertResizeSurface::Run() {
  const HipMark tick("ConvertResizeSurface::Run");
  ClearOutputs();
  auto* in = static_cast<Surface*>(GetInput(0));
  const ColorspaceConversionContext* cc = nullptr;
  if (auto* b = static_cast<Buffer*>(GetInput(1))) cc = b->GetDataAs<ColorspaceConversionContext>();
  if (!in || in->Empty() || !pImpl->out || pImpl->out->Empty()) return TASK_EXEC_SUCCESS;
  if (in->PixelFormat() != pImpl->pair->in || in->Width() != pImpl->sw || in->Height() != pImpl->sh) {
    std::cerr << "fused " << pImpl->pair->name << ": input surface is " << PixelFormatName(in->PixelFormat()) << " " << in->Width()
              << "x" << in->Height() << ", task was built for " << PixelFormatName(pImpl->pair->in) << " " << pImpl->sw << "x"
              << pImpl->sh << std::endl;
    return TASK_EXEC_SUCCESS;
  }
  int cs, cr;
  if (!resolve_ctx(*pImpl->pair, cc, &cs, &cr)) return TASK_EXEC_SUCCESS;
  vpf_plane src[3], dst[3];
  fill_planes(in, src);
  fill_planes(pImpl->out.get(), dst);
  const vpf_exec ex = make_exec(pImpl->ctx, pImpl->str);
  const vpf_status st = vpf_convert_resize(&ex, pImpl->pair->in, pImpl->pair->out, cs, cr, vpf_size{pImpl->sw, pImpl->sh}, src,
                                           vpf_size{pImpl->dw, pImpl->dh}, dst);
  if (st != VPF_OK) {
    std::cerr << "Failed to convert + resize surface. Error code: " << st << " (" << vpf_status_string(st) << ")" << std::endl;
    return TASK_EXEC_SUCCESS;
  }
  SetOutput(pImpl->out.get(), 0U);
  return TASK_EXEC_SUCCESS;
}
TaskExecStatus ConvertResizeSurface::RunBatch(Surface* const* ins, Surface* const* outs, uint32_t n, const ColorspaceConversionContext* cc) {
  if (!ins || !outs || !n) return TASK_EXEC_FAIL;
  int cs, cr;
  if (!resolve_ctx(*pImpl->pair, cc, &cs, &cr)) return TASK_EXEC_FAIL;
  std::vector<vpf_frame_io> io(n);
  for (uint32_t i = 0; i < n; i++) {
    Surface *s = ins[i], *d = outs[i];
    if (!s || !d || s->Empty() || d->Empty() || s->PixelFormat() != pImpl->pair->in || d->PixelFormat() != pImpl->pair->out ||
        s->Width() != pImpl->sw || s->Height() != pImpl->sh || d->Width() != pImpl->dw || d->Height() != pImpl->dh)
      return TASK_EXEC_FAIL;
    fill_planes(s, io[i].src);
    fill_planes(d, io[i].dst);
  }
  const vpf_exec ex = make_exec(pImpl->ctx, pImpl->str);
  const vpf_status st = vpf_convert_resize_batch(&ex, pImpl->pair->in, pImpl->pair->out, cs, cr, vpf_size{pImpl->sw, pImpl->sh},
                                                 vpf_size{pImpl->dw, pImpl->dh}, n, io.data());
  if (st != VPF_OK) {
    std::cerr << "Failed to convert + resize surfaces. Error code: " << st << " (" << vpf_status_string(st) << ")" << std::endl;
    return TASK_EXEC_FAIL;
  }
  return TASK_EXEC_SUCCESS;
}

// ------------------------------------------------------------------------------------------ ResizeSurface
struct ResizeSurface::Impl {
  Pixel_Format fmt;
  uint32_t w, h;
  StreamRef sref;
  std::unique_ptr<Surface> out;
  int interp = VPF_INTERP_LANCZOS3;  // what the reference's resizer asks NPP for (NPPI_INTER_LANCZOS, Tasks.cpp:1190,1248,1373,1431)
  bool async = false;
  // The filter's per-shape operand tables live in a workspace this task owns — like its destination surface (the reference's impls own
  // theirs: Tasks.cpp:1134-1150), sized by vpf_resize_workspace_bytes for the source size last seen, used on the task's one stream.
  std::unique_ptr<CudaBuffer> ws_mem;
  vpf_workspace ws{};
  uint32_t ws_sw = 0, ws_sh = 0;
  int ws_interp = -1;
  vpf_workspace* workspace(uint32_t sw, uint32_t sh) {
    if (ws_interp == interp && ws_sw == sw && ws_sh == sh) return ws.ptr ? &ws : nullptr;
    ws_interp = interp; ws_sw = sw; ws_sh = sh;
    const uint64_t need = vpf_resize_workspace_bytes(fmt, interp, vpf_size{sw, sh}, vpf_size{w, h});
    if (need > ws.bytes || !need) {
      if (ws_mem) drain_before_free(&sref, "ResizeSurface workspace");  // kernels queued with the old region's tables must have finished before it is freed (a change of input size: rare)
      ws_mem.reset(need ? CudaBuffer::Make(1, (size_t)need, sref.ctx) : nullptr);
      ws = vpf_workspace{};
      if (ws_mem && ws_mem->GpuMem()) { ws.ptr = (void*)ws_mem->GpuMem(); ws.bytes = need; }
    }
    return ws.ptr ? &ws : nullptr;  // (no memory for it: the library's own arena serves the call)
  }
};
void ResizeSurface::SetAsync(bool on) { pImpl->async = on; }
bool ResizeSurface::GetAsync() const { return pImpl->async; }
void ResizeSurface::SetInterpolation(int interp) {
  if (interp < VPF_INTERP_NEAREST || interp > VPF_INTERP_LANCZOS3) throw std::invalid_argument("ResizeSurface: unknown interpolation");
  pImpl->interp = interp;
}
int ResizeSurface::GetInterpolation() const { return pImpl->interp; }
static bool resize_format_ok(Pixel_Format f) {
  switch (f) {  // reference: packed 3C, planar (YUV420/YCBCR/YUV444/RGB_PLANAR), RGB_32F, RGB_32F_PLANAR, NV12; + Y (Tasks.cpp:1458-1476)
    case RGB: case BGR: case YUV420: case YCBCR: case YUV444: case RGB_PLANAR: case NV12: case Y: case RGB_32F: case RGB_32F_PLANAR: return true;
    default: return false;
  }
}
ResizeSurface::ResizeSurface(uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str)
    : Task("HipResizeSurface", numInputs, numOutputs, hip_stream_sync, nullptr), pImpl() {
  if (!resize_format_ok(f)) {
    std::stringstream ss;
    ss << "pixel format not supported";
    throw std::runtime_error(ss.str());
  }
  // Default = the reference's filter (round 3; rounds 1-2 defaulted to bilinear, which SetInterpolation(1) / VPF_HIP_RESIZE_INTERP=bilinear
  // still select per instance / per process: it is the cheaper filter, and the one the fused PySurfaceConvertResizer implements)
  pImpl.reset(new Impl{f, w, h, StreamRef{ctx, str}, nullptr, VPF_INTERP_LANCZOS3});
  if (const char* e = std::getenv("VPF_HIP_RESIZE_INTERP")) {
    if (!std::strcmp(e, "lanczos") || !std::strcmp(e, "2")) pImpl->interp = VPF_INTERP_LANCZOS3;
    else if (!std::strcmp(e, "bilinear") || !std::strcmp(e, "linear") || !std::strcmp(e, "1")) pImpl->interp = VPF_INTERP_LINEAR;
    else if (!std::strcmp(e, "nearest") || !std::strcmp(e, "0")) pImpl->interp = VPF_INTERP_NEAREST;
  }
  pImpl->out.reset(Surface::Make(f, w, h, ctx));
}
ResizeSurface::~ResizeSurface() {
  if (pImpl && pImpl->ws_mem) drain_before_free(&pImpl->sref, "~ResizeSurface");  // queued resizes still read the workspace's tables
}
ResizeSurface* ResizeSurface::Make(uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str) {
  return new ResizeSurface(w, h, f, ctx, str);
}
TaskExecStatus ResizeSurface::Run() {
  const HipMark tick("ResizeSurface::Run");
  ClearOutputs();
  auto* in = static_cast<Surface*>(GetInput(0));
  if (!in || in->Empty() || !pImpl->out || pImpl->out->Empty()) return TASK_EXEC_FAIL;
  if (in->PixelFormat() != pImpl->fmt) return TASK_EXEC_FAIL;  // Tasks.cpp:1166-1168
  vpf_plane src[3], dst[3];
  fill_planes(in, src);
  fill_planes(pImpl->out.get(), dst);
  const vpf_exec ex = make_exec(pImpl->sref.ctx, pImpl->sref.str);
  const vpf_status st = vpf_resize_ws(&ex, pImpl->fmt, pImpl->interp, vpf_size{in->Width(), in->Height()}, src,
                                      vpf_size{pImpl->w, pImpl->h}, dst, pImpl->workspace(in->Width(), in->Height()));
  if (!pImpl->async) hip_stream_sync(&pImpl->sref);  // the reference task is blocking (cuda_stream_sync callback); SetAsync(true) opts out
  if (st != VPF_OK) {
    std::cerr << "Failed to resize surface. Error code: " << st << " (" << vpf_status_string(st) << ")" << std::endl;
    return TASK_EXEC_FAIL;
  }
  SetOutput(pImpl->out.get(), 0U);
  return TASK_EXEC_SUCCESS;
}

// additive: n same-shape surfaces into n caller-owned surfaces of the task's size, every plane of every frame in as few dispatches
// as possible (vpf_resize_batch); asynchronous on the task's stream like ConvertSurface::RunBatch
TaskExecStatus ResizeSurface::RunBatch(Surface* const* ins, Surface* const* outs, uint32_t n) {
  const HipMark tick("ResizeSurface::RunBatch");
  if (!ins || !outs || !n || !ins[0]) return TASK_EXEC_FAIL;
  const uint32_t sw = ins[0]->Width(), sh = ins[0]->Height();
  std::vector<vpf_frame_io> io(n);
  for (uint32_t i = 0; i < n; i++) {
    Surface *s = ins[i], *d = outs[i];
    if (!s || !d || s->Empty() || d->Empty() || s->PixelFormat() != pImpl->fmt || d->PixelFormat() != pImpl->fmt || s->Width() != sw || s->Height() != sh ||
        d->Width() != pImpl->w || d->Height() != pImpl->h)
      return TASK_EXEC_FAIL;
    fill_planes(s, io[i].src);
    fill_planes(d, io[i].dst);
  }
  const vpf_exec ex = make_exec(pImpl->sref.ctx, pImpl->sref.str);
  const vpf_status st = vpf_resize_batch_ws(&ex, pImpl->fmt, pImpl->interp, vpf_size{sw, sh}, vpf_size{pImpl->w, pImpl->h}, n, io.data(), pImpl->workspace(sw, sh));
  if (st != VPF_OK) {
    std::cerr << "Failed to resize surfaces. Error code: " << st << " (" << vpf_status_string(st) << ")" << std::endl;
    return TASK_EXEC_FAIL;
  }
  return TASK_EXEC_SUCCESS;
}

// ------------------------------------------------------------------------------------------ RemapSurface
struct RemapSurface::Impl {
  Pixel_Format fmt;
  uint32_t w, h;
  StreamRef sref;
  std::unique_ptr<CudaBuffer> xmap, ymap;
  std::unique_ptr<Surface> out;
  bool async = false;
};
void RemapSurface::SetAsync(bool on) { pImpl->async = on; }
bool RemapSurface::GetAsync() const { return pImpl->async; }
RemapSurface::RemapSurface(const float* x_map, const float* y_map, uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str)
    : Task("HipRemapSurface", numInputs, numOutputs, hip_stream_sync, nullptr), pImpl() {
  if (f != RGB && f != BGR) throw std::runtime_error("pixel format not supported");  // Tasks.cpp:1615-1620
  if (!x_map || !y_map || !w || !h) throw std::runtime_error("RemapSurface: empty map");
  pImpl.reset(new Impl{f, w, h, StreamRef{ctx, str}, nullptr, nullptr, nullptr});
  // maps go to the device once, synchronously, as two tight float[h*w] buffers (Tasks.cpp:1523-1526)
  pImpl->xmap.reset(CudaBuffer::Make(x_map, sizeof(float), (size_t)w * h, ctx, str));
  pImpl->ymap.reset(CudaBuffer::Make(y_map, sizeof(float), (size_t)w * h, ctx, str));
  pImpl->out.reset(Surface::Make(f, w, h, ctx));
  // destination pixels whose source falls outside the image are left untouched: start from black, not garbage
  // (on the task's own stream and waited for: a null-stream hipMemset may still be in flight when the first Run()
  //  launches on a non-blocking user stream, and would then black out pixels the kernel has just written)
  DeviceScope scope(ctx);
  if (!hip_ok(hipMemsetAsync((void*)pImpl->out->PlanePtr(0), 0, (size_t)pImpl->out->Pitch(0) * h, (hipStream_t)str), "RemapSurface: hipMemsetAsync") ||
      !hip_ok(hipStreamSynchronize((hipStream_t)str), "RemapSurface: hipStreamSynchronize")) {
    throw std::runtime_error("RemapSurface: can't clear the output surface");
  }
}
RemapSurface::~RemapSurface() {}
RemapSurface* RemapSurface::Make(const float* x, const float* y, uint32_t w, uint32_t h, Pixel_Format f, HipContext ctx, HipStream str) {
  return new RemapSurface(x, y, w, h, f, ctx, str);
}
TaskExecStatus RemapSurface::Run() {
  const HipMark tick("RemapSurface::Run");
  ClearOutputs();
  auto* in = static_cast<Surface*>(GetInput(0));
  if (!in || in->Empty() || in->PixelFormat() != pImpl->fmt) return TASK_EXEC_FAIL;
  vpf_plane src[3], dst[3];
  fill_planes(in, src);
  fill_planes(pImpl->out.get(), dst);
  const vpf_exec ex = make_exec(pImpl->sref.ctx, pImpl->sref.str);
  const vpf_status st = vpf_remap(&ex, pImpl->fmt, vpf_size{in->Width(), in->Height()}, src,
                                  (const float*)pImpl->xmap->GpuMem(), pImpl->w * 4, (const float*)pImpl->ymap->GpuMem(),
                                  pImpl->w * 4, vpf_size{pImpl->w, pImpl->h}, dst);
  if (!pImpl->async) hip_stream_sync(&pImpl->sref);  // blocking like the reference (Tasks.cpp:1630-1640); SetAsync(true) opts out
  if (st != VPF_OK) {
    std::cerr << "Failed to remap surface. Error code: " << st << " (" << vpf_status_string(st) << ")" << std::endl;
    return TASK_EXEC_FAIL;
  }
  SetOutput(pImpl->out.get(), 0U);
  return TASK_EXEC_SUCCESS;
}

// additive: the task's maps applied to n same-shape surfaces into n caller-owned surfaces of the map's size in one dispatch per 32
// frames (vpf_remap_batch).  Destination pixels whose source falls outside the picture keep what the caller's surface held.
TaskExecStatus RemapSurface::RunBatch(Surface* const* ins, Surface* const* outs, uint32_t n) {
  const HipMark tick("RemapSurface::RunBatch");
  if (!ins || !outs || !n || !ins[0]) return TASK_EXEC_FAIL;
  const uint32_t sw = ins[0]->Width(), sh = ins[0]->Height();
  std::vector<vpf_frame_io> io(n);
  for (uint32_t i = 0; i < n; i++) {
    Surface *s = ins[i], *d = outs[i];
    if (!s || !d || s->Empty() || d->Empty() || s->PixelFormat() != pImpl->fmt || d->PixelFormat() != pImpl->fmt || s->Width() != sw || s->Height() != sh ||
        d->Width() != pImpl->w || d->Height() != pImpl->h)
      return TASK_EXEC_FAIL;
    fill_planes(s, io[i].src);
    fill_planes(d, io[i].dst);
  }
  const vpf_exec ex = make_exec(pImpl->sref.ctx, pImpl->sref.str);
  const vpf_status st = vpf_remap_batch(&ex, pImpl->fmt, vpf_size{sw, sh}, (const float*)pImpl->xmap->GpuMem(), pImpl->w * 4, (const float*)pImpl->ymap->GpuMem(),
                                        pImpl->w * 4, vpf_size{pImpl->w, pImpl->h}, n, io.data());
  if (st != VPF_OK) {
    std::cerr << "Failed to remap surfaces. Error code: " << st << " (" << vpf_status_string(st) << ")" << std::endl;
    return TASK_EXEC_FAIL;
  }
  return TASK_EXEC_SUCCESS;
}

// ------------------------------------------------------------------------------------------ HostPinCache (Tasks.hpp)
namespace {
struct PinEnt {
  uintptr_t p; size_t n; uint64_t owner; uint64_t tick; int dev; bool registered, failed;
};
struct PinState {
  std::mutex mu;
  std::vector<PinEnt> e;
  uint64_t tick = 0;
  HostPinCache::Stats st{};
  size_t cap_bytes = [] {
    const char* v = std::getenv("VPF_HIP_PIN_CACHE_MB");
    return (size_t)(v ? std::strtoull(v, nullptr, 10) : 0ull) << 20;  // opt-in (Tasks.hpp)
  }();
};
PinState& pins() { static PinState* s = new PinState; return *s; }  // (leaked on purpose: weak-reference callbacks may run during interpreter shutdown)
constexpr size_t kPinMaxEntries = 64, kPinMinBytes = 256u << 10;
// a DMA that an asynchronous upload queued from this range may still be running: the device it was used on is drained first
void pin_release(PinState& S, PinEnt& x) {
  if (!x.registered) return;
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess) {
    if (x.dev >= 0 && x.dev != cur) (void)hipSetDevice(x.dev);
    (void)hipDeviceSynchronize();
    if (x.dev >= 0 && x.dev != cur) (void)hipSetDevice(cur);
  }
  if (hipHostUnregister(reinterpret_cast<void*>(x.p)) != hipSuccess) (void)hipGetLastError();
  x.registered = false;
  S.st.registered--; S.st.bytes -= x.n; S.st.evictions++;
}
}  // namespace
HostPinCache::Use HostPinCache::note_use(const void* ptr, size_t bytes, uint64_t owner, int device) {
  PinState& S = pins();
  if (!S.cap_bytes || !ptr || bytes < kPinMinBytes || bytes > S.cap_bytes) return kStaged;
  const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
  std::lock_guard<std::mutex> g(S.mu);
  S.tick++;
  bool owner_known = false;
  PinEnt* hit = nullptr;
  for (size_t i = 0; i < S.e.size();) {
    PinEnt& x = S.e[i];
    owner_known = owner_known || x.owner == owner;
    if (x.p == p && x.n == bytes && x.owner == owner) { hit = &x; i++; continue; }
    if (x.p < p + bytes && p < x.p + x.n) {  // overlaps under another owner / other bounds: whatever was there is gone
      pin_release(S, x);
      S.e.erase(S.e.begin() + (ptrdiff_t)i);
      hit = nullptr;  // (erase moved the elements: look the hit up again below)
      for (PinEnt& y : S.e) if (y.p == p && y.n == bytes && y.owner == owner) hit = &y;
      continue;
    }
    i++;
  }
  if (hit) {
    hit->tick = S.tick; hit->dev = device;
    if (hit->registered) { S.st.hits++; return kInPlace; }
    if (hit->failed) { S.st.staged++; return kStaged; }
    // second sight: register, making room first (least recently used registered entries leave)
    while (S.st.bytes + bytes > S.cap_bytes) {
      PinEnt* lru = nullptr;
      for (PinEnt& x : S.e) if (x.registered && (!lru || x.tick < lru->tick)) lru = &x;
      if (!lru) break;
      pin_release(S, *lru);
    }
    if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterPortable) == hipSuccess) {
      hit->registered = true;
      S.st.registered++; S.st.bytes += bytes; S.st.hits++;
      return kInPlace;
    }
    (void)hipGetLastError();
    hit->failed = true; S.st.failures++; S.st.staged++;
    return kStaged;
  }
  if (S.e.size() >= kPinMaxEntries) {  // the least recently used entry leaves
    size_t lru = 0;
    for (size_t i = 1; i < S.e.size(); i++) if (S.e[i].tick < S.e[lru].tick) lru = i;
    pin_release(S, S.e[lru]);
    S.e.erase(S.e.begin() + (ptrdiff_t)lru);
  }
  S.e.push_back(PinEnt{p, bytes, owner, S.tick, device, false, false});
  S.st.staged++;
  return owner_known ? kStaged : kFirstSight;
}
void HostPinCache::owner_gone(uint64_t owner) {
  PinState& S = pins();
  std::lock_guard<std::mutex> g(S.mu);
  for (size_t i = 0; i < S.e.size();) {
    if (S.e[i].owner == owner) { pin_release(S, S.e[i]); S.e.erase(S.e.begin() + (ptrdiff_t)i); } else i++;
  }
}
bool HostPinCache::covers(const void* ptr) {
  PinState& S = pins();
  const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
  std::lock_guard<std::mutex> g(S.mu);
  for (const PinEnt& x : S.e)
    if (x.registered && p >= x.p && p < x.p + x.n) return true;
  return false;
}
void HostPinCache::set_budget_mb(size_t mb) {
  if (!mb) drop_all();
  PinState& S = pins();
  std::lock_guard<std::mutex> g(S.mu);
  S.cap_bytes = mb << 20;
}
void HostPinCache::drop_all() {
  PinState& S = pins();
  std::lock_guard<std::mutex> g(S.mu);
  for (PinEnt& x : S.e) pin_release(S, x);
  S.e.clear();
}
HostPinCache::Stats HostPinCache::stats() {
  PinState& S = pins();
  std::lock_guard<std::mutex> g(S.mu);
  Stats t = S.st;
  t.budget = S.cap_bytes;
  return t;
}

// ------------------------------------------------------------------------------------------ CudaUploadFrame
// MI355X-first: the host frame is staged through PINNED memory and copied on a dedicated copy stream; the
// task's stream only waits on the copy's event.  Two slots (pinned buffer + device surface each) let the upload of
// frame i+1 overlap the kernels still converting frame i.  The reference copies straight from the (pageable)
// numpy buffer on the task stream and blocks (src/TC/src/Tasks.cpp:625-662).
struct CudaUploadFrame::Impl {
  static constexpr int kSlots = 4;  // staging buffers / device surfaces in rotation: the host copy of frame i + 1 overlaps the DMA of frame i
  StreamRef sref;
  bool async_in_place = false;  // SetAsyncInPlace(true): asynchronous uploads may read frames HostPinCache page-locked where they lie
  bool async = false;  // SetAsync(true) / VPF_HIP_UPLOAD_ASYNC=1: Run() returns once the copy is QUEUED (default: it waits for the copy, like the reference's task, Tasks.cpp:617-618)
  Pixel_Format fmt;
  hipStream_t copy_stream = nullptr;
  hipEvent_t done[kSlots] = {};
  hipEvent_t consumed[kSlots] = {};  // task-stream work submitted before the slot's surface is rewritten
  std::unique_ptr<Buffer> staging[kSlots];
  std::unique_ptr<Surface> surf[kSlots];
  uint64_t n = 0;
  ~Impl() {  // owns the HIP objects itself, so a constructor that throws half-way releases what it had created
    DeviceScope scope(sref.ctx);
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
    for (auto& e : done)
      if (e) (void)hipEventDestroy(e);
    for (auto& e : consumed)
      if (e) (void)hipEventDestroy(e);
  }
};
CudaUploadFrame::CudaUploadFrame(HipStream str, HipContext ctx, uint32_t w, uint32_t h, Pixel_Format f)
    : Task("HipUploadFrame", numInputs, numOutputs, hip_stream_sync, nullptr), pImpl(new Impl) {
  pImpl->sref = StreamRef{ctx, str};
  pImpl->fmt = f;
  if (const char* e = std::getenv("VPF_HIP_UPLOAD_ASYNC")) pImpl->async = e[0] && e[0] != '0';
  DeviceScope scope(ctx);
  for (int i = 0; i < Impl::kSlots; i++) {
    pImpl->surf[i].reset(Surface::Make(f, w, h, ctx));
    if (!pImpl->surf[i]) throw std::invalid_argument("CudaUploadFrame: unsupported pixel format");
    pImpl->staging[i].reset(Buffer::MakeOwnMem(pImpl->surf[i]->HostMemSize(), ctx ? ctx : (HipContext)-1));
    if (hipEventCreateWithFlags(&pImpl->done[i], hipEventDisableTiming) != hipSuccess) pImpl->done[i] = nullptr;
    if (hipEventCreateWithFlags(&pImpl->consumed[i], hipEventDisableTiming) != hipSuccess) pImpl->consumed[i] = nullptr;
  }
  if (hipStreamCreateWithFlags(&pImpl->copy_stream, hipStreamNonBlocking) != hipSuccess) pImpl->copy_stream = nullptr;
}
CudaUploadFrame::~CudaUploadFrame() {}
void CudaUploadFrame::SetAsync(bool on) { pImpl->async = on; }
bool CudaUploadFrame::GetAsync() const { return pImpl->async; }
void CudaUploadFrame::SetAsyncInPlace(bool on) { pImpl->async_in_place = on; }
bool CudaUploadFrame::GetAsyncInPlace() const { return pImpl->async_in_place; }
CudaUploadFrame* CudaUploadFrame::Make(HipStream str, HipContext ctx, uint32_t w, uint32_t h, Pixel_Format f) {
  return new CudaUploadFrame(str, ctx, w, h, f);
}
TaskExecStatus CudaUploadFrame::Run() {
  const HipMark tick("CudaUploadFrame::Run");
  auto* host = static_cast<Buffer*>(GetInput(0));
  if (!host) return TASK_EXEC_FAIL;
  ClearOutputs();
  const int slot = (int)(pImpl->n++ % Impl::kSlots);
  Surface* s = pImpl->surf[slot].get();
  Buffer* stage = pImpl->staging[slot].get();
  if (!s || s->Empty() || host->GetRawMemSize() < s->HostMemSize()) return TASK_EXEC_FAIL;
  DeviceScope scope(pImpl->sref.ctx);
  hipStream_t cs = pImpl->copy_stream ? pImpl->copy_stream : (hipStream_t)pImpl->sref.str;
  // the slot's previous copy must have drained before its pinned buffer is overwritten
  if (pImpl->done[slot] && !hip_ok(hipEventSynchronize(pImpl->done[slot]), "CudaUploadFrame: hipEventSynchronize")) return TASK_EXEC_FAIL;
  // a frame that already lives in pinned (hipHostMalloc / registered) memory is DMA'd from where it is; pageable
  // memory is staged through the slot's pinned buffer (one host memcpy, then a true async copy)
  const uint8_t* src = host->GetDataAs<uint8_t>();
  hipPointerAttribute_t attr;
  bool pinned_src = (hipPointerGetAttributes(&attr, src) == hipSuccess) && attr.type == hipMemoryTypeHost;
  // an asynchronous upload lets the caller reuse an ordinary frame at once: a frame that is page-locked only because HostPinCache registered it
  // (by a blocking uploader, earlier) is still copied out first — unless the caller has promised to leave it alone (SetAsyncInPlace)
  if (pinned_src && pImpl->async && !pImpl->async_in_place && HostPinCache::covers(src)) pinned_src = false;
  if (!pinned_src) {
    (void)hipGetLastError();  // a pageable pointer makes hipPointerGetAttributes fail: not an error for us
    host_copy_large(stage->GetRawMemPtr(), host->GetRawMemPtr(), s->HostMemSize());
    src = stage->GetDataAs<uint8_t>();
  }
  // The surface of this slot was handed out two uploads ago; kernels that read it (converters on the task's stream)
  // may still be queued.  Whatever has been submitted to the task stream so far must finish before the DMA rewrites it.
  if (pImpl->consumed[slot] && cs != (hipStream_t)pImpl->sref.str) {
    if (!hip_ok(hipEventRecord(pImpl->consumed[slot], (hipStream_t)pImpl->sref.str), "CudaUploadFrame: hipEventRecord") ||
        !hip_ok(hipStreamWaitEvent(cs, pImpl->consumed[slot], 0), "CudaUploadFrame: hipStreamWaitEvent"))
      return TASK_EXEC_FAIL;
  }
  for (uint32_t p = 0; p < s->NumPlanes(); p++) {  // planes concatenated at tight width (Tasks.cpp:643-658)
    const size_t wb = s->WidthInBytes(p), rows = s->Height(p);
    if (!hip_ok(hipMemcpy2DAsync((void*)s->PlanePtr(p), s->Pitch(p), src, wb, wb, rows, hipMemcpyHostToDevice, cs), "CudaUploadFrame: hipMemcpy2DAsync"))
      return TASK_EXEC_FAIL;
    src += wb * rows;
  }
  if (pImpl->done[slot] && cs != (hipStream_t)pImpl->sref.str) {
    // order the task stream behind the copy, then — by default — wait for the copy itself, as the reference's task does (Tasks.cpp:617-618):
    // the surface this call returns is complete for EVERY consumer, whichever stream it reads on (a converter built on another stream, a
    // torch / DLPack view through PlanePtr().GpuMem(), SurfacePlane.Export).  Only the copy is waited for: kernels queued on the task stream
    // keep running underneath.  SetAsync(true) (or VPF_HIP_UPLOAD_ASYNC=1) returns as soon as the copy is queued: the surface is then valid
    // in stream order on the task's stream only, a page-locked source frame must not be rewritten before the caller's next
    // synchronisation, and the host goes on to decode / stage the next frame while this one crosses PCIe (2 200 -> 3 000 frames/s at 4K
    // from pageable memory, profiles/r03_pipeline_async_upload.txt).  Round 3 made that the default for staged (pageable) frames: a silent
    // change of the reference's contract, taken back in round 4.
    if (!hip_ok(hipEventRecord(pImpl->done[slot], cs), "CudaUploadFrame: hipEventRecord")) return TASK_EXEC_FAIL;
    if (!hip_ok(hipStreamWaitEvent((hipStream_t)pImpl->sref.str, pImpl->done[slot], 0), "CudaUploadFrame: hipStreamWaitEvent")) return TASK_EXEC_FAIL;
    if (!pImpl->async) {
      if (!hip_ok(hipEventSynchronize(pImpl->done[slot]), "CudaUploadFrame: hipEventSynchronize")) return TASK_EXEC_FAIL;
    }
  } else {
    hip_stream_sync(&pImpl->sref);
  }
  SetOutput(s, 0U);
  return TASK_EXEC_SUCCESS;
}

// ------------------------------------------------------------------------------------------ CudaDownloadSurface
struct CudaDownloadSurface::Impl {
  StreamRef sref;
  Pixel_Format fmt;
  uint32_t w, h;
  std::unique_ptr<Buffer> host;  // pinned
};
CudaDownloadSurface::CudaDownloadSurface(HipStream str, HipContext ctx, uint32_t w, uint32_t h, Pixel_Format f)
    : Task("HipDownloadSurface", numInputs, numOutputs, hip_stream_sync, nullptr), pImpl() {
  if (!Surface::Supported(f)) {
    std::stringstream ss;
    ss << "CudaDownloadSurface: unsupported pixel format: " << f;
    throw std::invalid_argument(ss.str());  // Tasks.cpp:759-762
  }
  pImpl.reset(new Impl{StreamRef{ctx, str}, f, w, h, nullptr});
  pImpl->host.reset(Buffer::MakeOwnMem(Surface::HostMemSizeOf(f, w, h), ctx ? ctx : (HipContext)-1));
}
CudaDownloadSurface::~CudaDownloadSurface() {}
CudaDownloadSurface* CudaDownloadSurface::Make(HipStream str, HipContext ctx, uint32_t w, uint32_t h, Pixel_Format f) {
  return new CudaDownloadSurface(str, ctx, w, h, f);
}
static bool download_planes(Surface* s, uint8_t* dst, hipStream_t str) {
  for (uint32_t p = 0; p < s->NumPlanes(); p++) {  // Tasks.cpp:832-849
    const size_t wb = s->WidthInBytes(p), rows = s->Height(p);
    if (!hip_ok(hipMemcpy2DAsync(dst, wb, (const void*)s->PlanePtr(p), s->Pitch(p), wb, rows, hipMemcpyDeviceToHost, str),
                "CudaDownloadSurface: hipMemcpy2DAsync")) {
      std::cerr << "  plane " << p << " dst " << (void*)dst << " dpitch " << wb << " src " << (void*)s->PlanePtr(p) << " spitch "
                << s->Pitch(p) << " width " << wb << " rows " << rows << std::endl;
      return false;
    }
    dst += wb * rows;
  }
  return true;
}
TaskExecStatus CudaDownloadSurface::Run() {
  const HipMark tick("CudaDownloadSurface::Run");
  auto* s = static_cast<Surface*>(GetInput(0));
  if (!s) return TASK_EXEC_FAIL;
  ClearOutputs();
  if (s->Empty() || s->HostMemSize() > pImpl->host->GetRawMemSize()) {
    std::cerr << "CudaDownloadSurface: surface is empty or larger (" << s->HostMemSize() << " B) than the downloader was built for ("
              << pImpl->host->GetRawMemSize() << " B)" << std::endl;
    return TASK_EXEC_FAIL;
  }
  DeviceScope scope(pImpl->sref.ctx);
  if (!download_planes(s, pImpl->host->GetDataAs<uint8_t>(), (hipStream_t)pImpl->sref.str)) return TASK_EXEC_FAIL;
  hip_stream_sync(&pImpl->sref);
  SetOutput(pImpl->host.get(), 0U);
  return TASK_EXEC_SUCCESS;
}
TaskExecStatus CudaDownloadSurface::DownloadInto(Surface* s, void* dst, size_t dst_bytes) {
  if (!s || !dst) return TASK_EXEC_FAIL;
  const size_t bytes = s->Empty() ? 0 : s->HostMemSize();
  if (!bytes || bytes > pImpl->host->GetRawMemSize() || bytes > dst_bytes) {
    std::cerr << "CudaDownloadSurface: surface is empty, larger (" << bytes << " B) than the downloader was built for ("
              << pImpl->host->GetRawMemSize() << " B) or than the destination (" << dst_bytes << " B)" << std::endl;
    return TASK_EXEC_FAIL;
  }
  DeviceScope scope(pImpl->sref.ctx);
  hipPointerAttribute_t attr{};
  const bool pinned_dst = (hipPointerGetAttributes(&attr, dst) == hipSuccess) && attr.type == hipMemoryTypeHost;
  if (!pinned_dst) (void)hipGetLastError();  // a pageable pointer makes hipPointerGetAttributes fail: not an error for us
  uint8_t* target = pinned_dst ? static_cast<uint8_t*>(dst) : pImpl->host->GetDataAs<uint8_t>();
  // A pageable destination takes two moves — DMA into the pinned staging buffer, host copy out of it — and the reference does them one
  // after the other.  Here the frame goes in pieces of ~2 MB, each DMA followed by a sequence number written into the task's completion
  // flag; the host copies piece k while the DMA engine delivers piece k + 1 (4K RGB into a numpy array: 686 -> 848 frames/s, the host copy
  // alone being the limit at 21 GB/s; profiles/r03_download_pipelined.txt).  Without a flag (VPF_HIP_SYNC_SPIN_US=0, or pinned memory exhausted): the two-step form.
  if (!pinned_dst && bytes >= (4u << 20) && flag_ready(&pImpl->sref)) {
    StreamRef* sr = &pImpl->sref;
    const hipStream_t st = (hipStream_t)sr->str;
    struct Piece { size_t off, len; uint32_t seq; };
    std::vector<Piece> pieces;
    size_t off = 0;
    bool ok = true;
    for (uint32_t p = 0; p < s->NumPlanes() && ok; p++) {
      const size_t wb = s->WidthInBytes(p), rows = s->Height(p);
      const size_t step = std::max<size_t>(1, (2u << 20) / std::max<size_t>(wb, 1));
      for (size_t r = 0; r < rows && ok; r += step) {
        const size_t n = std::min(step, rows - r);
        ok = hipMemcpy2DAsync(target + off, wb, (const uint8_t*)s->PlanePtr(p) + r * s->Pitch(p), s->Pitch(p), wb, n, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamWriteValue32(st, sr->flag_dev, sr->seq + 1, 0) == hipSuccess;
        if (ok) pieces.push_back(Piece{off, wb * n, ++sr->seq});
        off += wb * n;
      }
    }
    if (!ok) { (void)hipGetLastError(); sr->flag_failed = true; }
    for (const Piece& pc : pieces) {  // whatever was queued is consumed in order; a piece that does not arrive in time falls back to a stream sync
      if (!flag_wait(sr, pc.seq, 50 * sync_spin_ns())) (void)hipStreamSynchronize(st);
      host_copy_large(static_cast<uint8_t*>(dst) + pc.off, target + pc.off, pc.len);
    }
    if (ok) return TASK_EXEC_SUCCESS;
    // fall through: redo the whole frame the plain way
  }
  if (!download_planes(s, target, (hipStream_t)pImpl->sref.str)) return TASK_EXEC_FAIL;
  hip_stream_sync(&pImpl->sref);
  if (!pinned_dst) host_copy_large(dst, target, bytes);
  return TASK_EXEC_SUCCESS;
}

// ------------------------------------------------------------------------------------------ buffers
struct UploadBuffer::Impl {
  StreamRef sref;
  std::unique_ptr<CudaBuffer> buf;
};
UploadBuffer::UploadBuffer(HipStream str, HipContext ctx, uint32_t e, uint32_t n)
    : Task("HipUploadBuffer", numInputs, numOutputs, hip_stream_sync, nullptr), pImpl(new Impl{StreamRef{ctx, str}, nullptr}) {
  pImpl->buf.reset(CudaBuffer::Make(e, n, ctx));
}
UploadBuffer::~UploadBuffer() {}
UploadBuffer* UploadBuffer::Make(HipStream str, HipContext ctx, uint32_t e, uint32_t n) { return new UploadBuffer(str, ctx, e, n); }
TaskExecStatus UploadBuffer::Run() {
  const HipMark tick("UploadBuffer::Run");
  auto* host = static_cast<Buffer*>(GetInput(0));
  if (!host) return TASK_EXEC_FAIL;
  ClearOutputs();
  if (host->GetRawMemSize() < pImpl->buf->GetRawMemSize()) return TASK_EXEC_FAIL;
  DeviceScope scope(pImpl->sref.ctx);
  if (hipMemcpyAsync((void*)pImpl->buf->GpuMem(), host->GetRawMemPtr(), pImpl->buf->GetRawMemSize(), hipMemcpyHostToDevice,
                     (hipStream_t)pImpl->sref.str) != hipSuccess)
    return TASK_EXEC_FAIL;
  hip_stream_sync(&pImpl->sref);
  SetOutput(pImpl->buf.get(), 0U);
  return TASK_EXEC_SUCCESS;
}

struct DownloadCudaBuffer::Impl {
  StreamRef sref;
  std::unique_ptr<Buffer> host;
};
DownloadCudaBuffer::DownloadCudaBuffer(HipStream str, HipContext ctx, uint32_t e, uint32_t n)
    : Task("HipDownloadBuffer", numInputs, numOutputs, hip_stream_sync, nullptr), pImpl(new Impl{StreamRef{ctx, str}, nullptr}) {
  pImpl->host.reset(Buffer::MakeOwnMem((size_t)e * n, ctx ? ctx : (HipContext)-1));
}
DownloadCudaBuffer::~DownloadCudaBuffer() {}
DownloadCudaBuffer* DownloadCudaBuffer::Make(HipStream str, HipContext ctx, uint32_t e, uint32_t n) { return new DownloadCudaBuffer(str, ctx, e, n); }
TaskExecStatus DownloadCudaBuffer::Run() {
  const HipMark tick("DownloadCudaBuffer::Run");
  auto* b = static_cast<CudaBuffer*>(GetInput(0));
  if (!b) return TASK_EXEC_FAIL;
  ClearOutputs();
  if (b->GetRawMemSize() > pImpl->host->GetRawMemSize()) return TASK_EXEC_FAIL;
  DeviceScope scope(pImpl->sref.ctx);
  if (hipMemcpyAsync(pImpl->host->GetRawMemPtr(), (const void*)b->GpuMem(), b->GetRawMemSize(), hipMemcpyDeviceToHost,
                     (hipStream_t)pImpl->sref.str) != hipSuccess)
    return TASK_EXEC_FAIL;
  hip_stream_sync(&pImpl->sref);
  SetOutput(pImpl->host.get(), 0U);
  return TASK_EXEC_SUCCESS;
}

}  // namespace VPF
