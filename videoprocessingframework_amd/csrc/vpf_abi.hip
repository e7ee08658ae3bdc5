// vpf_abi.hip — the C ABI of libvpfhip (include/vpf_hip.h): argument validation, format dispatch,
// kernarg packing.  No CPU fallback of any kind: every success path ends in a HIP kernel launch.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <dlfcn.h>

#include "vpf_coef.h"
#include "vpf_internal.h"

namespace vpf {

bool make_yuv2rgb(int cs, int cr, Yuv2RgbCoef* o) { return coef_yuv2rgb(cs, cr, o); }

struct Rgb2YuvDec {
  int64_t m[3][3];
  int d[3];
};
static const Rgb2YuvDec kRgb2Yuv[2] = {
    {{{257000, 504000, 98000}, {-148000, -291000, 439000}, {439000, -368000, -71000}}, {16, 128, 128}},   // MPEG "YCbCr"
    {{{299000, 587000, 114000}, {-147108, -288804, 435912}, {614777, -514799, -99978}}, {0, 128, 128}}};  // JPEG "YUV"

bool make_rgb2yuv(int cr, Rgb2YuvCoef* o) {
  if (cr != VPF_MPEG && cr != VPF_JPEG) return false;
  const Rgb2YuvDec& m = kRgb2Yuv[cr];
  for (int k = 0; k < 3; k++) {
    for (int j = 0; j < 3; j++) o->m[k][j] = q6(m.m[k][j]);
    o->d[k] = q6((int64_t)m.d[k] * 1000000 + 500000);
  }
  return true;
}

// ------------------------------------------------------------------------------------------
// diagnostics: VPF_HIP_LOG / VPF_HIP_ROCTX (vpf_internal.h)
// ------------------------------------------------------------------------------------------
static std::atomic<int> g_log_level{-1}, g_trace{-1};
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static void (*g_roctx_mark)(const char*) = nullptr;
int log_level() {
  int v = g_log_level.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = std::getenv("VPF_HIP_LOG");
    v = e ? (std::atoi(e) > 0 ? std::atoi(e) : 1) : 0;
    g_log_level.store(v, std::memory_order_relaxed);
  }
  return v;
}
bool trace_on() {
  int v = g_trace.load(std::memory_order_acquire);
  if (v < 0) {
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    v = g_trace.load(std::memory_order_acquire);
    if (v < 0) {
      v = 0;
      const char* e = std::getenv("VPF_HIP_ROCTX");
      if (e && std::atoi(e) > 0) {
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
          g_roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
          g_roctx_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
          g_roctx_mark = reinterpret_cast<void (*)(const char*)>(dlsym(h, "roctxMarkA"));
        }
        if (g_roctx_push && g_roctx_pop) v = 1;
        else std::fprintf(stderr, "libvpfhip: VPF_HIP_ROCTX is set but no roctx library could be loaded; tracing stays off\n");
      }
      g_trace.store(v, std::memory_order_release);
    }
  }
  return v > 0;
}
void trace_push(const char* name) { if (g_roctx_push) g_roctx_push(name); }
void trace_pop() { if (g_roctx_pop) g_roctx_pop(); }
void note_kernel(const char* k) {
  if (log_level() >= 2) std::fprintf(stderr, "libvpfhip: launch %s\n", k);
  if (trace_on() && g_roctx_mark) g_roctx_mark(k);
}

static std::atomic<int> g_tune_variant{0}, g_tune_tile{0}, g_tune_band{0}, g_tune_mfma{0};
int tuning(int key) {
  return key == VPF_TUNE_NV12_RGB_VARIANT ? g_tune_variant.load() : key == VPF_TUNE_RESIZE_TILE ? g_tune_tile.load() : key == VPF_TUNE_RESIZE_BAND ? g_tune_band.load() : key == VPF_TUNE_RESIZE_MFMA ? g_tune_mfma.load() : 0;
}

// ------------------------------------------------------------------------------------------
// format helpers
// ------------------------------------------------------------------------------------------
static int num_planes(int f) {
  switch (f) {
    case VPF_FMT_Y: case VPF_FMT_RGB: case VPF_FMT_BGR: case VPF_FMT_RGB_32F: return 1;
    case VPF_FMT_NV12: case VPF_FMT_P10: case VPF_FMT_P12: return 2;
    case VPF_FMT_YUV420: case VPF_FMT_YCBCR: case VPF_FMT_YUV444: case VPF_FMT_RGB_PLANAR:
    case VPF_FMT_RGB_32F_PLANAR: return 3;
    default: return 0;
  }
}
static uint32_t row_bytes(int f, int k, uint32_t w) {
  const uint32_t cw = (w + 1) / 2;
  switch (f) {
    case VPF_FMT_Y: return w;
    case VPF_FMT_RGB: case VPF_FMT_BGR: return 3 * w;
    case VPF_FMT_NV12: return k == 0 ? w : 2 * cw;
    case VPF_FMT_YUV420: case VPF_FMT_YCBCR: return k == 0 ? w : cw;
    case VPF_FMT_YUV444: case VPF_FMT_RGB_PLANAR: return w;
    case VPF_FMT_RGB_32F: return 12 * w;
    case VPF_FMT_RGB_32F_PLANAR: return 4 * w;
    case VPF_FMT_P10: case VPF_FMT_P12: return k == 0 ? 2 * w : 4 * cw;
    default: return 0;
  }
}
// Frames per dispatch of the resize / fused batch entries (round 5).  A dispatch costs ~3 us between its last wave and the next one's first
// plus a tail of partly empty CUs of about one wave life: for SMALL planes (a 32-frame dispatch of Y 1080p -> 720p is 24 us of kernel) four
// times the frames per dispatch are worth 8-18 %; for larger ones they are level or a loss (RGB 1080p -> 720p Lanczos 2.04 -> 2.31 us per
// frame, RGB 720p -> 1080p bilinear 1.82 -> 2.14: dispatches of 60 us and more had little tail to amortise, and the launch planners are fitted
// at 32 and at 128 frames only).  Measured over seven shapes x 32 / 64 / 96 / 128 frames (profiles/r05_frames_per_dispatch_curve.txt): the
// gain sits with the shapes that move up to ~7 MB per frame (source + destination).
// Round 6: between the two, bilinear / nearest frames of 7-10 MB (packed RGB 1080p <-> 720p) go 64 to a dispatch: the r05 curve has them at 1.72 -> 1.60 us
// per frame (1080p -> 720p) and 1.82 -> 1.74 (720p -> 1080p) at 64 and losing again at 128; the Lanczos kernel is level at 64 and keeps 32.
static uint32_t frames_per_dispatch(uint64_t bytes_per_frame, bool mid_ok = false) {
  static const bool mid_on = [] { const char* e = std::getenv("VPF_HIP_MID_BATCH"); return !(e && e[0] == '0'); }();  // (A/B: VPF_HIP_MID_BATCH=0 keeps 32)
  mid_ok = mid_ok && mid_on;
  return bytes_per_frame <= 7000000ull ? (uint32_t)kMaxBatch : (mid_ok && bytes_per_frame <= 10000000ull) ? 64u : (uint32_t)kSmallBatch;
}
static uint64_t frame_bytes(int f, vpf_size s) {
  uint64_t b = 0;
  for (int k = 0; k < num_planes(f); k++) {
    const bool half = k > 0 && (f == VPF_FMT_NV12 || f == VPF_FMT_YUV420 || f == VPF_FMT_YCBCR || f == VPF_FMT_P10 || f == VPF_FMT_P12);
    b += (uint64_t)row_bytes(f, k, s.width) * (half ? (s.height + 1) / 2 : s.height);
  }
  return b;
}
// Pictures larger than 65536 in either dimension are refused: beyond that 12 * width (RGB_32F row bytes) and the kernels'
// 32-bit in-row offsets could wrap, and no video surface is that large.
static bool dims_ok(vpf_size s) { return s.width && s.height && s.width <= 65536u && s.height <= 65536u; }
static bool planes_ok(int f, uint32_t w, const vpf_plane* p) {
  const int n = num_planes(f);
  if (!n || !p) return false;
  for (int k = 0; k < n; k++)
    if (!p[k].ptr || p[k].pitch < row_bytes(f, k, w)) return false;
  return true;
}
static int yuv_src_class(int f) {
  switch (f) {
    case VPF_FMT_NV12: return FC_NV12;
    case VPF_FMT_YUV420: return FC_YUV420;
    case VPF_FMT_YUV444: return FC_YUV444;
    default: return -1;
  }
}
static int rgb_class(int f) {
  switch (f) {
    case VPF_FMT_RGB: return FC_RGB;
    case VPF_FMT_BGR: return FC_BGR;
    case VPF_FMT_RGB_PLANAR: return FC_PLANAR;
    default: return -1;
  }
}
static bool cscr_ok(int cs, int cr) {
  return (cs == VPF_BT_601 || cs == VPF_BT_709) && (cr == VPF_MPEG || cr == VPF_JPEG);
}

enum Family { FAM_NONE, FAM_YUV2RGB, FAM_RGB2YUV, FAM_RELAYOUT };
static Family classify(int sf, int df, int cs, int cr) {
  if (yuv_src_class(sf) >= 0 && rgb_class(df) >= 0) return cscr_ok(cs, cr) ? FAM_YUV2RGB : FAM_NONE;
  if (rgb_class(sf) >= 0 && (df == VPF_FMT_YUV444 || df == VPF_FMT_YUV420 || df == VPF_FMT_YCBCR))
    return (cs == VPF_BT_601 && (cr == VPF_MPEG || cr == VPF_JPEG)) ? FAM_RGB2YUV : FAM_NONE;
  if (sf == VPF_FMT_NV12 && (df == VPF_FMT_YUV420 || df == VPF_FMT_Y)) return FAM_RELAYOUT;
  if (sf == VPF_FMT_YUV420 && df == VPF_FMT_NV12) return FAM_RELAYOUT;
  if (rgb_class(sf) >= 0 && rgb_class(df) >= 0 && sf != df) return FAM_RELAYOUT;
  if (sf == VPF_FMT_Y && df == VPF_FMT_YUV444) return FAM_RELAYOUT;
  if (rgb_class(sf) >= 0 && df == VPF_FMT_Y) return FAM_RELAYOUT;
  if (sf == VPF_FMT_RGB && df == VPF_FMT_RGB_32F) return FAM_RELAYOUT;
  if (sf == VPF_FMT_RGB_32F && df == VPF_FMT_RGB_32F_PLANAR) return FAM_RELAYOUT;
  if ((sf == VPF_FMT_P10 || sf == VPF_FMT_P12) && df == VPF_FMT_NV12) return FAM_RELAYOUT;
  return FAM_NONE;
}

// RAII device guard: the ABI never leaves the caller's current device changed
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = (err == hipSuccess);
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

static void fill_desc(FrameDesc& d, const vpf_plane* s, int ns, const vpf_plane* o, int nd) {
  std::memset(&d, 0, sizeof(d));
  for (int k = 0; k < ns; k++) { d.s[k] = static_cast<const uint8_t*>(s[k].ptr); d.sp[k] = s[k].pitch; }
  for (int k = 0; k < nd; k++) { d.d[k] = static_cast<uint8_t*>(o[k].ptr); d.dp[k] = o[k].pitch; }
}

static vpf_status status_of(hipError_t e) {
  if (e == hipSuccess) return VPF_OK;
  if (log_level() >= 1) std::fprintf(stderr, "libvpfhip: %s (%s)\n", hipGetErrorName(e), hipGetErrorString(e));
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) return VPF_ERR_NO_DEVICE;
  return VPF_ERR_LAUNCH;
}

}  // namespace vpf

using namespace vpf;

extern "C" {

int vpf_convert_supported(int sf, int df, int cs, int cr) { return classify(sf, df, cs, cr) != FAM_NONE; }

vpf_status vpf_convert_batch(const vpf_exec* exec, int sf, int df, int cs, int cr, vpf_size size, uint32_t n,
                             const vpf_frame_io* frames) {
  const Mark mark("vpf_convert_batch");
  const Family fam = classify(sf, df, cs, cr);
  if (fam == FAM_NONE) return VPF_ERR_UNSUPPORTED;
  if (!exec || !frames || !n || !dims_ok(size)) return VPF_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n; i++)
    if (!planes_ok(sf, size.width, frames[i].src) || !planes_ok(df, size.width, frames[i].dst)) return VPF_ERR_BAD_ARG;
  DeviceGuard guard(exec->device);
  if (guard.err != hipSuccess) return status_of(guard.err);
  hipStream_t st = static_cast<hipStream_t>(exec->stream);
  const int ns = num_planes(sf), nd = num_planes(df);
  Yuv2RgbCoef yc;
  Rgb2YuvCoef rc;
  if (fam == FAM_YUV2RGB) make_yuv2rgb(cs, cr, &yc);
  if (fam == FAM_RGB2YUV) make_rgb2yuv(cr, &rc);
  for (uint32_t base = 0; base < n; base += kSmallBatch) {
    const uint32_t m = (n - base < (uint32_t)kSmallBatch) ? n - base : (uint32_t)kSmallBatch;
    BatchArgs a;
    for (uint32_t i = 0; i < m; i++) fill_desc(a.f[i], frames[base + i].src, ns, frames[base + i].dst, nd);
    for (uint32_t i = m; i < (uint32_t)kSmallBatch; i++) a.f[i] = a.f[0];
    hipError_t e;
    switch (fam) {
      case FAM_YUV2RGB:
        e = launch_yuv_to_rgb(st, yuv_src_class(sf), rgb_class(df), yc, size.width, size.height, m, a,
                              tuning(VPF_TUNE_NV12_RGB_VARIANT), (exec->flags & VPF_EXEC_DST_REUSED) != 0);
        break;
      case FAM_RGB2YUV:
        e = launch_rgb_to_yuv(st, rgb_class(sf), df == VPF_FMT_YUV444 ? FC_YUV444 : FC_YUV420, rc, size.width,
                              size.height, m, a);
        break;
      default: e = launch_relayout(st, sf, df, size.width, size.height, m, a);
    }
    if (e != hipSuccess) return status_of(e);
  }
  return VPF_OK;
}

vpf_status vpf_convert(const vpf_exec* exec, int sf, int df, int cs, int cr, vpf_size size, const vpf_plane src[3],
                       const vpf_plane dst[3]) {
  if (classify(sf, df, cs, cr) == FAM_NONE) return VPF_ERR_UNSUPPORTED;
  if (!src || !dst) return VPF_ERR_BAD_ARG;
  vpf_frame_io io;
  std::memset(&io, 0, sizeof(io));
  for (int k = 0; k < num_planes(sf); k++) io.src[k] = src[k];
  for (int k = 0; k < num_planes(df); k++) io.dst[k] = dst[k];
  return vpf_convert_batch(exec, sf, df, cs, cr, size, 1, &io);
}

// planes of `fmt` as resize jobs: (interleaved channels, plane index, plane size in pixels)
static int resize_jobs(int fmt, vpf_size ss, vpf_size ds, ResizeJob jobs[3], bool* f32) {
  const uint32_t scw = (ss.width + 1) / 2, sch = (ss.height + 1) / 2, dcw = (ds.width + 1) / 2, dch = (ds.height + 1) / 2;
  *f32 = fmt == VPF_FMT_RGB_32F || fmt == VPF_FMT_RGB_32F_PLANAR;
  const ResizeJob full1 = {1, 0, ss.width, ss.height, ds.width, ds.height};
  switch (fmt) {
    case VPF_FMT_RGB: case VPF_FMT_BGR: case VPF_FMT_RGB_32F: jobs[0] = ResizeJob{3, 0, ss.width, ss.height, ds.width, ds.height}; return 1;
    case VPF_FMT_Y: jobs[0] = full1; return 1;
    case VPF_FMT_YUV444: case VPF_FMT_RGB_PLANAR: case VPF_FMT_RGB_32F_PLANAR:
      for (int k = 0; k < 3; k++) { jobs[k] = full1; jobs[k].k = k; }
      return 3;
    case VPF_FMT_YUV420: case VPF_FMT_YCBCR:
      jobs[0] = full1;
      for (int k = 1; k < 3; k++) jobs[k] = ResizeJob{1, k, scw, sch, dcw, dch};
      return 3;
    case VPF_FMT_NV12:  // luma plane + the UV plane as a 2-channel image (== C3 -> R2 -> C4 of Tasks.cpp:1303-1318)
      jobs[0] = full1;
      jobs[1] = ResizeJob{2, 1, scw, sch, dcw, dch};
      return 2;
    default: return 0;
  }
}

vpf_status vpf_resize_batch(const vpf_exec* exec, int fmt, int interp, vpf_size ss, vpf_size ds, uint32_t n, const vpf_frame_io* frames) {
  const Mark mark("vpf_resize_batch");
  if (interp != VPF_INTERP_NEAREST && interp != VPF_INTERP_LINEAR && interp != VPF_INTERP_LANCZOS3) return VPF_ERR_UNSUPPORTED;
  ResizeJob jobs[3];
  bool f32 = false;
  const int nj = resize_jobs(fmt, ss, ds, jobs, &f32);
  if (!nj) return VPF_ERR_UNSUPPORTED;
  if (!exec || !frames || !n || !dims_ok(ss) || !dims_ok(ds)) return VPF_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n; i++) {
    if (!planes_ok(fmt, ss.width, frames[i].src) || !planes_ok(fmt, ds.width, frames[i].dst)) return VPF_ERR_BAD_ARG;
    if (f32)  // float samples: rows must be 4-B aligned
      for (int k = 0; k < num_planes(fmt); k++)
        if ((((uintptr_t)frames[i].src[k].ptr | frames[i].src[k].pitch | (uintptr_t)frames[i].dst[k].ptr | frames[i].dst[k].pitch) & 3)) return VPF_ERR_BAD_ARG;
  }
  DeviceGuard guard(exec->device);
  if (guard.err != hipSuccess) return status_of(guard.err);
  hipStream_t st = static_cast<hipStream_t>(exec->stream);
  const int np = num_planes(fmt);
  const bool forced_family = (tuning(VPF_TUNE_RESIZE_MFMA) & 0xffff) >= 2 || (tuning(VPF_TUNE_RESIZE_BAND) & 0xffff) >= 2 || (tuning(VPF_TUNE_RESIZE_BAND) >> 16);  // measurement runs: the batch kernels on one frame
  if (n == 1 && nj == 1 && !forced_family) {  // one plane of one frame: the scalar-argument kernel entries (kernarg preload)
    const vpf_plane &s0 = frames[0].src[0], &d0 = frames[0].dst[0];
    const hipError_t e = f32 ? launch_resize_f32(st, jobs[0].ch, interp, ss.width, ss.height, static_cast<const uint8_t*>(s0.ptr), s0.pitch, ds.width, ds.height,
                                                 static_cast<uint8_t*>(d0.ptr), d0.pitch)
                             : launch_resize(st, jobs[0].ch, interp, ss.width, ss.height, static_cast<const uint8_t*>(s0.ptr), s0.pitch, ds.width, ds.height,
                                             static_cast<uint8_t*>(d0.ptr), d0.pitch);
    return status_of(e);
  }
  const uint32_t per = frames_per_dispatch(frame_bytes(fmt, ss) + frame_bytes(fmt, ds), interp != VPF_INTERP_LANCZOS3);  // 128 for small planes, 64 for mid-sized bilinear ones, else 32 (the launchers take the small frame table for up to 32)
  for (uint32_t base = 0; base < n; base += per) {
    const uint32_t m = (n - base < per) ? n - base : per;
    BatchArgsL a;
    for (uint32_t i = 0; i < m; i++) fill_desc(a.f[i], frames[base + i].src, np, frames[base + i].dst, np);
    for (uint32_t i = m; i < ((m + 7u) & ~7u); i++) a.f[i] = a.f[0];  // (entries beyond the batch are never read: blockIdx runs over m frames; a few copies keep the tail of the last cache line defined)
    const hipError_t e = launch_resize_jobs(st, f32, interp, nj, jobs, m, a);
    if (e != hipSuccess) return status_of(e);
  }
  return VPF_OK;
}

vpf_status vpf_resize(const vpf_exec* exec, int fmt, int interp, vpf_size ss, const vpf_plane src[3], vpf_size ds,
                      const vpf_plane dst[3]) {
  if (interp != VPF_INTERP_NEAREST && interp != VPF_INTERP_LINEAR && interp != VPF_INTERP_LANCZOS3) return VPF_ERR_UNSUPPORTED;
  ResizeJob probe[3];
  bool f32 = false;
  if (!resize_jobs(fmt, vpf_size{2, 2}, vpf_size{2, 2}, probe, &f32)) return VPF_ERR_UNSUPPORTED;
  if (!src || !dst) return VPF_ERR_BAD_ARG;
  vpf_frame_io io;
  std::memset(&io, 0, sizeof(io));
  for (int k = 0; k < num_planes(fmt); k++) { io.src[k] = src[k]; io.dst[k] = dst[k]; }
  return vpf_resize_batch(exec, fmt, interp, ss, ds, 1, &io);
}

// the workspace forms: the same calls with the caller's table region made known to the Lanczos launcher for their duration
struct WorkspaceScope {
  explicit WorkspaceScope(vpf_workspace* ws) { set_lanczos_workspace(ws); }
  ~WorkspaceScope() { set_lanczos_workspace(nullptr); }
};
vpf_status vpf_resize_batch_ws(const vpf_exec* exec, int fmt, int interp, vpf_size ss, vpf_size ds, uint32_t n, const vpf_frame_io* frames, vpf_workspace* ws) {
  const WorkspaceScope scope(ws);
  return vpf_resize_batch(exec, fmt, interp, ss, ds, n, frames);
}
vpf_status vpf_resize_ws(const vpf_exec* exec, int fmt, int interp, vpf_size ss, const vpf_plane src[3], vpf_size ds, const vpf_plane dst[3], vpf_workspace* ws) {
  const WorkspaceScope scope(ws);
  return vpf_resize(exec, fmt, interp, ss, src, ds, dst);
}
uint64_t vpf_resize_workspace_bytes(int fmt, int interp, vpf_size ss, vpf_size ds) {
  ResizeJob jobs[3];
  bool f32 = false;
  if (interp != VPF_INTERP_LANCZOS3 || !dims_ok(ss) || !dims_ok(ds)) return 0;
  const int nj = resize_jobs(fmt, ss, ds, jobs, &f32);
  if (!nj || f32) return 0;
  uint64_t total = 256;  // (offset 0 of a region means "no table": its first 256 bytes stay unused)
  for (int p = 0; p < nj; p++) total += lanczos_table_bytes_bound(jobs[p].ch, jobs[p].dw, jobs[p].dh);
  return total;
}

vpf_status vpf_remap_batch(const vpf_exec* exec, int fmt, vpf_size ss, const float* xmap, uint32_t xp, const float* ymap, uint32_t yp, vpf_size ds,
                           uint32_t n, const vpf_frame_io* frames) {
  const Mark mark("vpf_remap_batch");
  if (fmt != VPF_FMT_RGB && fmt != VPF_FMT_BGR) return VPF_ERR_UNSUPPORTED;
  if (!exec || !frames || !n || !dims_ok(ss) || !dims_ok(ds) || !xmap || !ymap || xp < 4 * ds.width || yp < 4 * ds.width || (xp & 3) || (yp & 3) ||
      ((uintptr_t)xmap & 3) || ((uintptr_t)ymap & 3))
    return VPF_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n; i++)
    if (!planes_ok(fmt, ss.width, frames[i].src) || !planes_ok(fmt, ds.width, frames[i].dst)) return VPF_ERR_BAD_ARG;
  DeviceGuard guard(exec->device);
  if (guard.err != hipSuccess) return status_of(guard.err);
  for (uint32_t base = 0; base < n; base += kSmallBatch) {
    const uint32_t m = (n - base < (uint32_t)kSmallBatch) ? n - base : (uint32_t)kSmallBatch;
    BatchArgs a;
    for (uint32_t i = 0; i < m; i++) fill_desc(a.f[i], frames[base + i].src, 1, frames[base + i].dst, 1);
    for (uint32_t i = m; i < (uint32_t)kSmallBatch; i++) a.f[i] = a.f[0];
    const hipError_t e = launch_remap_batch(static_cast<hipStream_t>(exec->stream), ss.width, ss.height, xmap, xp, ymap, yp, ds.width, ds.height, m, a);
    if (e != hipSuccess) return status_of(e);
  }
  return VPF_OK;
}

vpf_status vpf_remap(const vpf_exec* exec, int fmt, vpf_size ss, const vpf_plane* src, const float* xmap, uint32_t xp,
                     const float* ymap, uint32_t yp, vpf_size ds, const vpf_plane* dst) {
  const Mark mark("vpf_remap");
  if (fmt != VPF_FMT_RGB && fmt != VPF_FMT_BGR) return VPF_ERR_UNSUPPORTED;
  if (!exec || !dims_ok(ss) || !dims_ok(ds) || !xmap || !ymap || !planes_ok(fmt, ss.width, src) ||
      !planes_ok(fmt, ds.width, dst) || xp < 4 * ds.width || yp < 4 * ds.width || (xp & 3) || (yp & 3) ||
      ((uintptr_t)xmap & 3) || ((uintptr_t)ymap & 3))
    return VPF_ERR_BAD_ARG;
  DeviceGuard guard(exec->device);
  if (guard.err != hipSuccess) return status_of(guard.err);
  return status_of(launch_remap(static_cast<hipStream_t>(exec->stream), ss.width, ss.height,
                                static_cast<const uint8_t*>(src->ptr), src->pitch, xmap, xp, ymap, yp, ds.width,
                                ds.height, static_cast<uint8_t*>(dst->ptr), dst->pitch));
}

vpf_status vpf_convert_resize_batch(const vpf_exec* exec, int sf, int df, int cs, int cr, vpf_size ss, vpf_size ds,
                                    uint32_t n, const vpf_frame_io* frames) {
  const Mark mark("vpf_convert_resize_batch");
  if (!(sf == VPF_FMT_NV12 || sf == VPF_FMT_YUV420) || rgb_class(df) < 0 || !cscr_ok(cs, cr)) return VPF_ERR_UNSUPPORTED;
  if (!exec || !frames || !n || !dims_ok(ss) || !dims_ok(ds)) return VPF_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n; i++)
    if (!planes_ok(sf, ss.width, frames[i].src) || !planes_ok(df, ds.width, frames[i].dst)) return VPF_ERR_BAD_ARG;
  DeviceGuard guard(exec->device);
  if (guard.err != hipSuccess) return status_of(guard.err);
  Yuv2RgbCoef c;
  make_yuv2rgb(cs, cr, &c);
  const uint32_t per = frames_per_dispatch(frame_bytes(sf, ss) + frame_bytes(df, ds), true);  // 128 for small frames (1080p -> 720p and smaller), 64 for 7-10 MB ones (720p -> 1080p), else 32
  for (uint32_t base = 0; base < n; base += per) {
    const uint32_t m = (n - base < per) ? n - base : per;
    BatchArgsL a;
    for (uint32_t i = 0; i < m; i++) fill_desc(a.f[i], frames[base + i].src, num_planes(sf), frames[base + i].dst, num_planes(df));
    for (uint32_t i = m; i < ((m + 7u) & ~7u); i++) a.f[i] = a.f[0];  // (entries beyond the batch are never read: blockIdx runs over m frames; a few copies keep the tail of the last cache line defined)
    const hipError_t e = launch_convert_resize(static_cast<hipStream_t>(exec->stream), yuv_src_class(sf), rgb_class(df), c,
                                               ss.width, ss.height, m, a, ds.width, ds.height);
    if (e != hipSuccess) return status_of(e);
  }
  return VPF_OK;
}

vpf_status vpf_convert_resize(const vpf_exec* exec, int sf, int df, int cs, int cr, vpf_size ss, const vpf_plane src[3],
                              vpf_size ds, const vpf_plane dst[3]) {
  if (!(sf == VPF_FMT_NV12 || sf == VPF_FMT_YUV420) || rgb_class(df) < 0 || !cscr_ok(cs, cr)) return VPF_ERR_UNSUPPORTED;
  if (!src || !dst) return VPF_ERR_BAD_ARG;
  vpf_frame_io io;
  std::memset(&io, 0, sizeof(io));
  for (int k = 0; k < num_planes(sf); k++) io.src[k] = src[k];
  for (int k = 0; k < num_planes(df); k++) io.dst[k] = dst[k];
  return vpf_convert_resize_batch(exec, sf, df, cs, cr, ss, ds, 1, &io);
}

const char* vpf_status_string(int s) {
  switch (s) {
    case VPF_OK: return "ok";
    case VPF_ERR_UNSUPPORTED: return "unsupported format / colour-space combination";
    case VPF_ERR_BAD_ARG: return "bad argument";
    case VPF_ERR_LAUNCH: return "HIP launch failed";
    case VPF_ERR_NO_DEVICE: return "no usable HIP device";
    default: return "unknown status";
  }
}
const char* vpf_version(void) { return "vpf-hip 0.1 (gfx950)"; }
int vpf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
/* Additive, for the Task layer's marks (the reference brackets every Task::Run with an NvtxMark): a roctx range when VPF_HIP_ROCTX=1,
 * nothing otherwise.  vpf_trace_push returns 1 if a range was opened (pass that to vpf_trace_pop). */
int vpf_trace_push(const char* name) {
  if (!trace_on()) return 0;
  trace_push(name);
  return 1;
}
void vpf_trace_pop(int opened) { if (opened) trace_pop(); }

int vpf_set_tuning(int key, int value) {
  if (key == VPF_TUNE_RESIZE_TILE) {
    const int ty = value & 0xff, wpb = value >> 8;
    if (value != 0 && (ty < 4 || ty > 64 || (ty & 3) || (wpb != 4 && wpb != 8))) return -1;
    return g_tune_tile.exchange(value);
  }
  if (key == VPF_TUNE_RESIZE_MFMA) {
    const int shape = value & 0xffff, nt = shape >> 8, tiles = shape & 0xff;  // | 0x10000: no weight tables; | 0x20000: the two-role kernel form; | 0x40000: small single planes too; | 0x80000: no ring of two
    if (value < 0 || (value & ~0xfffff)) return -1;
#ifndef VPF_LAB_FORMS
    if (value & 0x20000) return -1;  // the two-role form lives in the lab build (tools/lab/libvpfhip_forms.so)
#endif
    if (shape != 0 && shape != 1 && ((nt != 0 && nt != 4 && nt != 8) || tiles > 64 || (nt == 0 && tiles < 2))) return -1;
    return g_tune_mfma.exchange(value);
  }
  if (key == VPF_TUNE_RESIZE_BAND) {
    const int rows = value & 0xff, nb = (value >> 8) & 0xff, form = value >> 16;  // nb: bands per wave of the march form (4-row bands only) / bands per chunk of the persistent launch
    // form: | 1 the persistent launch, | 2 eight pixels per lane on every 1-channel plane, | 4 persistent waves visit every XCD's counter
    const bool ok = value >= 0 && (rows == 0 || rows == 1 || rows == 2 || rows == 4 || rows == 8 || rows == 16) && nb <= 8 && (nb == 0 || rows == 4 || (form & 1)) && form <= 15 && (!(form & 12) || (form & 1));  // | 8 (measurement) persistent waves stride through their share, no counters
#ifndef VPF_LAB_FORMS
    if (form & 13) return -1;  // the persistent launch and its knobs live in the lab build (tools/lab/libvpfhip_forms.so)
#endif
    return ok ? g_tune_band.exchange(value) : -1;
  }
  if (key != VPF_TUNE_NV12_RGB_VARIANT) return -1;
  switch (value) {  // the kernels libvpfhip contains: every one writes the same pixels (include/vpf_hip.h)
#ifdef VPF_LAB_FORMS
    case 47:  // the per-wave strips of the fused kernel (rounds 2-4): lab build only
#endif
    case 0: case 4: case 8: case 9: case 12: case 30: case 37: case 40: case 43: case 44: case 45: case 46: case 48: case 49: return g_tune_variant.exchange(value);
    default: return -1;  // unknown value: nothing changes
  }
}

}  // extern "C"
