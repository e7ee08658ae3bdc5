/*
 * vpf_hip.h — C ABI of libvpfhip: MI355X (gfx950) surface conversion / resize / remap.
 *
 * This is the drop-in boundary.  In the reference (NVIDIA/VideoProcessingFramework) the
 * C++ Task layer calls closed-source NPP at exactly this edge:
 *
 *   reference caller                                      reference callee (NPP)            replaced by
 *   ----------------------------------------------------  --------------------------------  -------------------
 *   nv12_rgb::Execute        src/TC/src/TasksColorCvt.cpp:122-182  nppiNV12ToRGB_*_8u_P2C3R_Ctx     vpf_convert
 *   nv12_bgr::Execute        TasksColorCvt.cpp:53-108             nppiNV12ToBGR_*_8u_P2C3R_Ctx     vpf_convert
 *   nv12_yuv420::Execute     TasksColorCvt.cpp:196-240            nppiNV12ToYUV420 / nppiYCbCr420  vpf_convert
 *   yuv420_nv12::Execute     TasksColorCvt.cpp:945-975            nppiYCbCr420_8u_P3P2R            vpf_convert
 *   yuv420_rgb/_bgr          TasksColorCvt.cpp:322-369,383-430    nppiYUV420ToRGB / YCbCr420ToRGB  vpf_convert
 *   rgb8_deinterleave/_inter TasksColorCvt.cpp:1059-1088,1102-1131 nppiCopy_8u_C3P3R / _P3C3R      vpf_convert
 *   rgb_bgr / bgr_rgb        TasksColorCvt.cpp:1145-1170,1184-1209 nppiSwapChannels_8u_C3R         vpf_convert
 *   (all other *_Impl in TasksColorCvt.cpp:245-1300)                                               vpf_convert
 *   NppResizeSurfacePacked3C_Impl::Run   src/TC/src/Tasks.cpp:1162-1203  nppiResize_8u_C3R         vpf_resize
 *   NppResizeSurfacePlanar_Impl::Run     Tasks.cpp:1217-1261             nppiResize_8u_C1R         vpf_resize
 *   NppResizeSurfacePacked32F3C_Impl / NppResizeSurface32FPlanar_Impl  Tasks.cpp:1334-1445  nppiResize_32f_C3R / _C1R  vpf_resize
 *   NppRemapSurfacePacked3C_Impl::Run    Tasks.cpp:1555-1602             nppiRemap_8u_C3R          vpf_remap
 *
 * Like the NPP `_Ctx` entry points it replaces, every function here takes raw device pointers,
 * byte pitches, a size, and the stream to launch on; it owns nothing, allocates nothing, never
 * synchronises, and reports failure through its return code (no C++ exceptions cross this edge).
 * All launches are asynchronous on `exec->stream`.
 *
 * Pure C: usable from C, C++, ctypes, cgo, JNI, ... No torch / HIP types in any signature
 * (`stream` is a hipStream_t passed as void*).
 */
#ifndef VPF_HIP_H_
#define VPF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define VPF_API __attribute__((visibility("default")))
#else
#define VPF_API
#endif

/* Pixel formats. Numeric values are the reference's `Pixel_Format`
 * (src/TC/inc/MemoryInterfaces.hpp:30-49); they are part of the Python ABI. */
typedef enum vpf_pixel_format {
  VPF_FMT_UNDEFINED = 0,
  VPF_FMT_Y = 1,
  VPF_FMT_RGB = 2,
  VPF_FMT_NV12 = 3,
  VPF_FMT_YUV420 = 4,
  VPF_FMT_RGB_PLANAR = 5,
  VPF_FMT_BGR = 6,
  VPF_FMT_YCBCR = 7,
  VPF_FMT_YUV444 = 8,
  VPF_FMT_RGB_32F = 9,
  VPF_FMT_RGB_32F_PLANAR = 10,
  VPF_FMT_YUV422 = 11,
  VPF_FMT_P10 = 12,
  VPF_FMT_P12 = 13,
  VPF_FMT_YUV444_10bit = 14,
  VPF_FMT_YUV420_10bit = 15,
  VPF_FMT_NV12_PLANAR = 16,
  VPF_FMT_GRAY12 = 17
} vpf_pixel_format;

/* src/TC/inc/MemoryInterfaces.hpp:51-61 */
typedef enum vpf_color_space { VPF_BT_601 = 0, VPF_BT_709 = 1, VPF_CS_UNSPEC = 2 } vpf_color_space;
typedef enum vpf_color_range { VPF_MPEG = 0, VPF_JPEG = 1, VPF_CR_UDEF = 2 } vpf_color_range;

typedef enum vpf_interp {
  VPF_INTERP_NEAREST = 0,
  VPF_INTERP_LINEAR = 1,
  VPF_INTERP_LANCZOS3 = 2
} vpf_interp;

typedef enum vpf_status {
  VPF_OK = 0,
  VPF_ERR_UNSUPPORTED = 1, /* format pair / matrix not implemented              */
  VPF_ERR_BAD_ARG = 2,     /* null pointer, zero size, pitch < row bytes, ...    */
  VPF_ERR_LAUNCH = 3,      /* HIP runtime reported an error at launch            */
  VPF_ERR_NO_DEVICE = 4    /* no usable gfx950 device                            */
} vpf_status;

/* One 2-D plane in device memory.  `pitch` is in bytes.  16 bytes, no implicit padding. */
typedef struct vpf_plane {
  void* ptr;
  uint32_t pitch;
  uint32_t reserved; /* must be 0 */
} vpf_plane;

/* Size in pixels of the full-resolution image (plane 0 for YUV formats).  1 .. 65536 per dimension; anything else is
 * VPF_ERR_BAD_ARG (row-byte and offset arithmetic is 32-bit). */
typedef struct vpf_size {
  uint32_t width;
  uint32_t height;
} vpf_size;

/* Where and on which stream to run.  Replaces NppStreamContext
 * (src/TC/src/NppCommon.cpp:10-60).  device < 0 means "current device". */
typedef struct vpf_exec {
  int32_t device;
  uint32_t flags; /* 0 or VPF_EXEC_* hints below; unknown bits are ignored */
  void* stream;   /* hipStream_t; NULL = default stream */
} vpf_exec;

/* Hint: the destination is read again right away (the next kernel of a per-frame chain).  MI355X has a 256 MiB
 * last-level Infinity Cache; by default the converters stream their output past it with non-temporal stores (fastest
 * when nothing re-reads it, e.g. 32-frame batches).  With this flag the single-frame NV12 -> RGB / BGR / RGB_PLANAR
 * kernels use allocating stores instead, so a consumer launched next finds its input on chip (measured: 4K NV12 -> RGB ->
 * RGB_PLANAR 17.7 -> 16.4 us per frame; the converter alone is ~12 % slower).  Pixels are identical either way. */
#define VPF_EXEC_DST_REUSED 1u

/*
 * Plane conventions for `src[]` / `dst[]` (unused entries are ignored, may be zeroed):
 *   Y                      [0] = W x H bytes
 *   NV12                   [0] = Y  (W x H), [1] = interleaved UV (2*ceil(W/2) bytes x ceil(H/2) rows)
 *   YUV420, YCBCR          [0] = Y, [1] = U (ceil(W/2) x ceil(H/2)), [2] = V
 *   YUV444, RGB_PLANAR     [0],[1],[2] = three W x H planes (the reference stacks them in one
 *                          allocation, plane i at base + i*H*pitch: MemoryInterfaces.cpp:1593-1600;
 *                          the caller resolves that to three pointers)
 *   RGB, BGR               [0] = packed 3 bytes / pixel (3W bytes x H)
 *   RGB_32F                [0] = packed 3 floats / pixel; RGB_32F_PLANAR: three float planes
 *   P10, P12               [0] = Y 16-bit, [1] = interleaved UV 16-bit (MSB-aligned samples)
 */

/* Colour / layout conversion of one frame.  Replaces every nppi* call in TasksColorCvt.cpp. */
VPF_API vpf_status vpf_convert(const vpf_exec* exec, int src_fmt, int dst_fmt, int color_space,
                               int color_range, vpf_size size, const vpf_plane src[3],
                               const vpf_plane dst[3]);

typedef struct vpf_frame_io {
  vpf_plane src[3];
  vpf_plane dst[3];
} vpf_frame_io;

/* The same conversion over `n` independent frames of identical size/format, dispatched as few
 * launches as possible (one per 32 frames).  `frames` is a HOST array, consumed before return.
 * Exists because a 4K frame is ~5 us of HBM time, the same order as a kernel boundary. */
VPF_API vpf_status vpf_convert_batch(const vpf_exec* exec, int src_fmt, int dst_fmt,
                                     int color_space, int color_range, vpf_size size, uint32_t n,
                                     const vpf_frame_io* frames);

/* 1 if vpf_convert implements (src_fmt,dst_fmt) under (color_space,color_range), else 0.
 * Pure host logic; callable without a GPU. */
VPF_API int vpf_convert_supported(int src_fmt, int dst_fmt, int color_space, int color_range);

/* Whole-image resize.  `fmt` in {RGB, BGR, Y, YUV420, YCBCR, YUV444, RGB_PLANAR, NV12, RGB_32F,
 * RGB_32F_PLANAR}; every plane is resized independently (chroma planes at their own resolution).
 * Replaces nppiResize_8u_C3R / _C1R (Tasks.cpp:1193,1227-1253) and nppiResize_32f_C3R / _C1R
 * (Tasks.cpp:1376,1434; float results are neither rounded nor clamped, rows must be 4-B aligned). */
VPF_API vpf_status vpf_resize(const vpf_exec* exec, int fmt, int interp, vpf_size src_size,
                              const vpf_plane src[3], vpf_size dst_size, const vpf_plane dst[3]);

/* The same resize over `n` independent same-shape frames, every plane of every frame in as few dispatches as possible (one per 128
 * frames when a frame moves at most 7 000 000 bytes, source + destination — Y / NV12 / YUV420 at 1080p -> 720p and smaller —, one per 64
 * frames for bilinear / nearest frames of up to 10 000 000 bytes — packed RGB 1080p <-> 720p —, else one
 * per 32 frames; when all planes take the same kernel family — always the case for NV12 / YUV420 / planar surfaces allocated by this
 * library — one dispatch carries every plane).  A 720p plane is 2-3 us of GPU work, the same order as a kernel boundary: per-frame, per-plane dispatch leaves the chip
 * idle most of the time.  `frames` is a HOST array consumed before return.  vpf_resize is this call with n = 1. */
VPF_API vpf_status vpf_resize_batch(const vpf_exec* exec, int fmt, int interp, vpf_size src_size, vpf_size dst_size, uint32_t n,
                                    const vpf_frame_io* frames);

/* Scratch memory for the resize filters that keep per-shape operand tables (today: LANCZOS3 on 8-bit surfaces — column / row weight
 * operands of the matrix-core kernel, built once per (source size, destination size) by two small kernels and read by every later call).
 * NPP's pattern for this is a caller-provided buffer (nppiResizeGetBufferSize-style calls next to the NppResize*_Impl classes'
 * own destination surface, Tasks.cpp:1134-1150); here:
 *   - the caller owns `ptr` (device memory, 256-B aligned, `bytes` long) AND this struct: zero `opaque` once, then pass the same struct
 *     with every call that should share the tables.  The library records in `opaque` which tables the region holds.
 *   - a workspace is used on ONE stream at a time (its builds and its launches are ordered by that stream); a call on another stream, or
 *     while the stream is being captured, or with a shape the region does not hold, rebuilds (a few microseconds) — never an error.
 *   - too small a region (or ws == NULL, or the plain vpf_resize / vpf_resize_batch): the library's own small static arena is used
 *     instead (4 MiB per device, least-recently-used eviction); shapes that fit neither evaluate their weights inside the kernel.
 * Pixels are identical on every path.  Freeing `ptr` is the caller's business, after the stream has finished with it. */
typedef struct vpf_workspace {
  void* ptr;
  uint64_t bytes;
  uint64_t opaque[40];
} vpf_workspace;

/* Bytes of workspace that hold the tables of this resize for any batch size (0: this (fmt, interp, sizes) keeps no tables).
 * Pure host logic; callable without a GPU. */
VPF_API uint64_t vpf_resize_workspace_bytes(int fmt, int interp, vpf_size src_size, vpf_size dst_size);

/* vpf_resize / vpf_resize_batch with a caller-owned workspace (may be NULL). */
VPF_API vpf_status vpf_resize_ws(const vpf_exec* exec, int fmt, int interp, vpf_size src_size, const vpf_plane src[3], vpf_size dst_size,
                                 const vpf_plane dst[3], vpf_workspace* ws);
VPF_API vpf_status vpf_resize_batch_ws(const vpf_exec* exec, int fmt, int interp, vpf_size src_size, vpf_size dst_size, uint32_t n,
                                       const vpf_frame_io* frames, vpf_workspace* ws);

/* Per-pixel remap with bilinear sampling, packed RGB/BGR only (Tasks.cpp:1555-1602,
 * nppiRemap_8u_C3R + NPPI_INTER_LINEAR).  xmap/ymap are device pointers to float32 rows of
 * dst_size.width entries, row pitch in bytes.  Destination pixels whose source coordinate lies
 * outside [0,W-1]x[0,H-1] are left untouched. */
VPF_API vpf_status vpf_remap(const vpf_exec* exec, int fmt, vpf_size src_size, const vpf_plane* src,
                             const float* xmap, uint32_t xmap_pitch, const float* ymap,
                             uint32_t ymap_pitch, vpf_size dst_size, const vpf_plane* dst);

/* One pair of maps applied to `n` independent same-shape frames in one dispatch per 32 frames (a camera-undistortion map is the
 * same for every frame of a stream; frames after the first find the maps in the Infinity Cache).  frames[i].src[0] / dst[0] only. */
VPF_API vpf_status vpf_remap_batch(const vpf_exec* exec, int fmt, vpf_size src_size, const float* xmap, uint32_t xmap_pitch, const float* ymap,
                                   uint32_t ymap_pitch, vpf_size dst_size, uint32_t n, const vpf_frame_io* frames);

/* Fused NV12 -> bilinear resize -> packed RGB/BGR / RGB_PLANAR in one pass: reads only the source
 * texels it needs (BASELINE.md config 3 "fused").  Result is defined as: convert every NV12 texel
 * with vpf_convert's arithmetic, then vpf_resize(LINEAR) of the RGB image. */
VPF_API vpf_status vpf_convert_resize(const vpf_exec* exec, int src_fmt, int dst_fmt,
                                      int color_space, int color_range, vpf_size src_size,
                                      const vpf_plane src[3], vpf_size dst_size,
                                      const vpf_plane dst[3]);

/* The same over `n` independent same-shape frames in as few dispatches as possible (one per 128 frames when a frame moves at most
 * 7 000 000 bytes, source + destination — up to 1080p -> 720p —, one per 64 frames up to 10 000 000 bytes — 720p -> 1080p —, else one per 32 frames): a 720p output is ~2 us of HBM time, far below
 * a kernel boundary, so per-frame dispatch leaves the GPU mostly idle. */
VPF_API vpf_status vpf_convert_resize_batch(const vpf_exec* exec, int src_fmt, int dst_fmt, int color_space,
                                            int color_range, vpf_size src_size, vpf_size dst_size, uint32_t n,
                                            const vpf_frame_io* frames);

VPF_API const char* vpf_status_string(int status);
VPF_API const char* vpf_version(void);
/* hipGetDeviceCount; 0 when no GPU / no driver (never fails). Replaces GetNumGpus
 * (src/PyNvCodec/src/PyNvCodec.cpp:427-429). */
VPF_API int vpf_device_count(void);

/* Tracing (additive): with VPF_HIP_ROCTX=1 in the environment every entry point above runs inside a roctx range of its own name
 * and every kernel selection leaves a roctx marker, visible in `rocprofv3 --marker-trace`; VPF_HIP_LOG=2 prints the selected kernel
 * of every launch on stderr (=1: errors only).  Both replace the reference's compile-time NvtxMark (src/TC/inc/Tasks.hpp:27-52).
 * vpf_trace_push / vpf_trace_pop let a caller (the Task layer) open ranges of its own through the same switch. */
VPF_API int vpf_trace_push(const char* name);
VPF_API void vpf_trace_pop(int opened);

/* Tuning hook used by tests / benchmarks to select a kernel family (process-wide; 0 = default policy).  A hint, never a
 * correctness switch: every accepted value produces identical pixels and silently falls back where it does not apply.
 *   0        default policy (fastest applicable kernel per call)
 *   9        the any-size / any-alignment generic kernels everywhere (byte accesses, gather resize / remap)
 *   40       the narrower fast paths instead of the 16-px "r16" / tiled / quad kernels (A/B runs, test coverage)
 *   43       resize: the tiled separable kernel for bilinear down-scales as well (default: up-scales only)
 *   48       fused convert + resize: the workgroup-shared strip also beyond ~2x down-scales, where the policy takes the per-tap kernel
 *   49       fused convert + resize: the per-tap kernel's row-band form (four rows per wave) at every general factor and launch size
 *            (policy: beyond ~2x, on launches of >= 2048 workgroups)
 *   4, 8, 12, 30, 37, 44, 45, 46   one named NV12 / YUV420 -> RGB kernel of the default policy's set (k_yuv2rgb.hip launch_420)
 * Any other value is rejected: -1 is returned and nothing changes.  This library holds the kernels some policy path can select, nothing
 * else: the experimental kernels and bandwidth probes of round 1 live in tools/lab/libvpfhip_lab.so, and the kernel FORMS that were
 * measured and lost — the fused kernel's per-wave strips (variant 47), the two-role Lanczos form (VPF_TUNE_RESIZE_MFMA | 0x20000), the
 * persistent launch of the band kernels (VPF_TUNE_RESIZE_BAND | 0x10000 [| 0x40000 | 0x80000]) — in tools/lab/libvpfhip_forms.so, a build of
 * these same sources with -DVPF_LAB_FORMS that accepts those values (tools/lab/build_lab.py).  Not part of the reference surface.
 * Returns the previous value. */
VPF_API int vpf_set_tuning(int key, int value);
#define VPF_TUNE_NV12_RGB_VARIANT 1
#define VPF_TUNE_RESIZE_TILE 2 /* shape of the tiled resize kernels for measurement sweeps: 0 = policy, else rows-per-tile | waves-per-workgroup << 8
                                  (rows 4..64 in steps of 4, waves 4 or 8); same pixels whatever the shape */
#define VPF_TUNE_RESIZE_MFMA 5 /* 8-bit Lanczos-3 on the matrix cores (k_lanczos_mfma.hip): 0 = policy, 1 = never (the tiled / gather kernels take Lanczos), else
                                  N-tiles per wave (0 = policy, 4 or 8) << 8 | destination 16-row tiles per band (0 = policy, 1..64); | 0x10000: the kernel
                                  evaluates its filter weights itself instead of loading the per-shape tables (the path taken when no table fits);
                                  | 0x40000: one small plane per dispatch takes the matrix-core kernel too (the policy sends it to
                                  the tile kernel, whose single-launch latency is lower); | 0x80000: up-scales march with the ring of four source tiles like
                                  everything else (policy: a ring of two, one K chunk in pass 2); same pixels whatever the value */
#define VPF_TUNE_RESIZE_BAND 3 /* destination rows per wave of the row-pair bilinear kernels: 0 = policy, 1, 2, 4, 8 or 16; 4 | nb << 8 (nb = 1..8): the march form
                                  (nb 4-row bands per wave, 8 pixels per lane on 1-channel planes) where it applies; | 0x20000: 8 pixels per lane on every 1-channel
                                  plane of a band launch however well 512-column chunks fill its rows (policy: only at >= 80 % fill); same pixels
                                  whatever the value */

#ifdef __cplusplus
}
#endif
#endif /* VPF_HIP_H_ */
