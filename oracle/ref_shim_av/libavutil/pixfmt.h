/* oracle/ref_shim_av — TEST INFRASTRUCTURE.  Stand-in declarations (our own, names only) for the handful of libav types and constants
 * the reference's Tasks.cpp / FFmpegDemuxer.h / NvCodecUtils.h mention, so that oracle/Makefile `ref_tc_hip` can compile the reference's
 * Tasks.cpp (ResizeSurface / RemapSurface) where it lies.  Nothing here decodes anything; the demux / decode / encode tasks of that file
 * link against abort stubs (oracle/ref_tasks_stubs.py) and are never called. */
#ifndef VPF_REF_SHIM_AV_PIXFMT_H_
#define VPF_REF_SHIM_AV_PIXFMT_H_
enum AVPixelFormat { AV_PIX_FMT_NONE = -1, AV_PIX_FMT_YUV420P, AV_PIX_FMT_YUV422P, AV_PIX_FMT_YUV444P, AV_PIX_FMT_YUVJ420P, AV_PIX_FMT_NV12,
                     AV_PIX_FMT_YUV420P10, AV_PIX_FMT_YUV420P12, AV_PIX_FMT_YUV444P10LE, AV_PIX_FMT_YUV444P16LE, AV_PIX_FMT_P016LE };
enum AVColorSpace { AVCOL_SPC_RGB = 0, AVCOL_SPC_BT709 = 1, AVCOL_SPC_UNSPECIFIED = 2, AVCOL_SPC_BT470BG = 5, AVCOL_SPC_SMPTE170M = 6 };
enum AVColorRange { AVCOL_RANGE_UNSPECIFIED = 0, AVCOL_RANGE_MPEG = 1, AVCOL_RANGE_JPEG = 2 };
#endif
