/* TEST INFRASTRUCTURE: see ../libavutil/pixfmt.h */
#ifndef VPF_REF_SHIM_AV_AVCODEC_H_
#define VPF_REF_SHIM_AV_AVCODEC_H_
#include <stdint.h>
#include "../libavutil/frame.h"
#include "../libavutil/pixfmt.h"
enum AVCodecID { AV_CODEC_ID_NONE = 0, AV_CODEC_ID_MPEG1VIDEO, AV_CODEC_ID_MPEG2VIDEO, AV_CODEC_ID_MJPEG, AV_CODEC_ID_MPEG4, AV_CODEC_ID_H264,
                 AV_CODEC_ID_VC1, AV_CODEC_ID_VP8, AV_CODEC_ID_VP9, AV_CODEC_ID_HEVC, AV_CODEC_ID_AV1 };
typedef struct AVBSFContext AVBSFContext;
typedef struct AVPacket { uint8_t* data; int size; int64_t pts, dts, duration, pos; int stream_index, flags; } AVPacket;
#endif
