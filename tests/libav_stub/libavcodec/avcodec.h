#pragma once
#include <stdint.h>
#include "../libavutil/dict.h"
#include "../libavutil/pixfmt.h"
enum AVCodecID { AV_CODEC_ID_NONE = 0, AV_CODEC_ID_H264 = 27 };
typedef struct AVCodec AVCodec;
typedef struct AVCodecParameters { enum AVCodecID codec_id; } AVCodecParameters;
typedef struct AVCodecContext { int width, height; enum AVColorSpace colorspace; enum AVColorRange color_range; } AVCodecContext;
typedef struct AVPacket { int stream_index; } AVPacket;
typedef struct AVFrame { uint8_t* data[8]; int linesize[8]; int width, height; int format; } AVFrame;
const AVCodec* avcodec_find_decoder(enum AVCodecID id);
AVCodecContext* avcodec_alloc_context3(const AVCodec* codec);
int avcodec_parameters_to_context(AVCodecContext* codec, const AVCodecParameters* par);
int avcodec_open2(AVCodecContext* avctx, const AVCodec* codec, AVDictionary** options);
void avcodec_free_context(AVCodecContext** avctx);
int avcodec_send_packet(AVCodecContext* avctx, const AVPacket* avpkt);
int avcodec_receive_frame(AVCodecContext* avctx, AVFrame* frame);
unsigned avcodec_version(void);
AVPacket* av_packet_alloc(void);
void av_packet_free(AVPacket** pkt);
void av_packet_unref(AVPacket* pkt);
AVFrame* av_frame_alloc(void);
void av_frame_free(AVFrame** frame);
void av_frame_unref(AVFrame* frame);
