/* tests/libav_stub/stub_libav.c — TEST INFRASTRUCTURE.  A stand-in IMPLEMENTATION of the dozen libav* entry points that
 * videoprocessingframework_amd/csrc/feeder/FfmpegFeeder.cpp uses (declared in the stub headers next to this file), so that the
 * feeder — open, stream selection, the send / receive loop with decoder delay, the YUV420P -> NV12 repack, end of stream and every
 * error path — EXECUTES in an image that has no FFmpeg.  Not FFmpeg code; written from the public API's documented behaviour.
 *
 * The "container" is described by the URL:  synth:key=value,key=value,...
 *   w, h        picture size (default 64 x 32)          n       frames in the clip (default 5)
 *   fmt         AVPixelFormat of decoded frames: 0 YUV420P (default), 12 YUVJ420P, 23 NV12, anything else = unsupported by the feeder
 *   seed        content seed                            delay   frames the decoder holds back before the first output (default 2)
 *   cs, cr      AVColorSpace / AVColorRange tags of the stream (default 1 = BT709, 1 = MPEG)
 *   novideo=1   the container has only an audio stream  nocodec=1  no decoder for the stream's codec
 *   fail_at=k   avcodec_receive_frame fails with AVERROR(EINVAL) at output frame k
 *   change_at=k from output frame k on, pictures are (w2 x h2) (default 2w x 2h): a mid-stream resolution change
 *   audio=m     every m-th packet read belongs to stream 1 (audio) and must be skipped by the caller
 * Frame i, plane p, pixel (x, y) = (a_p x + b_p y + c_p i + seed d_p) mod 256 with (a,b,c,d) = Y (3,5,7,1) U (1,2,11,3) V (3,1,13,5);
 * NV12 frames interleave the same U and V.  Rows are padded (linesize = width rounded up to 32, plus 32) and the padding is 0xEE.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libavformat/avformat.h"
#include "libavutil/error.h"

struct AVDictionary { int n; };
struct AVCodec { int id; };
struct AVInputFormat { int unused; };

typedef struct {
  int w, h, n, fmt, seed, delay, cs, cr, novideo, nocodec, fail_at, change_at, w2, h2, audio;
  int packets_read, video_sent, frames_out, flushing;
} Clip;

/* the public structs carry the clip behind their declared members */
typedef struct { AVFormatContext pub; AVStream st0, st1; AVStream* list[2]; AVCodecParameters par0, par1; Clip clip; } FormatImpl;
typedef struct { AVCodecContext pub; Clip* clip; int queued; } CodecImpl;
typedef struct { AVFrame pub; uint8_t* buf[3]; } FrameImpl;

static Clip* g_last_clip; /* avcodec_parameters_to_context has no pointer back to the container in this tiny API surface */

static int geti(const char* url, const char* key, int def) {
  const size_t k = strlen(key);
  for (const char* p = url; (p = strstr(p, key)) != NULL; p += k)
    if ((p == url || p[-1] == ',' || p[-1] == ':') && p[k] == '=') return atoi(p + k + 1);
  return def;
}

unsigned avformat_version(void) { return (60u << 16) | 0x5700u; } /* "stub" */
unsigned avcodec_version(void) { return (60u << 16) | 0x5700u; }
int av_strerror(int errnum, char* buf, size_t n) {
  if (errnum == AVERROR_EOF) snprintf(buf, n, "End of file");
  else if (errnum == AVERROR(ENOENT)) snprintf(buf, n, "No such file or directory");
  else if (errnum == AVERROR(EINVAL)) snprintf(buf, n, "Invalid data found when processing input");
  else return -1;
  return 0;
}
int av_dict_set(AVDictionary** pm, const char* key, const char* value, int flags) {
  (void)key; (void)value; (void)flags;
  if (!*pm) *pm = (AVDictionary*)calloc(1, sizeof(AVDictionary));
  (*pm)->n++;
  return 0;
}
void av_dict_free(AVDictionary** m) { if (m && *m) { free(*m); *m = NULL; } }

int avformat_open_input(AVFormatContext** ps, const char* url, const AVInputFormat* fmt, AVDictionary** options) {
  (void)fmt; (void)options;
  if (!url || strncmp(url, "synth:", 6) != 0) return AVERROR(ENOENT);
  FormatImpl* f = (FormatImpl*)calloc(1, sizeof(FormatImpl));
  Clip* c = &f->clip;
  c->w = geti(url, "w", 64); c->h = geti(url, "h", 32); c->n = geti(url, "n", 5); c->fmt = geti(url, "fmt", 0); c->seed = geti(url, "seed", 0);
  c->delay = geti(url, "delay", 2); c->cs = geti(url, "cs", 1); c->cr = geti(url, "cr", 1); c->novideo = geti(url, "novideo", 0);
  c->nocodec = geti(url, "nocodec", 0); c->fail_at = geti(url, "fail_at", -1); c->change_at = geti(url, "change_at", -1);
  c->w2 = geti(url, "w2", 2 * c->w); c->h2 = geti(url, "h2", 2 * c->h); c->audio = geti(url, "audio", 3);
  f->par0.codec_id = c->nocodec ? AV_CODEC_ID_NONE : AV_CODEC_ID_H264;
  f->par1.codec_id = AV_CODEC_ID_NONE;
  f->st0.codecpar = &f->par0; f->st0.avg_frame_rate.num = 30000; f->st0.avg_frame_rate.den = 1001;
  f->st1.codecpar = &f->par1;
  f->list[0] = &f->st0; f->list[1] = &f->st1;
  f->pub.nb_streams = 2; f->pub.streams = f->list;
  *ps = &f->pub;
  g_last_clip = c;
  return 0;
}
int avformat_find_stream_info(AVFormatContext* ic, AVDictionary** options) { (void)options; return ic ? 0 : AVERROR(EINVAL); }
int av_find_best_stream(AVFormatContext* ic, enum AVMediaType type, int wanted, int related, const AVCodec** dec, int flags) {
  (void)wanted; (void)related; (void)dec; (void)flags;
  FormatImpl* f = (FormatImpl*)ic;
  if (type != AVMEDIA_TYPE_VIDEO || f->clip.novideo) return AVERROR(ENOENT); /* real libav: AVERROR_STREAM_NOT_FOUND, also negative */
  return 0;
}
void avformat_close_input(AVFormatContext** s) { if (s && *s) { free(*s); *s = NULL; } }
int av_read_frame(AVFormatContext* s, AVPacket* pkt) {
  Clip* c = &((FormatImpl*)s)->clip;
  if (c->video_sent >= c->n) return AVERROR_EOF;
  c->packets_read++;
  if (c->audio > 0 && c->packets_read % c->audio == 0) { pkt->stream_index = 1; return 0; }
  pkt->stream_index = 0;
  c->video_sent++;
  return 0;
}

static const AVCodec g_h264 = {27};
const AVCodec* avcodec_find_decoder(enum AVCodecID id) { return id == AV_CODEC_ID_H264 ? &g_h264 : NULL; }
AVCodecContext* avcodec_alloc_context3(const AVCodec* codec) { (void)codec; return (AVCodecContext*)calloc(1, sizeof(CodecImpl)); }
int avcodec_parameters_to_context(AVCodecContext* ctx, const AVCodecParameters* par) {
  (void)par;
  CodecImpl* d = (CodecImpl*)ctx;
  d->clip = g_last_clip;
  ctx->width = d->clip->w; ctx->height = d->clip->h;
  ctx->colorspace = (enum AVColorSpace)d->clip->cs; ctx->color_range = (enum AVColorRange)d->clip->cr;
  return 0;
}
int avcodec_open2(AVCodecContext* ctx, const AVCodec* codec, AVDictionary** options) { (void)options; return (ctx && codec) ? 0 : AVERROR(EINVAL); }
void avcodec_free_context(AVCodecContext** ctx) { if (ctx && *ctx) { free(*ctx); *ctx = NULL; } }
int avcodec_send_packet(AVCodecContext* ctx, const AVPacket* pkt) {
  CodecImpl* d = (CodecImpl*)ctx;
  if (!pkt) { d->clip->flushing = 1; return 0; }
  d->queued++;
  return 0;
}

AVPacket* av_packet_alloc(void) { return (AVPacket*)calloc(1, sizeof(AVPacket)); }
void av_packet_free(AVPacket** p) { if (p && *p) { free(*p); *p = NULL; } }
void av_packet_unref(AVPacket* p) { if (p) p->stream_index = -1; }
AVFrame* av_frame_alloc(void) { return (AVFrame*)calloc(1, sizeof(FrameImpl)); }
void av_frame_unref(AVFrame* f) {
  FrameImpl* fi = (FrameImpl*)f;
  for (int k = 0; k < 3; k++) { free(fi->buf[k]); fi->buf[k] = NULL; f->data[k] = NULL; f->linesize[k] = 0; }
  f->width = f->height = 0;
}
void av_frame_free(AVFrame** f) { if (f && *f) { av_frame_unref(*f); free(*f); *f = NULL; } }

static const int kA[3] = {3, 1, 3}, kB[3] = {5, 2, 1}, kC[3] = {7, 11, 13}, kD[3] = {1, 3, 5};
static uint8_t sample(int p, int x, int y, int i, int seed) { return (uint8_t)((kA[p] * x + kB[p] * y + kC[p] * i + seed * kD[p]) & 0xff); }

int avcodec_receive_frame(AVCodecContext* ctx, AVFrame* frame) {
  CodecImpl* d = (CodecImpl*)ctx;
  Clip* c = d->clip;
  FrameImpl* fi = (FrameImpl*)frame;
  const int avail = d->queued - c->frames_out;
  if (avail <= 0) return c->flushing ? AVERROR_EOF : AVERROR(EAGAIN);
  if (!c->flushing && avail <= c->delay) return AVERROR(EAGAIN); /* reorder delay: output lags the input */
  const int i = c->frames_out;
  if (i == c->fail_at) return AVERROR(EINVAL);
  const int big = c->change_at >= 0 && i >= c->change_at;
  const int w = big ? c->w2 : c->w, h = big ? c->h2 : c->h, cw = (w + 1) / 2, ch = (h + 1) / 2;
  av_frame_unref(frame);
  ctx->width = w; ctx->height = h; /* libav updates the context when the stream's size changes */
  frame->width = w; frame->height = h; frame->format = c->fmt;
  const int nv12 = c->fmt == AV_PIX_FMT_NV12;
  const int pw[3] = {w, nv12 ? 2 * cw : cw, cw}, ph[3] = {h, ch, ch};
  for (int p = 0; p < (nv12 ? 2 : 3); p++) {
    const int ls = (pw[p] + 31) / 32 * 32 + 32;
    fi->buf[p] = (uint8_t*)malloc((size_t)ls * ph[p]);
    memset(fi->buf[p], 0xEE, (size_t)ls * ph[p]);
    frame->data[p] = fi->buf[p]; frame->linesize[p] = ls;
    for (int y = 0; y < ph[p]; y++)
      for (int x = 0; x < pw[p]; x++)
        fi->buf[p][(size_t)y * ls + x] = (nv12 && p == 1) ? sample(1 + (x & 1), x >> 1, y, i, c->seed) : sample(p, x, y, i, c->seed);
  }
  c->frames_out++;
  return 0;
}
