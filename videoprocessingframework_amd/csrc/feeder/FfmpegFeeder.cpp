// FfmpegFeeder.cpp — see FfmpegFeeder.hpp.  libav usage follows the public decode example flow
// (avformat_open_input -> avformat_find_stream_info -> av_find_best_stream -> avcodec_open2 -> av_read_frame ->
// avcodec_send_packet / avcodec_receive_frame), the same calls the reference makes (FfmpegSwDecoder.cpp:72-135,264,334-342).
#include "FfmpegFeeder.hpp"

#include <cstring>
#include <sstream>
#include <stdexcept>

extern "C" {
#include <libavcodec/avcodec.h>
#include <libavformat/avformat.h>
#include <libavutil/dict.h>
#include <libavutil/error.h>
#include <libavutil/pixdesc.h>
#include <libavutil/pixfmt.h>
}

namespace VPF {

namespace {
std::string av_err(int code) {
  char buf[256] = {0};
  if (av_strerror(code, buf, sizeof(buf) - 1) != 0) {
    std::stringstream ss;
    ss << "Unknown libav error " << code;
    return ss.str();
  }
  return std::string(buf);
}
}  // namespace

struct FfmpegFeeder::Impl {
  AVFormatContext* fmt = nullptr;
  AVCodecContext* dec = nullptr;
  AVFrame* frame = nullptr;
  AVPacket* pkt = nullptr;
  int stream = -1;
  bool draining = false, done = false, have_frame = false;

  ~Impl() {
    if (pkt) av_packet_free(&pkt);
    if (frame) av_frame_free(&frame);
    if (dec) avcodec_free_context(&dec);
    if (fmt) avformat_close_input(&fmt);
  }

  // one decoded picture -> tight NV12.  Planar 4:2:0 (YUV420P / YUVJ420P) gets its chroma interleaved; NV12 is copied.
  // Sizes come from the FRAME (a stream may change resolution mid-way; the codec context's width / height are only what the
  // container announced): the caller checks frame_bytes() against its buffer before calling this.
  size_t frame_bytes() const {
    const size_t w = (size_t)frame->width, h = (size_t)frame->height;
    return w * h + 2 * ((w + 1) / 2) * ((h + 1) / 2);
  }
  bool to_nv12(uint8_t* out) const {
    const int w = frame->width, h = frame->height, cw = (w + 1) / 2, ch = (h + 1) / 2;
    for (int y = 0; y < h; y++) std::memcpy(out + (size_t)y * w, frame->data[0] + (size_t)y * frame->linesize[0], (size_t)w);
    uint8_t* uv = out + (size_t)w * h;
    switch ((AVPixelFormat)frame->format) {
      case AV_PIX_FMT_YUV420P:
      case AV_PIX_FMT_YUVJ420P:
        for (int y = 0; y < ch; y++) {
          const uint8_t* u = frame->data[1] + (size_t)y * frame->linesize[1];
          const uint8_t* v = frame->data[2] + (size_t)y * frame->linesize[2];
          uint8_t* o = uv + (size_t)y * 2 * cw;
          for (int x = 0; x < cw; x++) { o[2 * x] = u[x]; o[2 * x + 1] = v[x]; }
        }
        return true;
      case AV_PIX_FMT_NV12:
        for (int y = 0; y < ch; y++) std::memcpy(uv + (size_t)y * 2 * cw, frame->data[1] + (size_t)y * frame->linesize[1], (size_t)2 * cw);
        return true;
      default:
        return false;  // 4:2:2 / 4:4:4 / high bit depth: not a feeder for the NV12 path
    }
  }
};

FfmpegFeeder::FfmpegFeeder(const std::string& url, const std::map<std::string, std::string>& options) : p(new Impl) {
  AVDictionary* opts = nullptr;
  for (const auto& kv : options) av_dict_set(&opts, kv.first.c_str(), kv.second.c_str(), 0);
  int res = avformat_open_input(&p->fmt, url.c_str(), nullptr, &opts);
  if (res < 0) {
    av_dict_free(&opts);
    throw std::runtime_error("FfmpegFeeder: can't open " + url + ": " + av_err(res));
  }
  res = avformat_find_stream_info(p->fmt, nullptr);
  if (res < 0) {
    av_dict_free(&opts);
    throw std::runtime_error("FfmpegFeeder: can't find stream information: " + av_err(res));
  }
  p->stream = av_find_best_stream(p->fmt, AVMEDIA_TYPE_VIDEO, -1, -1, nullptr, 0);
  if (p->stream < 0) {
    av_dict_free(&opts);
    throw std::runtime_error("FfmpegFeeder: no video stream in " + url);
  }
  AVStream* st = p->fmt->streams[p->stream];
  const AVCodec* codec = avcodec_find_decoder(st->codecpar->codec_id);
  if (!codec) {
    av_dict_free(&opts);
    throw std::runtime_error("FfmpegFeeder: no software decoder for this codec");
  }
  p->dec = avcodec_alloc_context3(codec);
  if (!p->dec || avcodec_parameters_to_context(p->dec, st->codecpar) < 0) {
    av_dict_free(&opts);
    throw std::runtime_error("FfmpegFeeder: can't set up the codec context");
  }
  res = avcodec_open2(p->dec, codec, &opts);
  av_dict_free(&opts);
  if (res < 0) throw std::runtime_error("FfmpegFeeder: can't open the codec: " + av_err(res));
  p->frame = av_frame_alloc();
  p->pkt = av_packet_alloc();
  if (!p->frame || !p->pkt) throw std::runtime_error("FfmpegFeeder: out of memory");
}
FfmpegFeeder::~FfmpegFeeder() = default;

uint32_t FfmpegFeeder::Width() const { return (uint32_t)p->dec->width; }
uint32_t FfmpegFeeder::Height() const { return (uint32_t)p->dec->height; }
double FfmpegFeeder::Framerate() const {
  const AVRational r = p->fmt->streams[p->stream]->avg_frame_rate;
  return r.den ? (double)r.num / (double)r.den : 0.0;
}
ColorSpace FfmpegFeeder::GetColorSpace() const {
  switch (p->dec->colorspace) {  // same mapping as the reference (FfmpegSwDecoder.cpp:437-451)
    case AVCOL_SPC_BT709: return BT_709;
    case AVCOL_SPC_BT470BG:
    case AVCOL_SPC_SMPTE170M: return BT_601;
    default: return UNSPEC;
  }
}
ColorRange FfmpegFeeder::GetColorRange() const {
  switch (p->dec->color_range) {  // (FfmpegSwDecoder.cpp:453-465)
    case AVCOL_RANGE_MPEG: return MPEG;
    case AVCOL_RANGE_JPEG: return JPEG;
    default: return UDEF;
  }
}

bool FfmpegFeeder::NextFrame() {
  if (p->have_frame) { av_frame_unref(p->frame); p->have_frame = false; }
  if (p->done) return false;
  for (;;) {
    const int got = avcodec_receive_frame(p->dec, p->frame);
    if (got == 0) {
      if (p->frame->width <= 0 || p->frame->height <= 0) {
        av_frame_unref(p->frame);
        throw std::runtime_error("FfmpegFeeder: decoder returned an empty picture");
      }
      p->have_frame = true;
      return true;
    }
    if (got == AVERROR_EOF) { p->done = true; return false; }
    if (got != AVERROR(EAGAIN)) throw std::runtime_error("FfmpegFeeder: decode error: " + av_err(got));
    // the decoder wants more input
    if (p->draining) { p->done = true; return false; }
    int rd;
    while ((rd = av_read_frame(p->fmt, p->pkt)) >= 0 && p->pkt->stream_index != p->stream) av_packet_unref(p->pkt);
    if (rd < 0) {  // end of file: flush
      p->draining = true;
      avcodec_send_packet(p->dec, nullptr);
      continue;
    }
    const int sent = avcodec_send_packet(p->dec, p->pkt);
    av_packet_unref(p->pkt);
    if (sent < 0 && sent != AVERROR(EAGAIN)) throw std::runtime_error("FfmpegFeeder: send_packet failed: " + av_err(sent));
  }
}
uint32_t FfmpegFeeder::FrameWidth() const { return p->have_frame ? (uint32_t)p->frame->width : 0; }
uint32_t FfmpegFeeder::FrameHeight() const { return p->have_frame ? (uint32_t)p->frame->height : 0; }
size_t FfmpegFeeder::PendingFrameBytes() const { return p->have_frame ? p->frame_bytes() : 0; }

bool FfmpegFeeder::CopyFrameNV12(uint8_t* nv12, size_t capacity) {
  if (!p->have_frame || !nv12) return false;
  if (capacity < p->frame_bytes()) {  // never write past the caller's buffer (a mid-stream resolution change lands here)
    std::stringstream ss;
    ss << "FfmpegFeeder: decoded frame is " << p->frame->width << "x" << p->frame->height << " (" << p->frame_bytes() << " B of NV12) but the destination holds "
       << capacity << " B (stream announced " << Width() << "x" << Height() << ")";
    av_frame_unref(p->frame);
    p->have_frame = false;
    throw std::runtime_error(ss.str());
  }
  const bool ok = p->to_nv12(nv12);
  av_frame_unref(p->frame);
  p->have_frame = false;
  if (!ok) throw std::runtime_error("FfmpegFeeder: decoded pixel format is not 8-bit 4:2:0");
  return true;
}

bool FfmpegFeeder::DecodeNextFrame(uint8_t* nv12, size_t capacity) {
  if (!nv12) return false;
  return NextFrame() && CopyFrameNV12(nv12, capacity);
}

}  // namespace VPF
