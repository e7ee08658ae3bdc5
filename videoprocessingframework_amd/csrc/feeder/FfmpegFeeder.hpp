// FfmpegFeeder.hpp — OPTIONAL host-side feeder: demux + software decode with FFmpeg's libav*, handing frames to the
// surface path as real NV12.  SURVEY §8(f) N3; reference: src/TC/src/FfmpegSwDecoder.cpp:60-170,254-360 (open, find the
// best video stream, send/receive loop, SaveYUV420 :141-168) and src/PyNvCodec/src/PyFFMpegDecoder.cpp:37-70.
//
// Built into the product ONLY where the libav headers and libraries exist (videoprocessingframework_amd/_build_bindings.py probes
// for them); this image has none, so here it is compiled and RUN against tests/libav_stub — a stand-in implementation of the dozen
// libav entry points that decodes synthetic clips (tests/test_feeder_stub_libav.py): the send / receive loop, the YUV420P -> NV12
// repack, end of stream and the error paths execute; real bitstreams do not.
// Not on the conversion hot path: decode stays on the host, as north_star prescribes.
//
// Deliberate difference from the reference: its decoder stores planar YUV420P and labels it NV12
// (FfmpegSwDecoder.cpp:405-410 vs :141-168).  This feeder emits actual NV12 (UV interleaved).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "MemoryInterfaces.hpp"

namespace VPF {

class FfmpegFeeder {
public:
  // throws std::runtime_error when the input cannot be opened / has no decodable video stream
  FfmpegFeeder(const std::string& url, const std::map<std::string, std::string>& options);
  ~FfmpegFeeder();
  FfmpegFeeder(const FfmpegFeeder&) = delete;
  FfmpegFeeder& operator=(const FfmpegFeeder&) = delete;

  uint32_t Width() const;
  uint32_t Height() const;
  double Framerate() const;
  ColorSpace GetColorSpace() const;   // from the stream's colorspace tag (BT.709 / BT.601 / UNSPEC)
  ColorRange GetColorRange() const;   // MPEG (limited) / JPEG (full) / UDEF
  Pixel_Format GetPixelFormat() const { return NV12; }
  // tight NV12 bytes of a Width() x Height() picture: luma + interleaved chroma of ceil(W/2) x ceil(H/2) samples (odd sizes round chroma up,
  // the same formula DecodeNextFrame checks its capacity against)
  size_t FrameBytes() const { return (size_t)Width() * Height() + 2 * (size_t)((Width() + 1) / 2) * ((Height() + 1) / 2); }

  // Decode the next frame into `nv12` (tight W x 1.5H bytes: Y plane then interleaved UV).  false at end of stream.  Throws when
  // the decoded frame does not fit `capacity` (sizes are taken from the frame itself, not from the container's announcement).
  bool DecodeNextFrame(uint8_t* nv12, size_t capacity);

  // The same in two steps, for callers that size their buffer per frame (streams may change resolution): NextFrame() decodes
  // and holds one picture (false at end of stream), FrameWidth / FrameHeight / PendingFrameBytes describe it, CopyFrameNV12
  // writes it out as tight NV12 and releases it.
  bool NextFrame();
  uint32_t FrameWidth() const;
  uint32_t FrameHeight() const;
  size_t PendingFrameBytes() const;
  bool CopyFrameNV12(uint8_t* nv12, size_t capacity);

private:
  struct Impl;
  std::unique_ptr<Impl> p;
};

}  // namespace VPF
