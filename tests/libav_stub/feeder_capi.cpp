// tests/libav_stub/feeder_capi.cpp — TEST INFRASTRUCTURE: a C view of VPF::FfmpegFeeder for ctypes (tests/test_feeder_stub_libav.py).
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <string>

#include "FfmpegFeeder.hpp"

using VPF::FfmpegFeeder;

static void put(char* err, int cap, const char* what) {
  if (err && cap > 0) std::snprintf(err, cap, "%s", what);
}

extern "C" {
void* feeder_open(const char* url, char* err, int cap) {
  try {
    return new FfmpegFeeder(url, {{"threads", "1"}});
  } catch (std::exception& e) {
    put(err, cap, e.what());
    return nullptr;
  }
}
void feeder_close(void* f) { delete static_cast<FfmpegFeeder*>(f); }
// out = {Width, Height, ColorSpace, ColorRange, PixelFormat, FrameBytes, framerate * 1000}
void feeder_info(void* f, long long out[7]) {
  auto* d = static_cast<FfmpegFeeder*>(f);
  out[0] = d->Width(); out[1] = d->Height(); out[2] = d->GetColorSpace(); out[3] = d->GetColorRange(); out[4] = d->GetPixelFormat();
  out[5] = (long long)d->FrameBytes(); out[6] = (long long)(d->Framerate() * 1000.0 + 0.5);
}
// 1 = a frame was written, 0 = end of stream, -1 = the feeder threw (message in err)
int feeder_decode(void* f, unsigned char* buf, size_t cap, char* err, int errcap) {
  try {
    return static_cast<FfmpegFeeder*>(f)->DecodeNextFrame(buf, cap) ? 1 : 0;
  } catch (std::exception& e) {
    put(err, errcap, e.what());
    return -1;
  }
}
// two-step form: dims = {FrameWidth, FrameHeight, PendingFrameBytes}
int feeder_next(void* f, long long dims[3], char* err, int errcap) {
  try {
    auto* d = static_cast<FfmpegFeeder*>(f);
    if (!d->NextFrame()) return 0;
    dims[0] = d->FrameWidth(); dims[1] = d->FrameHeight(); dims[2] = (long long)d->PendingFrameBytes();
    return 1;
  } catch (std::exception& e) {
    put(err, errcap, e.what());
    return -1;
  }
}
int feeder_copy(void* f, unsigned char* buf, size_t cap, char* err, int errcap) {
  try {
    return static_cast<FfmpegFeeder*>(f)->CopyFrameNV12(buf, cap) ? 1 : 0;
  } catch (std::exception& e) {
    put(err, errcap, e.what());
    return -1;
  }
}
}
