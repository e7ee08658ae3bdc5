/* TEST INFRASTRUCTURE: see ../libavutil/pixfmt.h */
#ifndef VPF_REF_SHIM_AV_AVFORMAT_H_
#define VPF_REF_SHIM_AV_AVFORMAT_H_
#include "../libavcodec/avcodec.h"
typedef struct AVFormatContext AVFormatContext;
#endif
