/* TEST INFRASTRUCTURE: see pixfmt.h in this directory */
#ifndef VPF_REF_SHIM_AV_PIXDESC_H_
#define VPF_REF_SHIM_AV_PIXDESC_H_
#include "pixfmt.h"
#ifdef __cplusplus
extern "C" {
#endif
const char* av_get_pix_fmt_name(enum AVPixelFormat pix_fmt);
#ifdef __cplusplus
}
#endif
#endif
