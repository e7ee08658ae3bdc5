/* TEST INFRASTRUCTURE: see ../libavutil/pixfmt.h */
#ifndef VPF_REF_SHIM_AV_AVIO_H_
#define VPF_REF_SHIM_AV_AVIO_H_
typedef struct AVIOContext AVIOContext;
#endif
