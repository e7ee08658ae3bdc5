/* TEST INFRASTRUCTURE: see ../libavutil/pixfmt.h */
#ifndef VPF_REF_SHIM_AV_BSF_H_
#define VPF_REF_SHIM_AV_BSF_H_
typedef struct AVBSFContext AVBSFContext;
#endif
