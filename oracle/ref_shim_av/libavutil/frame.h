/* TEST INFRASTRUCTURE: see pixfmt.h in this directory */
#ifndef VPF_REF_SHIM_AV_FRAME_H_
#define VPF_REF_SHIM_AV_FRAME_H_
#include "pixfmt.h"
enum AVFrameSideDataType { AV_FRAME_DATA_PANSCAN = 0, AV_FRAME_DATA_MOTION_VECTORS = 8 };
#endif
