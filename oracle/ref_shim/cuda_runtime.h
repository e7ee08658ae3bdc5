#pragma once
#include "cuda.h"
typedef CUstream cudaStream_t; /* NvCodecUtils.h mentions the runtime-API name */
