/* oracle/ref_shim/libavutil/frame.h — TEST INFRASTRUCTURE: the reference's Tasks.hpp includes <libavutil/frame.h> for
 * pointer-typed members only; an opaque declaration is all the converter path needs. */
#pragma once
typedef struct AVFrame AVFrame;
typedef struct AVFrameSideData AVFrameSideData;
typedef struct AVDictionary AVDictionary;
enum AVFrameSideDataType { AV_FRAME_DATA_PANSCAN = 0 };
