#pragma once
#include "../libavcodec/avcodec.h"
enum AVMediaType { AVMEDIA_TYPE_UNKNOWN = -1, AVMEDIA_TYPE_VIDEO = 0 };
typedef struct AVRational { int num, den; } AVRational;
typedef struct AVInputFormat AVInputFormat;
typedef struct AVStream { AVCodecParameters* codecpar; AVRational avg_frame_rate; } AVStream;
typedef struct AVFormatContext { unsigned nb_streams; AVStream** streams; } AVFormatContext;
int avformat_open_input(AVFormatContext** ps, const char* url, const AVInputFormat* fmt, AVDictionary** options);
int avformat_find_stream_info(AVFormatContext* ic, AVDictionary** options);
int av_find_best_stream(AVFormatContext* ic, enum AVMediaType type, int wanted, int related, const AVCodec** decoder_ret, int flags);
int av_read_frame(AVFormatContext* s, AVPacket* pkt);
void avformat_close_input(AVFormatContext** s);
unsigned avformat_version(void);
