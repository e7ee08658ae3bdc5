#pragma once
#include <errno.h>
#include <stddef.h>
#define AVERROR(e) (-(e))
#define AVERROR_EOF (-541478725)
int av_strerror(int errnum, char* errbuf, size_t errbuf_size);
