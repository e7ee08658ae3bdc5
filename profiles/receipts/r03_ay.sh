#!/bin/bash
# pipelined pageable download: parity test + rate with and without the completion flag
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pynvcodec.py -m gpu -x -q 2>&1 | tail -5
{ timeout 300 python tools/download_bench.py; VPF_HIP_SYNC_SPIN_US=0 timeout 300 python tools/download_bench.py; timeout 300 python tools/download_bench.py 1920 1080 1000; VPF_HIP_SYNC_SPIN_US=0 timeout 300 python tools/download_bench.py 1920 1080 1000; } 2>&1 | grep "\[download\]" | tee gpurun_out/r03_download_pipelined.txt
