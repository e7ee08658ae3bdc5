#!/bin/bash
# round 3, visit bj: non-temporal host copies between caller frames and the pinned staging buffers — parity of the up / download paths, A/B of both directions
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pynvcodec.py -m gpu -x -q 2>&1 | tail -2
{ for i in 1 2; do VPF_HIP_NT_COPY=0 timeout 300 python tools/download_bench.py 2>&1 | grep "pageable" | sed 's/$/ NT_COPY=0/'; timeout 300 python tools/download_bench.py 2>&1 | grep "pageable" | sed 's/$/ NT_COPY=1/'; done
  for i in 1 2; do VPF_HIP_NT_COPY=0 timeout 300 python tools/pipeline_bench.py 2>&1 | grep "1 thread" | sed 's/$/ NT_COPY=0/'; timeout 300 python tools/pipeline_bench.py 2>&1 | grep "1 thread" | sed 's/$/ NT_COPY=1/'; done; } | tee gpurun_out/r03_host_copy_nt_ab.txt
