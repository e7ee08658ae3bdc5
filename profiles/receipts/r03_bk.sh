#!/bin/bash
# round 3, visit bk: download A/B of the non-temporal host copy, five alternating rounds; uploader threads
mkdir -p gpurun_out
{ for i in 1 2 3 4 5; do VPF_HIP_NT_COPY=0 timeout 300 python tools/download_bench.py 3840 2160 300 2>&1 | grep "pageable" | sed 's/$/ NT_COPY=0/'; timeout 300 python tools/download_bench.py 3840 2160 300 2>&1 | grep "pageable" | sed 's/$/ NT_COPY=1/'; done; } | tee gpurun_out/r03_host_copy_nt_download_ab.txt
nproc; lscpu | grep -i "numa\|model name" | head -6
