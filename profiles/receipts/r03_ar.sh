#!/bin/bash
# round 3, visit ar: whole GPU suite + Lanczos / bilinear tables + sample chain on the final code
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r03ar_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r03ar_pytest_gpu.txt
VPF_BENCH_ONE=1 timeout 400 python tools/resize_batch_bench.py 2>&1 | grep -E "resize_batch|remap" > gpurun_out/r03ar_bench.txt; cut -c1-230 gpurun_out/r03ar_bench.txt
timeout 300 python tools/chain_bench.py 2>&1 | grep chain | cut -c1-330 | tee gpurun_out/r03ar_chain.txt
