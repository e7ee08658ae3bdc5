#!/bin/bash
# round 3, visit ak: planner moved to vpf_lzm_plan.h (refit on unrestricted band heights): Lanczos tests + timings at 32 / 8 / 1 frames per dispatch
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or policy or fuzz_resize or graph or arena or tables" 2>&1 | tail -2
for n in 0 8; do VPF_BENCH_N=$n VPF_BENCH_ONLY=lanczos VPF_BENCH_ONE=1 timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | sed "s/^/[N=$n] /" | cut -c1-260; done | tee gpurun_out/r03ak_bench.txt
