#!/bin/bash
# round 3, visit at: four source tiles in flight per wave for small launches of the matrix-core Lanczos kernel: tests, one frame per dispatch with / without (| 0x20000)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or fuzz_resize or policy or graph" 2>&1 | tail -2
for knob in 0 0x20000 0 0x20000; do VPF_BENCH_MFMA=$knob VPF_BENCH_ONLY=lanczos VPF_BENCH_ONE=1 timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | grep -v 416 | sed "s/^/[knob $knob] /" | cut -c1-16,52-70,130-260; done | tee gpurun_out/r03at_ab.txt
