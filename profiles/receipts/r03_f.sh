#!/bin/bash
# round 3, visit f: quick A/B of one kernel change: parity (lanczos subset) + policy-shape timings
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma" 2>&1 | tail -2
VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch" | grep -v 416x416 | cut -c1-175
