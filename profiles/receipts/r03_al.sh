#!/bin/bash
# round 3, visit al: tiled separable Lanczos back (integer definition) for the shapes beyond the matrix-core kernel: tests, strong down-scales, the sample chain
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or fuzz_resize or resize_batch or tiled or policy" 2>&1 | tail -3
VPF_BENCH_ONLY=lanczos VPF_BENCH_ONE=1 timeout 300 python tools/resize_batch_bench.py 2>&1 | grep -E "416x416" | tee gpurun_out/r03al_416.txt
VPF_BENCH_MFMA=1 VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | sed 's/^/[no mfma] /' | cut -c1-220 | tee gpurun_out/r03al_tile.txt
timeout 300 python tools/chain_bench.py 2>&1 | grep chain | cut -c1-330 | tee gpurun_out/r03al_chain.txt
