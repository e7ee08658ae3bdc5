#!/bin/bash
# round 3, visit as: tiled Lanczos with the SDWA shift / clamp / pack in its vertical pass: tests, timings (matrix-core kernel off = every shape on the tiled kernel; default policy)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or fuzz_resize or tiled" 2>&1 | tail -2
VPF_BENCH_MFMA=1 VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | sed 's/^/[tile everywhere] /' | cut -c1-220 | tee gpurun_out/r03as_tile.txt
timeout 300 python tools/chain_bench.py 2>&1 | grep "lanczos3" | cut -c1-330
