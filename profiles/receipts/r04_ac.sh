#!/bin/bash
# round 4, visit ac: the march form of the row-band bilinear kernel in the product: parity (forced + by policy + fuzz), then the bilinear table
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "row_band or march or fuzz_resize or graph" > gpurun_out/r04ac_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04ac_pytest.txt | cut -c1-400
VPF_BENCH_Y=1 VPF_BENCH_ONLY=bilinear timeout 600 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | cut -c1-150 | tee gpurun_out/r04ac_bilinear.txt
