#!/bin/bash
# round 3, visit x: Lanczos matrix-core kernel with per-shape weight tables: tests, timings, shape sweep (the planner's fixed cost changed)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or policy or fuzz_resize or graph" > gpurun_out/r03x_pytest.txt 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r03x_pytest.txt
VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03x_bench.txt; cat gpurun_out/r03x_bench.txt
timeout 600 python tools/lanczos_shape_sweep.py 32 2>&1 | grep lzm-sweep > gpurun_out/r03x_sweep_n32.txt; cat gpurun_out/r03x_sweep_n32.txt
