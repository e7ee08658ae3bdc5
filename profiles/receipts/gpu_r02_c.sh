#!/bin/bash
# round 2, visit C: full GPU suite after the product/lab split + bench (default, sweep) to confirm the headline is unchanged
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -40 ) > gpurun_out/r02_pytest_gpu.log 2>&1
timeout 300 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
timeout 900 python bench.py --sweep --steps 20 --warmup 3 --no-cpu > gpurun_out/r02_bench_sweep.json 2> gpurun_out/r02_bench_sweep.log
tail -25 gpurun_out/r02_pytest_gpu.log; cat gpurun_out/r02_bench_default.json; tail -3 gpurun_out/r02_bench_default.err; cat gpurun_out/r02_bench_sweep.log | grep -v amdgpu.ids | tail -50
