#!/bin/bash
# round 2, visit O: p16x (blocks numbered straight through the picture) as the default packed NV12 / YUV420 -> RGB kernel
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 -n 4 2>&1 | tail -6 ) > gpurun_out/r02_o_pytest.txt
timeout 600 python bench.py --sweep > gpurun_out/r02_bench_sweep.log 2>&1
timeout 300 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
cat gpurun_out/r02_o_pytest.txt; grep -E "variant|lab" gpurun_out/r02_bench_sweep.log | cut -c1-120; cat gpurun_out/r02_bench_default.json | cut -c1-400
