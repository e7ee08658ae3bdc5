#!/bin/bash
# round 2, visit L: the whole GPU suite after the band / march kernels + the batched resize table
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 -n 4 2>&1 | tail -15 ) > gpurun_out/r02_pytest_gpu.log 2>&1
timeout 600 python tools/resize_batch_bench.py > gpurun_out/r02_resize_batch.txt 2>&1
cat gpurun_out/r02_pytest_gpu.log gpurun_out/r02_resize_batch.txt
