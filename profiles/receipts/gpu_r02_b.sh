#!/bin/bash
# round 2, visit B: whole GPU suite incl. the pin-kit rehearsal and the reference's own ConvertSurface on the GPU
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -60 ) > gpurun_out/r02_pytest_gpu.log 2>&1
cat gpurun_out/r02_pytest_gpu.log
