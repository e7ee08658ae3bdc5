#!/bin/bash
# round 2, visit I: convert-once fused kernel: parity + scales bench; float tests with variants
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pynvcodec.py -q -x -k "fused or float or fuzz_resize" 2>&1 | tail -12 ) > gpurun_out/r02_i_pytest.log 2>&1
timeout 300 python tools/fused_scales_bench.py > gpurun_out/r02_fused_scales.txt 2>&1
cat gpurun_out/r02_i_pytest.log; grep -v amdgpu.ids gpurun_out/r02_fused_scales.txt
