#!/bin/bash
# round 2, visit K: fused strip kernel on the band walk
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pynvcodec.py -q -x -k "band or fuzz_resize or fused or convert_resize or resize" -n 4 2>&1 | tail -8 > gpurun_out/r02_k_pytest.txt
timeout 300 python tools/fused_scales_bench.py > gpurun_out/r02_k_fused.txt 2>&1
cat gpurun_out/r02_k_pytest.txt gpurun_out/r02_k_fused.txt
