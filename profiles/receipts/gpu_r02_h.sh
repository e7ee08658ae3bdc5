#!/bin/bash
# round 2, visit H: Lanczos RNE pack + float tiled path: parity + benches
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pynvcodec.py -q -x -k "lanczos or resize or tiled or batch or float" 2>&1 | tail -12 ) > gpurun_out/r02_h_pytest.log 2>&1
timeout 300 python tools/lanczos_bench.py > gpurun_out/r02_h_lanczos.txt 2>&1
timeout 300 python tools/float_resize_bench.py > gpurun_out/r02_h_float.txt 2>&1
cat gpurun_out/r02_h_pytest.log; grep -v amdgpu.ids gpurun_out/r02_h_lanczos.txt; grep -v amdgpu.ids gpurun_out/r02_h_float.txt
