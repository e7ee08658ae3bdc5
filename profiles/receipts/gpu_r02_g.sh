#!/bin/bash
# round 2, visit G: parity of the resize family after packed-fp32 / 2-row unroll + batch bench + per-frame bench
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pynvcodec.py -q -x -k "lanczos or resize or remap or tiled or batch or fused" 2>&1 | tail -12 ) > gpurun_out/r02_g_pytest.log 2>&1
timeout 600 python tools/resize_batch_bench.py > gpurun_out/r02_resize_batch.txt 2>&1
timeout 300 python tools/lanczos_bench.py > gpurun_out/r02_g_lanczos.txt 2>&1
cat gpurun_out/r02_g_pytest.log; grep -v amdgpu.ids gpurun_out/r02_resize_batch.txt; grep -v amdgpu.ids gpurun_out/r02_g_lanczos.txt
