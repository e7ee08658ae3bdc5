#!/bin/bash
# round 2, visit F: batch resize / remap: parity + bench; tile shape sweep of the current kernel
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pynvcodec.py -q -x -k "lanczos or resize or remap or tiled or batch" 2>&1 | tail -25 ) > gpurun_out/r02_f_pytest.log 2>&1
timeout 600 python tools/resize_batch_bench.py > gpurun_out/r02_resize_batch.txt 2>&1
timeout 300 python tools/lanczos_bench.py > gpurun_out/r02_f_lanczos.txt 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/tile_sweep" -o t -- python "$GRAFT_REPO_ROOT/tools/tile_shape_sweep.py" "$GRAFT_REPO_ROOT/gpurun_out/tile_sweep_log.json" > "$GRAFT_REPO_ROOT/gpurun_out/tile_sweep.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/tile_shape_post.py gpurun_out/tile_sweep_log.json $(find gpurun_out/tile_sweep -name "*kernel_trace.csv" | head -1) > gpurun_out/r02_tile_shape_sweep.txt 2>&1
cat gpurun_out/r02_f_pytest.log; grep -v amdgpu.ids gpurun_out/r02_resize_batch.txt; grep -v amdgpu.ids gpurun_out/r02_f_lanczos.txt; grep -E "^[0-9]|policy|ty 16 wpb 8|ty 32 wpb 8|ty 64 wpb 8|ty 32 wpb 4" gpurun_out/r02_tile_shape_sweep.txt
