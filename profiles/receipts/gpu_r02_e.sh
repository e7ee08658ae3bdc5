#!/bin/bash
# round 2, visit E: tile shape sweep under rocprofv3 kernel-trace + resize parity tests
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or resize or fuzz_resize" 2>&1 | tail -8 ) > gpurun_out/r02_e_pytest.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/tile_sweep" -o t -- python "$GRAFT_REPO_ROOT/tools/tile_shape_sweep.py" "$GRAFT_REPO_ROOT/gpurun_out/tile_sweep_log.json" > "$GRAFT_REPO_ROOT/gpurun_out/tile_sweep.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/tile_shape_post.py gpurun_out/tile_sweep_log.json $(find gpurun_out/tile_sweep -name "*kernel_trace.csv" | head -1) > gpurun_out/r02_tile_shape_sweep.txt 2>&1
cat gpurun_out/r02_e_pytest.log; tail -3 gpurun_out/tile_sweep.log; cat gpurun_out/r02_tile_shape_sweep.txt
