#!/bin/bash
# round 5, visit w: launch-shape sweep of the matrix-core Lanczos kernel at 64 frames per dispatch (between the two fitted points of the planner)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python tools/lanczos_shape_sweep.py 64 2 2>&1 | grep lzm-sweep) > $O/r05_w_lanczos_shape_sweep_n64.txt
(SWEEP_Y=1 timeout 400 python tools/lanczos_shape_sweep.py 64 2 2>&1 | grep lzm-sweep) >> $O/r05_w_lanczos_shape_sweep_n64.txt
cut -c1-400 $O/r05_w_lanczos_shape_sweep_n64.txt
