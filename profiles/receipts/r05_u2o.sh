#!/bin/bash
# round 5, visit u2o: launch-shape sweeps of the down-scales with the round's final kernels (the planner's <= 32-frame constants date from round 3)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
export SWEEP_SIZES="1920x1080:1280x720,3840x2160:1920x1080,1920x1080:1600x900,3840x2160:2560x1440"
for n in 32 8 1; do
  (timeout 900 python tools/lanczos_shape_sweep.py $n 3 2>&1 | grep lzm-sweep; SWEEP_Y=1 timeout 600 python tools/lanczos_shape_sweep.py $n 3 2>&1 | grep lzm-sweep) > $O/r05_u2o_sweep_down_n$n.txt; cut -c1-110 $O/r05_u2o_sweep_down_n$n.txt
done
