#!/bin/bash
# round 5, visit u2r: launch-shape sweeps of the up-scales once more, with the wide 8-tile ring-of-two strips in (1.5 x shapes changed kernels), five size pairs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
export SWEEP_SIZES="1280x720:1920x1080,1920x1080:3840x2160,960x540:1920x1080,2560x1440:3840x2160,1280x720:1600x900"
for n in 32 8 1; do
  (timeout 900 python tools/lanczos_shape_sweep.py $n 3 2>&1 | grep lzm-sweep; SWEEP_Y=1 timeout 600 python tools/lanczos_shape_sweep.py $n 3 2>&1 | grep lzm-sweep) > $O/r05_u2r_sweep_up_n$n.txt; cut -c1-100 $O/r05_u2r_sweep_up_n$n.txt
done
export SWEEP_SIZES="1280x720:1920x1080,960x540:1920x1080,640x360:1280x720,1280x720:1600x900"
for n in 64 128; do
  (timeout 900 python tools/lanczos_shape_sweep.py $n 3 2>&1 | grep lzm-sweep; SWEEP_Y=1 timeout 600 python tools/lanczos_shape_sweep.py $n 3 2>&1 | grep lzm-sweep) > $O/r05_u2r_sweep_up_n$n.txt; cut -c1-100 $O/r05_u2r_sweep_up_n$n.txt
done
