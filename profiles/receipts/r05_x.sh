#!/bin/bash
# round 5, visit x: us per frame against the frames per dispatch (32 / 64 / 96 / 128) for the larger planes — where do 128-frame dispatches stop paying?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
for i in 1 2; do for n in 32 64 96 128 32; do echo "== frames per dispatch $n, interp $i"; SWEEP_CASES="RGB:1920x1080:1280x720,RGB:1280x720:1920x1080,RGB:3840x2160:1920x1080,NV12:3840x2160:1920x1080,YUV420:1920x1080:1280x720,Y:3840x2160:1920x1080,RGB:1920x1080:416x416" SWEEP_N=$n SWEEP_INTERP=$i timeout 300 python tools/band_knob_sweep.py 0 2>&1 | grep knobs | tail -2; done; done > $O/r05_x_frames_per_dispatch_curve.txt; cat $O/r05_x_frames_per_dispatch_curve.txt
