#!/bin/bash
# round 6, visit p: the planner's n = 1 regret on the 4K DOWN-scales, on the final tree — ONE frame per dispatch, policy (knob 0) against 8- and 4-tile
# strips in bands of 2 .. 6 tiles, RGB / NV12 / YUV420 / Y, 4K -> 1080p and 4K -> 1440p, sustained protocol.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
timeout 200 python tools/lab/ab/lone_lanczos.py videoprocessingframework_amd/libvpfhip.so --down 2>&1 | grep "\[lone\]" > $O/r06_p_lone_downscales.txt
cut -c1-120 $O/r06_p_lone_downscales.txt
