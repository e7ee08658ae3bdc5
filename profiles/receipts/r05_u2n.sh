#!/bin/bash
# round 5, visit u2n: the shared-column form at forced band heights (it wants its own launch shapes: a workgroup is four bands of one strip)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD SWEEP_INTERP=2 SWEEP_N=32 SWEEP_CASES="RGB:1920x1080:1280x720,Y:1920x1080:1280x720,NV12:1920x1080:1280x720,YUV420:1920x1080:1280x720,RGB:1920x1080:1600x900,RGB:3840x2160:2560x1440"
(timeout 900 python tools/band_knob_sweep.py 0 0x800 0x400 0x100802 0x100803 0x100804 0x100806 0x100808 0x10080c 0x100402 0x100403 0x100404 0x100406 0x100408 0x10040c 2>&1 | grep knobs) > $O/r05_u2n_sc_band_heights.txt; cat $O/r05_u2n_sc_band_heights.txt
