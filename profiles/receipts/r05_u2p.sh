#!/bin/bash
# round 5, visit u2p: the 4-tile ring-of-two kernel with two source tiles in flight (126 VGPRs, x0) against one (116, x1): two builds, alternating on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD SWEEP_INTERP=2 SWEEP_N=32 SWEEP_CASES="RGB:1280x720:1920x1080,Y:1280x720:1920x1080,NV12:1280x720:1920x1080,YUV420:1280x720:1920x1080,RGB:1920x1080:3840x2160,Y:1920x1080:3840x2160"
for p in 1 2 3; do for x in 0 1; do echo "== build x$x (0: two tiles in flight, 1: one)"; SWEEP_LIB=tools/lab/ablate/libvpfhip_x$x.so timeout 300 python tools/band_knob_sweep.py 0 0x400 2>&1 | grep knobs | tail -3; done; done > $O/r05_u2p_4u_prefetch_ab.txt; cat $O/r05_u2p_4u_prefetch_ab.txt
