#!/bin/bash
# round 5, visit u2v: one frame per dispatch — the two-role form (pass 1 / pass 2 on different waves, | 0x20000) against the policy: does splitting a lone wave's serial chain pay?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD SWEEP_INTERP=2 SWEEP_N=1 SWEEP_CASES="RGB:3840x2160:1920x1080,NV12:3840x2160:1920x1080,RGB:1920x1080:1280x720,RGB:3840x2160:2560x1440,RGB:1920x1080:3840x2160"
(timeout 900 python tools/band_knob_sweep.py 0 0x802 0x804 0x20002 0x20003 0x20004 0x20006 0x20008 2>&1 | grep knobs) > $O/r05_u2v_single_pair.txt; cat $O/r05_u2v_single_pair.txt
