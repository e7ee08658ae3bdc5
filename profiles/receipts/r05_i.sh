#!/bin/bash
# round 5, visit i: 8 pixels per lane on every 1-channel plane (YUV420's chroma planes fill 512-column chunks to 62 %): knob 0x20000 against the policy, 128 frames per dispatch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "row_band or persistent or march" 2>&1 | tail -3)
(SWEEP_N=128 SWEEP_CASES="YUV420:1920x1080:1280x720,YUV420:3840x2160:1920x1080,YUV420:1280x720:1920x1080,NV12:1920x1080:1280x720,Y:1920x1080:1280x720" timeout 400 python tools/band_knob_sweep.py 0 0x20000 0x20304 0x20204 0x20008 0x8 2>&1 | grep knobs) > $O/r05_i_band_knobs_px8.txt; cat $O/r05_i_band_knobs_px8.txt
