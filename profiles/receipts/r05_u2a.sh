#!/bin/bash
# round 5, visit u2a: the ring of two (Lanczos up-scales, one K chunk in pass 2): parity, then A/B against the ring of four (VPF_TUNE_RESIZE_MFMA | 0x80000)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "lanczos or fuzz or mfma" 2>&1 | tail -6) > $O/r05_u2a_pytest.txt; tail -3 $O/r05_u2a_pytest.txt
export SWEEP_INTERP=2 SWEEP_CASES="RGB:1920x1080:3840x2160,RGB:1280x720:1920x1080,Y:1280x720:1920x1080,NV12:1280x720:1920x1080,YUV420:1920x1080:3840x2160,RGB:960x540:1920x1080,RGB:1280x720:3840x2160"
(SWEEP_N=32 timeout 600 python tools/band_knob_sweep.py 0 0x80000 2>&1 | grep knobs) > $O/r05_u2a_ab_n32.txt; cat $O/r05_u2a_ab_n32.txt
(SWEEP_N=128 timeout 600 python tools/band_knob_sweep.py 0 0x80000 2>&1 | grep knobs) > $O/r05_u2a_ab_n128.txt; cat $O/r05_u2a_ab_n128.txt
