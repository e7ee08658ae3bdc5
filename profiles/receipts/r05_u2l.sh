#!/bin/bash
# round 5, visit u2l: the shared-column form of the matrix-core Lanczos kernel (VPF_TUNE_RESIZE_MFMA | 0x100000): parity, then A/B against the classic form
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "lanczos or fuzz or mfma" 2>&1 | tail -8) > $O/r05_u2l_pytest.txt; tail -4 $O/r05_u2l_pytest.txt
export SWEEP_INTERP=2 SWEEP_CASES="RGB:1920x1080:1280x720,Y:1920x1080:1280x720,NV12:1920x1080:1280x720,YUV420:1920x1080:1280x720,RGB:3840x2160:1920x1080,NV12:3840x2160:1920x1080,RGB:1920x1080:1600x900"
(SWEEP_N=32 timeout 600 python tools/band_knob_sweep.py 0 0x100000 0x800 0x100800 0x400 0x100400 2>&1 | grep knobs) > $O/r05_u2l_ab_n32.txt; cat $O/r05_u2l_ab_n32.txt
(SWEEP_N=128 timeout 600 python tools/band_knob_sweep.py 0 0x100000 2>&1 | grep knobs) > $O/r05_u2l_ab_n128.txt; cat $O/r05_u2l_ab_n128.txt
