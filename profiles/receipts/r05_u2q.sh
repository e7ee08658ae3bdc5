#!/bin/bash
# round 5, visit u2q: 8-tile ring-of-two strips at 1.5 x (LzMfma8uw, rows of up to 256 B): parity, then the launch-shape sweep of the 1.5 x up-scales
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "lanczos or fuzz or mfma" 2>&1 | tail -4) > $O/r05_u2q_pytest.txt; tail -2 $O/r05_u2q_pytest.txt
export SWEEP_SIZES="1280x720:1920x1080,2560x1440:3840x2160,1280x720:1600x900"
(timeout 900 python tools/lanczos_shape_sweep.py 32 3 2>&1 | grep lzm-sweep; SWEEP_Y=1 timeout 600 python tools/lanczos_shape_sweep.py 32 3 2>&1 | grep lzm-sweep) > $O/r05_u2q_sweep_n32.txt; cut -c1-420 $O/r05_u2q_sweep_n32.txt
