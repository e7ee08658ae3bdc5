#!/bin/bash
# round 5, visit u2c: launch-shape sweep of the up-scales with the ring-of-two kernels (8-tile strips: 3 workgroups per CU; 4-tile strips: 4, LDS trimmed)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "lanczos or fuzz or mfma" 2>&1 | tail -3) > $O/r05_u2c_pytest.txt; tail -1 $O/r05_u2c_pytest.txt
export SWEEP_SIZES="1280x720:1920x1080,1920x1080:3840x2160,960x540:1920x1080"
(timeout 900 python tools/lanczos_shape_sweep.py 32 3 2>&1 | grep lzm-sweep) > $O/r05_u2c_sweep_n32.txt; cut -c1-400 $O/r05_u2c_sweep_n32.txt
(SWEEP_Y=1 timeout 600 python tools/lanczos_shape_sweep.py 32 3 2>&1 | grep lzm-sweep) > $O/r05_u2c_sweep_y_n32.txt; cut -c1-400 $O/r05_u2c_sweep_y_n32.txt
