#!/bin/bash
# (the lab variant LzMfma8w6 / knob 0x80000 this script drives was reverted after the visit: no gain — profiles/r05_lanczos_six_staging_loads_ab.txt)
# round 5, visit v: the 320-B rows of the 2x down-scales staged by six loads per lane (eight lanes per row: conflict-free) against five (knob 0x80000), Lanczos-3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "lanczos_mfma_kernel_shapes" 2>&1 | tail -2)
(SWEEP_INTERP=2 SWEEP_PASSES=4 SWEEP_CASES="RGB:3840x2160:1920x1080,NV12:3840x2160:1920x1080,YUV420:3840x2160:1920x1080,Y:3840x2160:1920x1080,RGB:1920x1080:960x540" timeout 400 python tools/band_knob_sweep.py 0 0x80000 0x811 0x80811 0x822 0x80822 2>&1 | grep knobs) > $O/r05_v_lanczos_six_loads.txt; cat $O/r05_v_lanczos_six_loads.txt
