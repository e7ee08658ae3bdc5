#!/bin/bash
# round 5, visit u2s: the planner with the wide 8-tile ring-of-two strips behind a volume gate: the GPU suite, the same-box A/B on 1.5 x shapes, the resize table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -4) > $O/r05_u2s_pytest.txt; tail -1 $O/r05_u2s_pytest.txt
export SWEEP_INTERP=2 SWEEP_CASES="RGB:2560x1440:3840x2160,RGB:1280x720:1920x1080,RGB:1920x1080:3840x2160,NV12:2560x1440:3840x2160,RGB:1920x1080:2880x1620,RGB:1280x720:1600x900"
(SWEEP_N=32 timeout 600 python tools/band_knob_sweep.py 0 0x400 0x800 0x80000 2>&1 | grep knobs) > $O/r05_u2s_ab_n32.txt; cat $O/r05_u2s_ab_n32.txt
(VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\|remap") > $O/r05_u2s_resize_batch.txt; grep -i "lanczos" $O/r05_u2s_resize_batch.txt | grep "3840x2160 \|->1920x1080 l\|>1920x1080 lan" | cut -c1-140
