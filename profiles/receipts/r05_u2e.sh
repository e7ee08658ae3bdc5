#!/bin/bash
# round 5, visit u2e: the planner knows the ring-of-two kernels (per candidate): the whole GPU suite, the resize table, the same-box A/B against the ring of four
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -4) > $O/r05_u2e_pytest.txt; tail -1 $O/r05_u2e_pytest.txt
export SWEEP_INTERP=2 SWEEP_CASES="RGB:1920x1080:3840x2160,RGB:1280x720:1920x1080,Y:1280x720:1920x1080,NV12:1280x720:1920x1080,YUV420:1920x1080:3840x2160,RGB:960x540:1920x1080,NV12:1920x1080:3840x2160"
(SWEEP_N=32 timeout 600 python tools/band_knob_sweep.py 0 0x80000 2>&1 | grep knobs) > $O/r05_u2e_ab_n32.txt; cat $O/r05_u2e_ab_n32.txt
(VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\|remap") > $O/r05_u2e_resize_batch.txt; grep -i lanczos $O/r05_u2e_resize_batch.txt | cut -c1-150
