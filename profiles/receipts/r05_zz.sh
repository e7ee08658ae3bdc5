#!/bin/bash
# round 5, visit zz: the resize table and the suite once more after the planner's large-launch branch was refitted to the 64- and 128-frame sweeps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -4) > $O/r05_zz_pytest.txt; tail -1 $O/r05_zz_pytest.txt
(VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\|remap") > $O/r05_zz_resize_batch.txt; cut -c1-150 $O/r05_zz_resize_batch.txt
