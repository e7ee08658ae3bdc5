#!/bin/bash
# round 6, visit q: the final tree with the planner's n = 1 rule for the down-scales (ring of four) — the whole GPU suite, then ONE down-scaled 4K
# frame per dispatch: policy (knob 0 = the new pick) against both strip widths in bands of 2 .. 6 tiles (the old pick among them); the up-scales once more.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
timeout 200 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -6 > $O/r06_q_pytest.txt; cat $O/r06_q_pytest.txt
timeout 120 python tools/lab/ab/lone_lanczos.py videoprocessingframework_amd/libvpfhip.so --down 2>&1 | grep "\[lone\]" > $O/r06_q_lone_downscales.txt
grep -A1 "lanczos3" $O/r06_q_lone_downscales.txt | cut -c1-120
