#!/bin/bash
# round 6, visit n: the final tree — the planner's n = 1 band-height rule restricted to launches whose planes have equal strip counts (visit m's first
# form lost 9-14 % on YUV420): the whole GPU suite, then ONE up-scaled frame per dispatch, policy (knob 0) against forced bands of 3 .. 9 tiles.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -6 > $O/r06_n_pytest.txt; cat $O/r06_n_pytest.txt
timeout 150 python tools/lab/ab/lone_lanczos.py videoprocessingframework_amd/libvpfhip.so --up 2>&1 | grep "\[lone\]" > $O/r06_n_lone_upscales.txt
cut -c1-120 $O/r06_n_lone_upscales.txt
