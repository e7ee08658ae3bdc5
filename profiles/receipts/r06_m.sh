#!/bin/bash
# round 6, visit m: the planner's n = 1 rule for the ring-of-two kernels (band height re-chosen with an occupancy term) — the whole GPU suite on the
# tree that carries it, then ONE up-scaled frame per dispatch: policy (knob 0 = the new pick) against forced 4-tile strips with bands of 3 .. 9 tiles
# (the old pick is among them), sustained protocol, two passes.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
timeout 400 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -6 > $O/r06_m_pytest.txt; cat $O/r06_m_pytest.txt
for pass in 1 2; do timeout 150 python tools/lab/ab/lone_lanczos.py videoprocessingframework_amd/libvpfhip.so --up 2>&1 | grep "\[lone\]"; done > $O/r06_m_lone_upscales.txt
cut -c1-160 $O/r06_m_lone_upscales.txt
timeout 120 python bench.py > $O/r06_m_bench_default.json 2> $O/r06_m_bench_default.err; tail -c 600 $O/r06_m_bench_default.json
