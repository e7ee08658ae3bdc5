#!/bin/bash
# round 6, visit r (visit o again on the FINAL tree, after the n = 1 down-scale rule): the driver's own round-end sequence on the final tree — the GPU suite in ONE process (-x -q), smoke(), the default bench line.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 380 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -5 > $O/r06_r_pytest_single_process.txt; cat $O/r06_r_pytest_single_process.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 > $O/r06_r_smoke.txt; cat $O/r06_r_smoke.txt
timeout 100 python bench.py > $O/r06_r_bench_default.json 2> $O/r06_r_bench_default.err; cut -c1-400 $O/r06_r_bench_default.json
