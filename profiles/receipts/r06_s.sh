#!/bin/bash
# round 6, visit s: a fuzz soak over fresh seeds on the FINAL tree (the planner's two n = 1 rules choose new launch shapes for one-frame Lanczos cases)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
(VPF_FUZZ_SEEDS=4000 VPF_FUZZ_FIRST=700000 timeout 80 python -m pytest tests/test_gpu_parity.py -q -n 8 -k fuzz -p no:cacheprovider 2>&1 | tail -3) > $O/r06_s_fuzz_soak.txt; tail -1 $O/r06_s_fuzz_soak.txt
