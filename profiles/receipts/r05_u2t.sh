#!/bin/bash
# round 5, visit u2t: fuzz soak of all four families on the FINAL tree, fresh seeds (VPF_FUZZ_SEEDS=12000 seeds 500 000 .. 511 999: 48 000 tests)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONPATH=$PWD
VPF_FUZZ_SEEDS=12000 VPF_FUZZ_FIRST=500000 timeout 3000 python -m pytest tests/test_gpu_parity.py -q -x -n 6 -k "fuzz" > gpurun_out/r05_u2t_fuzz_soak.txt 2>&1; tail -3 gpurun_out/r05_u2t_fuzz_soak.txt
