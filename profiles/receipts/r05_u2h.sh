#!/bin/bash
# round 5, visit u2h: fuzz soak of all four families on the code with the ring-of-two Lanczos kernels (VPF_FUZZ_SEEDS=6000: 24 000 tests)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONPATH=$PWD
VPF_FUZZ_SEEDS=6000 timeout 3000 python -m pytest tests/test_gpu_parity.py -q -x -n 6 -k "fuzz" > gpurun_out/r05_u2h_fuzz_soak.txt 2>&1; tail -3 gpurun_out/r05_u2h_fuzz_soak.txt
