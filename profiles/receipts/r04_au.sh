#!/bin/bash
# round 4, visit au: a second fuzz soak over FRESH seeds (100 000 ..): 4 families x 12 000 = 48 000 more cases, EXACT asserted
mkdir -p gpurun_out
VPF_FUZZ_FIRST=100000 VPF_FUZZ_SEEDS=12000 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -n 8 -k "fuzz" > gpurun_out/r04au_fuzz_soak_fresh.txt 2>&1; echo "soak rc $?"; tail -3 gpurun_out/r04au_fuzz_soak_fresh.txt | cut -c1-300
