#!/bin/bash
# round 4, visit ak: whole GPU suite + the resize fuzz families (wider down-scale range: two-chunk windows, half tiles), 12 000 seeds per family = 36 000 cases, EXACT asserted
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -n 4 > gpurun_out/r04ak_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04ak_pytest.txt | cut -c1-300
VPF_FUZZ_SEEDS=12000 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -n 8 -k "fuzz" > gpurun_out/r04ak_fuzz_soak.txt 2>&1; echo "soak rc $?"; tail -3 gpurun_out/r04ak_fuzz_soak.txt | cut -c1-300
