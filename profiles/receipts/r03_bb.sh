#!/bin/bash
# round 3, visit bb: fuzz soak of all four families after the pack / band-walk changes (VPF_FUZZ_SEEDS=6000 -> 24 000 tests), then the whole GPU suite
mkdir -p gpurun_out
VPF_FUZZ_SEEDS=6000 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -x -n 6 -k "fuzz" > gpurun_out/r03bb_fuzz_soak.txt 2>&1; tail -3 gpurun_out/r03bb_fuzz_soak.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r03bb_pytest.txt
