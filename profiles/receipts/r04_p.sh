#!/bin/bash
# round 4, visit p: the whole GPU suite with the two-role Lanczos form in the fuzz families' knob list; a soak of the resize fuzz families (EXACT asserted in every case)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04p_pytest.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04p_pytest.txt | cut -c1-300
VPF_FUZZ_SEEDS=3000 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -n 6 -k "fuzz_resize" > gpurun_out/r04p_fuzz_soak.txt 2>&1; echo "soak rc $?"; tail -3 gpurun_out/r04p_fuzz_soak.txt | cut -c1-300
