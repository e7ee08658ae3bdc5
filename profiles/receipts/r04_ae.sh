#!/bin/bash
# round 4, visit ae: windowed tap reads in the product: bilinear parity families (band kernels forced and by policy, fuzz, multi-plane, graph), a short fuzz soak
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "row_band or march or fuzz_resize or graph or bilinear or resize" > gpurun_out/r04ae_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04ae_pytest.txt | cut -c1-300
VPF_FUZZ_SEEDS=1500 timeout 1200 python -m pytest tests/test_gpu_parity.py -q -n 6 -k "fuzz_resize" > gpurun_out/r04ae_fuzz.txt 2>&1; echo "soak rc $?"; tail -2 gpurun_out/r04ae_fuzz.txt | cut -c1-300
