#!/bin/bash
# round 4, visit ao: the march form of the fused convert + resize strip kernel: parity (fused tests + fuzz), then 1 .. 4 bands per wave against the plain form
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "fused or convert_resize or fuzz_resize_and_fused" > gpurun_out/r04ao_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04ao_pytest.txt | cut -c1-300
timeout 600 python tools/lab/ab/fused_nb.py 32 2>&1 | grep "fused-nb" | tee gpurun_out/r04ao_fused_march.txt
