#!/bin/bash
# round 3, visit i: the reference's own ConvertSurface / ResizeSurface / RemapSurface on the MI355X through the C ABI
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_caller.py -q -x > gpurun_out/r03i_pytest.txt 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r03i_pytest.txt
