#!/bin/bash
# round 3, visit k: the whole GPU suite after the uploader change
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r03k_pytest.txt 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r03k_pytest.txt
