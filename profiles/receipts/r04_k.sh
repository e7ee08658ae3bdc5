#!/bin/bash
# round 4, visit k: caller-owned Lanczos workspace + LRU fallback arena: parity tests, the whole GPU suite
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04k_pytest.txt 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r04k_pytest.txt | cut -c1-300
