#!/bin/bash
# round 4, visit j: whole GPU suite (per-device tests, cross-stream upload test, EXACT in the fuzz families) + the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04j_pytest.txt 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r04j_pytest.txt
timeout 600 python bench.py > gpurun_out/r04j_bench.json 2> gpurun_out/r04j_bench.err; echo "bench rc $?"; cut -c1-700 gpurun_out/r04j_bench.json
