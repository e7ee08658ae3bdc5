#!/bin/bash
# round 4, visit v: single-frame Lanczos rule (small single planes -> tile kernel): parity tests of the Lanczos families, the sweep under the policy, the per-frame table, the sample chain
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "lanczos or fuzz_resize or workspace or resize" > gpurun_out/r04v_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04v_pytest.txt | cut -c1-300
SWEEP_POLICY_ONLY=1 timeout 600 python tools/lanczos_single_sweep.py 2>&1 | grep "lz-single" > gpurun_out/r04v_lanczos_single_policy.txt; tail -3 gpurun_out/r04v_lanczos_single_policy.txt
timeout 300 python tools/chain_bench.py 2>&1 | grep chain | tee gpurun_out/r04v_chain.txt
