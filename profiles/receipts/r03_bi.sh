#!/bin/bash
# round 3, visit bi: final code — whole GPU suite, smoke, fuzz soak of all four families (VPF_FUZZ_SEEDS=12000 -> 48 000 tests)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee gpurun_out/r03bi_pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/r03bi_smoke.txt
VPF_FUZZ_SEEDS=12000 timeout 3000 python -m pytest tests/test_gpu_parity.py -q -x -n 6 -k "fuzz" > gpurun_out/r03bi_fuzz_soak_big.txt 2>&1; tail -1 gpurun_out/r03bi_fuzz_soak_big.txt
