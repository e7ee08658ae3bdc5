#!/bin/bash
# round 2, visit N: march policy over batch sizes (Lanczos lines, all formats), optional forced rows per wave
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "march or fuzz_resize_batch or resize_batch_equals or policy_picks" -n 4 2>&1 | tail -3 > gpurun_out/r02_n_pytest.txt
for m in ${MARCH_SWEEP:-0}; do for nb in ${NB_SWEEP:-0 16 8 4}; do echo "== frames per batch $nb (0 = 32), march $m (0 = policy)"; VPF_BENCH_MARCH=$m VPF_BENCH_N=$nb VPF_BENCH_Y=1 VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch" | grep -v 416x416 | cut -c1-100; done; done > gpurun_out/r02_n_policy.txt
cat gpurun_out/r02_n_pytest.txt gpurun_out/r02_n_policy.txt
