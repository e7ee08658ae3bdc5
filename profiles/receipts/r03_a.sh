#!/bin/bash
# round 3, visit a: first contact of the matrix-core Lanczos kernel — diagnostics, the Lanczos parity tests, batched timings
mkdir -p gpurun_out
export VPF_HIP_LOG=1
timeout 300 python tools/lzm_debug.py > gpurun_out/r03a_debug.txt 2>&1; echo "debug rc $?"; grep -c "^OK" gpurun_out/r03a_debug.txt; grep -m 40 -A9 "^FAIL" gpurun_out/r03a_debug.txt | head -120; tail -1 gpurun_out/r03a_debug.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma" > gpurun_out/r03a_pytest.txt 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r03a_pytest.txt
for shape in 0 0x402 0x802 0x404 0x808; do
  echo "== VPF_BENCH_MFMA=$shape"
  VPF_BENCH_MFMA=$shape VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch
done > gpurun_out/r03a_bench.txt 2>&1
cat gpurun_out/r03a_bench.txt
