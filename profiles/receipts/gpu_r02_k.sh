#!/bin/bash
# round 2, visit K: row-band kernels: parity (forced rows per wave, fuzz) and the batched bilinear lines, A/B of rows per wave
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pynvcodec.py -q -x -k "band or fuzz_resize or fused or convert_resize or resize" -n 4 2>&1 | tail -8 > gpurun_out/r02_k_pytest.txt
for b in ${BAND_SWEEP:-0}; do echo "== band $b"; VPF_BENCH_BAND=$b VPF_BENCH_ONLY=bilinear timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | cut -c1-130; done > gpurun_out/r02_k_band.txt
cat gpurun_out/r02_k_pytest.txt gpurun_out/r02_k_band.txt
