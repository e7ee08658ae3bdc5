#!/bin/bash
# round 3, visit e: strip width / band height A/B of the matrix-core Lanczos kernel (RGB lines only)
mkdir -p gpurun_out
for shape in 0 0x417 0x40c 0x817 0x80c 0x82e; do
  echo "== VPF_BENCH_MFMA=$shape"
  VPF_BENCH_MFMA=$shape VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\] RGB" | grep -v 416x416
done > gpurun_out/r03e_bench.txt 2>&1
cat gpurun_out/r03e_bench.txt
