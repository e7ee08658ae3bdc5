#!/bin/bash
# round 2, visit N: the batch kernels on ONE frame per dispatch (forced rows per wave) against the single-frame kernels: bilinear up-scales
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
{ echo "== single-frame kernels"; VPF_BENCH_ONLY=bilinear timeout 300 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch. RGB" | grep "3840x2160\$\|->3840x2160\|->1920x1080" | sed 's/batched.*| one/one/' | cut -c1-140;
  for b in 8 16; do echo "== band rows $b"; VPF_BENCH_BAND=$b VPF_BENCH_ONLY=bilinear timeout 300 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch. RGB" | grep "\->3840x2160\|->1920x1080" | sed 's/batched.*| one/one/' | cut -c1-140; done; } > gpurun_out/r02_n_single_up.txt
cat gpurun_out/r02_n_single_up.txt
