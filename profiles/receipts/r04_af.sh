#!/bin/bash
# round 4, visit af: two-chunk windows in the matrix-core Lanczos kernel (strong horizontal down-scales): parity of the Lanczos families, then timings
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "lanczos or fuzz_resize" > gpurun_out/r04af_pytest.txt 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r04af_pytest.txt | cut -c1-400
VPF_BENCH_Y=1 VPF_BENCH_ONLY=lanczos timeout 600 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch" | grep "416" | cut -c1-200 | tee gpurun_out/r04af_lanczos_416.txt
