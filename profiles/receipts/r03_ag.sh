#!/bin/bash
# round 3, visit ag: gather Lanczos with all rows' taps in flight together + 24-bit multiplies: tests, the sample chain, strong down-scales
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or fuzz_resize or resize_batch" 2>&1 | tail -3
timeout 300 python tools/chain_bench.py 2>&1 | grep chain | tee gpurun_out/r03ag_chain.txt
VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep -E "416x416" | tee gpurun_out/r03ag_416.txt
VPF_BENCH_MFMA=1 VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | sed 's/^/[gather everywhere] /' | cut -c1-220 | tee gpurun_out/r03ag_gather.txt
