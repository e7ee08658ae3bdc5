#!/bin/bash
# round 3, visit ap: blocking tasks wait on a completion flag in pinned host memory (hipStreamWriteValue32 + spin) instead of hipStreamSynchronize: API tests, sample chain, transfer rates
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pynvcodec.py tests/test_gpu_reference_PySurface.py tests/test_gpu_multirank.py -q -x 2>&1 | tail -3
for spin in 200 0 200 0; do VPF_HIP_SYNC_SPIN_US=$spin timeout 300 python tools/chain_bench.py 2>&1 | grep "chain" | grep -v "ONE pass" | sed "s/^/[spin $spin] /" | cut -c1-12,100-330; done | tee gpurun_out/r03ap_chain.txt
for spin in 200 0; do VPF_HIP_SYNC_SPIN_US=$spin timeout 300 python tools/pipeline_bench.py 2>&1 | grep -E "1 thread|download" | sed "s/^/[spin $spin] /"; done | tee gpurun_out/r03ap_pipeline.txt
VPF_BENCH_ONLY=lanczos timeout 200 python tools/resize_batch_bench.py 2>&1 | grep -E "RGB    1920x1080->1280x720" | cut -c1-200
