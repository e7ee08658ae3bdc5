#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "variant or fuzz or full_size" 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --sweep --steps 20 --warmup 3 --no-cpu 2>&1 | grep "sweep" | grep "batch" | grep -v planar; done
