#!/bin/bash
# round 3, visit bf: the refitted Lanczos launch planner — sweeps (minimum of three interleaved passes per shape; policy column = the new planner), Lanczos parity tests, the resize table
mkdir -p gpurun_out
for n in 32 8 1; do timeout 900 python tools/lanczos_shape_sweep.py $n 3 2>&1 | grep lzm-sweep > gpurun_out/r03bf_shape_sweep_n${n}.txt; done
cut -c1-100 gpurun_out/r03bf_shape_sweep_n*.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "lanczos or resize or fuzz" 2>&1 | tail -2 | tee gpurun_out/r03bf_pytest.txt
timeout 900 python tools/resize_batch_bench.py 2>&1 | grep "lanczos3" | cut -c1-120 | tee gpurun_out/r03bf_resize_batch_lanczos.txt
