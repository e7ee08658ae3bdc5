#!/bin/bash
# round 3, visit o: launch-shape sweep of the matrix-core Lanczos kernel (planner input)
mkdir -p gpurun_out
timeout 900 python tools/lanczos_shape_sweep.py 32 2>&1 | grep lzm-sweep > gpurun_out/r03o_shape_sweep.txt; cat gpurun_out/r03o_shape_sweep.txt
timeout 600 python tools/lanczos_shape_sweep.py 8 2>&1 | grep lzm-sweep > gpurun_out/r03o_shape_sweep_n8.txt; cat gpurun_out/r03o_shape_sweep_n8.txt
