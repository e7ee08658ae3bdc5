#!/bin/bash
# round 3, visit ab: shape sweep of the Lanczos matrix-core kernel with weight tables (32 / 8 / 1 frames per dispatch) to refit the planner
mkdir -p gpurun_out
for n in 32 8 1; do timeout 600 python tools/lanczos_shape_sweep.py $n 2>&1 | grep lzm-sweep > gpurun_out/r03ab_sweep_n$n.txt; cat gpurun_out/r03ab_sweep_n$n.txt; done
