#!/bin/bash
# round 4, visit u: single-frame Lanczos dispatches, matrix-core kernel against tile kernel over sizes and factors (the data the single-frame launch rule is fitted to)
mkdir -p gpurun_out
timeout 900 python tools/lanczos_single_sweep.py 2>&1 | grep "lz-single" | tee gpurun_out/r04u_lanczos_single_sweep.txt
