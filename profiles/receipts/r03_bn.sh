#!/bin/bash
# round 3, visit bn: fused strip kernel walking several bands per wave with the next band's rows requested before the blend — parity, then 1..4 bands per wave
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "fused or convert_resize" 2>&1 | tail -2
VPF_LAB_FUSED_NB=4 timeout 900 python -m pytest tests -m gpu -x -q -k "fused or convert_resize" 2>&1 | tail -2
VPF_LAB_FUSED_NB=3 timeout 900 python -m pytest tests -m gpu -x -q -k "fused or convert_resize" 2>&1 | tail -2
{ timeout 600 python tools/lab/ab/fused_nb.py 32; timeout 600 python tools/lab/ab/fused_nb.py 16; } 2>&1 | grep fused-nb | tee gpurun_out/r03_fused_bands_per_wave.txt
