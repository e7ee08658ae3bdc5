#!/bin/bash
# round 3, visit au: per-wave phase timers of the matrix-core Lanczos kernel (instrumented lab build): where a wave's lifetime goes, 32 frames and 1 frame per dispatch
mkdir -p gpurun_out
for n in 32 1; do PROF_N=$n timeout 200 python tools/lab/ablate/prof_one.py tools/lab/ablate/libvpfhip_prof.so 2>&1 | grep prof; done | tee gpurun_out/r03au_prof.txt
