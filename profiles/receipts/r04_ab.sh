#!/bin/bash
# round 4, visit ab: the multi-band row-band forms on every plane kind (invalid combinations now fail instead of covering 1 / nb of the picture)
mkdir -p gpurun_out
{
echo "== product"; AB_PASSES=2 timeout 300 python tools/lab/ablate/time_bl.py tools/lab/ablate/libvpfhip_bl0.so 2>&1 | grep "\[bl\]"
for B in 4 8; do for NB in 1 2 3; do
  echo "== band rows $B, bands per wave $NB"
  VPF_BENCH_BAND=$B VPF_LAB_BAND_NB=$NB AB_PASSES=2 timeout 300 python tools/lab/ablate/time_bl.py tools/lab/ablate/libvpfhip_blm.so 2>&1 | grep "\[bl\] lib" | cut -c1-400
done; done
} | tee gpurun_out/r04ab_bilinear_march.txt
