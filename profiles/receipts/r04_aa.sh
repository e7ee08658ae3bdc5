#!/bin/bash
# round 4, visit aa: row-band bilinear kernel as a march: short bands (4 / 2 rows), nb bands per wave with the next band's rows in flight and the walk's
# lerps kept across bands, 8 px per lane on 1-channel planes — against the product kernel (bl0, policy)
mkdir -p gpurun_out
{
echo "== product"; AB_PASSES=2 timeout 300 python tools/lab/ablate/time_bl.py tools/lab/ablate/libvpfhip_bl0.so 2>&1 | grep "\[bl\]"
for B in 4 2; do for NB in 1 2 4 8; do
  echo "== band rows $B, bands per wave $NB"
  VPF_BENCH_BAND=$B VPF_LAB_BAND_NB=$NB AB_PASSES=2 timeout 300 python tools/lab/ablate/time_bl.py tools/lab/ablate/libvpfhip_blm.so 2>&1 | grep "\[bl\] lib"
done; done
} | tee gpurun_out/r04aa_bilinear_march.txt
