#!/bin/bash
# round 4, visit z: row-band bilinear kernel with nb bands per wave (next band's rows requested while this band is blended): nb = 1..4 against the product kernel
mkdir -p gpurun_out
for NB in 1 2 3 4; do
  echo "== bands per wave $NB"
  VPF_LAB_BAND_NB=$NB AB_PASSES=2 timeout 600 python tools/lab/ablate/time_bl.py tools/lab/ablate/libvpfhip_bl0.so tools/lab/ablate/libvpfhip_blm.so 2>&1 | grep "\[bl\]"
done | tee gpurun_out/r04z_bilinear_bands_per_wave.txt
