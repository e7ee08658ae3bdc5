#!/bin/bash
# (apply tools/lab/bilinear_band_ablation_hooks.patch first: the product source carries no lab hooks)
# builds tools/lab/ablate/libvpfhip_bl<N>.so = the product library with k_resize.hip compiled under -DVPF_BL_X=N (timing ablations of the row-band
# bilinear kernel: 1 no blend, 2 no staging, 3 neither; wrong pixels) — or, for N = a file name ending in .hip, with that file in k_resize.hip's place
cd "$(dirname "$0")/../../.."
C=videoprocessingframework_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -fvisibility=hidden -Iinclude -I$C"
OBJS=$(ls videoprocessingframework_amd/build/k_*.o videoprocessingframework_amd/build/vpf_abi.o | grep -v "k_resize.o")
for X in "$@"; do D="-DVPF_BL_X=$X"; [ "$X" = nowin ] && D="-DVPF_BL_NOWIN=1"; hipcc $FLAGS $D -c $C/k_resize.hip -o /tmp/bl_x$X.o 2>&1 | grep -v warning & done
wait
for X in "$@"; do hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/ablate/libvpfhip_bl$X.so $OBJS /tmp/bl_x$X.o; done
ls -la tools/lab/ablate/*.so
