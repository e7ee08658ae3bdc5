#!/bin/bash
# builds libvpfhip_x<N>.so = the product library with k_lanczos_mfma.hip compiled under -DVPF_LZM_X=N (timing ablations; wrong pixels)
cd "$(dirname "$0")/../../.."
C=videoprocessingframework_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -fvisibility=hidden -Iinclude -I$C"
OBJS=$(ls videoprocessingframework_amd/build/k_*.o videoprocessingframework_amd/build/vpf_abi.o | grep -v k_lanczos_mfma)
for X in "$@"; do
  if [ "$X" = prof ]; then hipcc $FLAGS -DVPF_LZM_PROF=1 -c $C/k_lanczos_mfma.hip -o /tmp/lzm_xprof.o 2>&1 | grep -v warning & continue; fi
  hipcc $FLAGS -DVPF_LZM_X=$X -c $C/k_lanczos_mfma.hip -o /tmp/lzm_x$X.o 2>&1 | grep -v warning &
done
wait
for X in "$@"; do hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/ablate/libvpfhip_x$X.so $OBJS /tmp/lzm_x$X.o; done
ls -la tools/lab/ablate/
