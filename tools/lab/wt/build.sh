#!/bin/bash
# builds tools/lab/wt/libvpfhip_wt.so = the product library with the resize / Lanczos / fused TUs compiled under -DVPF_WAVE_TIMES (csrc/vpf_wave_times.h:
# one record per wave — start, duration on the 100 MHz constant clock, XCC / SE / CU / SIMD).  Same pixels; ~20 extra instructions per wave.
cd "$(dirname "$0")/../../.."
C=videoprocessingframework_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -fvisibility=hidden -Iinclude -I$C -DVPF_WAVE_TIMES $VPF_WT_EXTRA"
python -c "from videoprocessingframework_amd import _build; _build.build_kernels()"
rm -f /tmp/wt_k_*.o
for T in k_resize k_lanczos_mfma k_convert_resize; do hipcc $FLAGS -c $C/$T.hip -o /tmp/wt_$T.o 2>&1 | grep -E "error" -A4 & done
wait
for T in k_resize k_lanczos_mfma k_convert_resize; do [ -f /tmp/wt_$T.o ] || { echo "lab build FAILED: $T"; exit 1; }; done
OBJS=$(ls videoprocessingframework_amd/build/k_*.o videoprocessingframework_amd/build/vpf_abi.o | grep -v "k_resize.o\|k_lanczos_mfma.o\|k_convert_resize.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/wt/libvpfhip_wt${VPF_WT_SUFFIX}.so $OBJS /tmp/wt_k_resize.o /tmp/wt_k_lanczos_mfma.o /tmp/wt_k_convert_resize.o
ls -la tools/lab/wt/*.so
