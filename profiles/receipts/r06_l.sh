#!/bin/bash
# round 6, visit l: the lone matrix-core Lanczos launch.  (1) two lab variants measured at ONE frame per dispatch against
# the product, sustained protocol, interleaved: "early fetch" (the first two source tiles requested before the column operands: -DVPF_LZM_EARLY_FETCH,
# parity-checked first) and round 5's shared-column form (tools/lab/lanczos_shared_columns.patch on its own tree: four bands of one strip per workgroup,
# column operands once per workgroup in LDS — measured batched only until now); (2) where a lone launch's waves spend their lives (phase marks).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
P=videoprocessingframework_amd/libvpfhip.so; E=tools/lab/ab/libvpfhip_early_fetch.so; S=tools/lab/ab/libvpfhip_r05_shared_columns.so
timeout 200 python tools/lab/ab/pytest_with_lib.py $E tests/test_gpu_parity.py -q -x -p no:cacheprovider \
  -k "test_lanczos_mfma_kernel_shapes or test_lanczos_upscales_with_the_ring_of_two or test_lanczos_two_chunk or test_resize_lanczos3" 2>&1 | grep -v amdgpu.ids | tail -4 > $O/r06_l_early_fetch_parity.txt
cat $O/r06_l_early_fetch_parity.txt
{
for L in $P $E $S; do timeout 120 python tools/lab/ab/lone_lanczos.py $L 2>&1 | grep "\[lone\]"; done
for L in $S $E $P; do timeout 70 python tools/lab/ab/lone_lanczos.py $L --quick 2>&1 | grep "\[lone\]"; done
} > $O/r06_l_lone_lanczos_ab.txt
cut -c1-200 $O/r06_l_lone_lanczos_ab.txt
export VPF_WT_LIB=$PWD/tools/lab/wt/libvpfhip_wt_marks.so
{
timeout 60 python tools/wave_times.py lanczos RGB 3840 2160 1920 1080 --n 1 2>&1 | grep wave_times
timeout 60 python tools/wave_times.py lanczos NV12 3840 2160 1920 1080 --n 1 2>&1 | grep wave_times
timeout 60 python tools/wave_times.py lanczos RGB 1920 1080 3840 2160 --n 1 2>&1 | grep wave_times
timeout 60 python tools/wave_times.py lanczos RGB 3840 2160 1920 1080 --n 32 2>&1 | grep wave_times
} > $O/r06_l_wave_marks_lanczos.txt
cut -c1-260 $O/r06_l_wave_marks_lanczos.txt
