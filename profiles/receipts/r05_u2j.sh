#!/bin/bash
# round 5, visit u2j: where a lone matrix-core Lanczos launch spends its wave lives (marks: setup done | first source tile through pass 1 | first destination tile stored)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD VPF_WT_LIB=$PWD/tools/lab/wt/libvpfhip_wt_marks.so
{
for n in 1 32; do
  timeout 300 python tools/wave_times.py lanczos RGB 3840 2160 1920 1080 --n $n 2>&1 | grep wave_times
  timeout 300 python tools/wave_times.py lanczos NV12 3840 2160 1920 1080 --n $n 2>&1 | grep wave_times
done
timeout 300 python tools/wave_times.py lanczos RGB 1920 1080 3840 2160 --n 32 2>&1 | grep wave_times
} > $O/r05_u2j_wave_marks_lanczos.txt; cut -c1-250 $O/r05_u2j_wave_marks_lanczos.txt
