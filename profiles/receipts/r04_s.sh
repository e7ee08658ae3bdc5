#!/bin/bash
# round 4, visit s: per-frame resizes split into kernel time and inter-kernel gap (rocprofv3 kernel trace)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sf -o s -- python $R/tools/single_frame_trace.py > $R/gpurun_out/r04s_run.txt 2>&1; echo "trace rc $?"
cd $R
python tools/single_frame_trace.py --report gpurun_out/sf | tee gpurun_out/r04s_single_frame.txt
cp $(find gpurun_out/sf -name "*kernel_trace.csv" | head -1) gpurun_out/r04s_kernel_trace.csv; rm -rf gpurun_out/sf
