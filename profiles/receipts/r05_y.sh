#!/bin/bash
# round 5, visit y: two frame-table sizes (32 / 128 frames): the suite, the per-call host cost from plain C, the sample chain, the small-plane tables
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -6) > $O/r05_y_pytest.txt; tail -2 $O/r05_y_pytest.txt
gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/abi_launch_rate.c -o /tmp/abi_launch_rate -Lvideoprocessingframework_amd -lvpfhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/videoprocessingframework_amd -Wl,-rpath,/opt/rocm/lib && /tmp/abi_launch_rate > $O/r05_y_abi_launch_rate.txt 2>&1; cat $O/r05_y_abi_launch_rate.txt | cut -c1-200
timeout 300 python tools/chain_bench.py > $O/r05_y_chain.txt 2>&1; tail -4 $O/r05_y_chain.txt | cut -c1-300
for n in 32 128; do for i in 1 2; do echo "== frames per dispatch $n, interp $i"; SWEEP_N=$n SWEEP_INTERP=$i timeout 300 python tools/band_knob_sweep.py 0 2>&1 | grep knobs; done; done > $O/r05_y_frames_per_dispatch_ab.txt; cat $O/r05_y_frames_per_dispatch_ab.txt
