#!/bin/bash
# round 5, visit z: the round's evidence on the final code — whole GPU suite, a fuzz soak over fresh seeds, the headline bench (+ rocprofv3 kernel
# stats, PMC traffic), the other configs, the resize / fused / secondary / chain / launch-rate / pipeline tables, same-box A/B of 32 against 128 frames per dispatch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -6) > $O/r05_z_pytest.txt; tail -2 $O/r05_z_pytest.txt
(VPF_FUZZ_SEEDS=6000 VPF_FUZZ_FIRST=300000 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -n 8 -k fuzz 2>&1 | tail -4) > $O/r05_z_fuzz_soak.txt; tail -1 $O/r05_z_fuzz_soak.txt
timeout 600 python bench.py > $O/r05_z_bench_default.json 2> $O/r05_z_bench_default.err; cut -c1-500 $O/r05_z_bench_default.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_z_prof -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu > $O/r05_z_prof_bench.json 2> $O/r05_z_prof.err
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r05_z_prof -name "*kernel_stats.csv" | head -2
timeout 400 python bench.py --extra --no-cpu > $O/r05_z_bench_extra.json 2> $O/r05_z_bench_extra.err; cut -c1-200 $O/r05_z_bench_extra.json
bash tools/gpu_pmc.sh > $O/r05_z_pmc.log 2>&1; tail -2 $O/r05_z_pmc.log
(VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\|remap") > $O/r05_z_resize_batch.txt; grep -c . $O/r05_z_resize_batch.txt
(FUSED_VARIANTS=0,47 timeout 400 python tools/fused_scales_bench.py 2>&1 | grep fused) > $O/r05_z_fused_scales.txt
(FUSED_N=128 FUSED_VARIANTS=0 timeout 400 python tools/fused_scales_bench.py 2>&1 | grep fused) > $O/r05_z_fused_scales_n128.txt; cut -c1-300 $O/r05_z_fused_scales.txt $O/r05_z_fused_scales_n128.txt
for n in 32 128; do for i in 1 2; do echo "== frames per dispatch $n, interp $i"; SWEEP_N=$n SWEEP_INTERP=$i timeout 300 python tools/band_knob_sweep.py 0 2>&1 | grep knobs; done; done > $O/r05_z_frames_per_dispatch_ab.txt; cat $O/r05_z_frames_per_dispatch_ab.txt
timeout 400 python tools/secondary_bench.py > $O/r05_z_secondary.txt 2>&1; tail -30 $O/r05_z_secondary.txt | cut -c1-200
timeout 300 python tools/chain_bench.py > $O/r05_z_chain.txt 2>&1; tail -6 $O/r05_z_chain.txt | cut -c1-300
gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/abi_launch_rate.c -o /tmp/abi_launch_rate -Lvideoprocessingframework_amd -lvpfhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/videoprocessingframework_amd -Wl,-rpath,/opt/rocm/lib && /tmp/abi_launch_rate > $O/r05_z_abi_launch_rate.txt 2>&1; cat $O/r05_z_abi_launch_rate.txt | cut -c1-200
timeout 300 python tools/pipeline_bench.py > $O/r05_z_pipeline_blocking.txt 2>&1; tail -8 $O/r05_z_pipeline_blocking.txt | cut -c1-200
timeout 300 python tools/pipeline_bench.py --async > $O/r05_z_pipeline_async.txt 2>&1; tail -8 $O/r05_z_pipeline_async.txt | cut -c1-200
