#!/bin/bash
# round 6, visit z: the round's evidence on the final code, every table under the sustained protocol (bench.sustained: 300 ms pre-heat of the same
# call, median of five >= 60 ms blocks, shader clock beside every number) — whole GPU suite, a fuzz soak over fresh seeds, the headline bench
# (+ rocprofv3 kernel stats, PMC traffic), the other configs, the resize / fused / secondary / chain / launch-rate / pipeline tables, the Lanczos
# kernel's counters and traffic, 32 against 128 frames per dispatch on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/r06_z_gpu.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $O/r06_z_gpu.txt
(timeout 2400 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -6) > $O/r06_z_pytest.txt; tail -2 $O/r06_z_pytest.txt
(VPF_FUZZ_SEEDS=4000 VPF_FUZZ_FIRST=600000 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -n 8 -k fuzz 2>&1 | tail -4) > $O/r06_z_fuzz_soak.txt; tail -1 $O/r06_z_fuzz_soak.txt
timeout 600 python bench.py > $O/r06_z_bench_default.json 2> $O/r06_z_bench_default.err; cut -c1-600 $O/r06_z_bench_default.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_z_prof -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu > $O/r06_z_prof_bench.json 2> $O/r06_z_prof.err
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r06_z_prof -name "*kernel_stats.csv" | head -2
timeout 600 python bench.py --extra --no-cpu > $O/r06_z_bench_extra.json 2> $O/r06_z_bench_extra.err; cut -c1-200 $O/r06_z_bench_extra.json
bash tools/gpu_pmc.sh > $O/r06_z_pmc.log 2>&1; tail -2 $O/r06_z_pmc.log
python tools/pmc_summary.py r06 > $O/r06_z_pmc_summary.log 2>&1; tail -3 $O/r06_z_pmc_summary.log
(VPF_BENCH_Y=1 timeout 1500 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\|remap") > $O/r06_z_resize_batch.txt; grep -c . $O/r06_z_resize_batch.txt
(FUSED_VARIANTS=0 timeout 900 python tools/fused_scales_bench.py 2>&1 | grep fused) > $O/r06_z_fused_scales.txt
(FUSED_N=128 FUSED_VARIANTS=0 timeout 900 python tools/fused_scales_bench.py 2>&1 | grep fused) > $O/r06_z_fused_scales_n128.txt; cut -c1-200 $O/r06_z_fused_scales.txt $O/r06_z_fused_scales_n128.txt
for n in 32 128; do for i in 1 2; do echo "== frames per dispatch $n, interp $i"; SWEEP_N=$n SWEEP_INTERP=$i timeout 400 python tools/band_knob_sweep.py 0 2>&1 | grep knobs; done; done > $O/r06_z_frames_per_dispatch_ab.txt; cat $O/r06_z_frames_per_dispatch_ab.txt
timeout 600 python tools/secondary_bench.py > $O/r06_z_secondary.txt 2>&1; tail -30 $O/r06_z_secondary.txt | cut -c1-200
timeout 400 python tools/chain_bench.py > $O/r06_z_chain.txt 2>&1; tail -6 $O/r06_z_chain.txt | cut -c1-300
gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/abi_launch_rate.c -o /tmp/abi_launch_rate -Lvideoprocessingframework_amd -lvpfhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/videoprocessingframework_amd -Wl,-rpath,/opt/rocm/lib && /tmp/abi_launch_rate > $O/r06_z_abi_launch_rate.txt 2>&1; cat $O/r06_z_abi_launch_rate.txt | cut -c1-200
timeout 400 python tools/pipeline_bench.py > $O/r06_z_pipeline_blocking.txt 2>&1; tail -8 $O/r06_z_pipeline_blocking.txt | cut -c1-200
timeout 400 python tools/pipeline_bench.py --async > $O/r06_z_pipeline_async.txt 2>&1; tail -8 $O/r06_z_pipeline_async.txt | cut -c1-200
for S in "1920 1080 1280 720" "3840 2160 1920 1080"; do T=$(echo $S | tr ' ' '_')
  bash tools/gpu_pmc_resize_batch.sh $S 2 > $O/r06_z_pmc_lanczos_mfma_$T.txt 2>&1; tail -3 $O/r06_z_pmc_lanczos_mfma_$T.txt | cut -c1-200
  bash tools/gpu_pmc_resize_traffic.sh $S 2 > $O/r06_z_pmc_lanczos_traffic_$T.txt 2>&1; tail -1 $O/r06_z_pmc_lanczos_traffic_$T.txt | cut -c1-300
done
bash tools/gpu_pmc_resize_traffic.sh 1920 1080 1280 720 1 > $O/r06_z_pmc_bilinear_traffic_1920_1080_1280_720.txt 2>&1; tail -1 $O/r06_z_pmc_bilinear_traffic_1920_1080_1280_720.txt | cut -c1-300
