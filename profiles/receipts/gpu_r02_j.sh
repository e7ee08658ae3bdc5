#!/bin/bash
# round 2, visit J: the whole GPU suite, headline bench (+extra), kernel-trace stats, PMC traffic / SQ passes, secondary benches
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/r02_box.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/r02_box.txt
( time timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -40 ) > gpurun_out/r02_pytest_gpu.log 2>&1
timeout 300 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
timeout 600 python bench.py --extra --no-cpu > gpurun_out/r02_bench_extra.json 2> gpurun_out/r02_bench_extra.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc; bash tools/gpu_pmc.sh > gpurun_out/r02_gpu_pmc.log 2>&1
python tools/pmc_summary.py r02 > gpurun_out/r02_pmc_summary.txt 2>&1
cp profiles/r02_pmc_traffic.json profiles/r02_pmc_sq.json gpurun_out/ 2>/dev/null
timeout 600 python tools/resize_batch_bench.py > gpurun_out/r02_resize_batch.txt 2>&1
timeout 300 python tools/lanczos_bench.py > gpurun_out/r02_resize_modes.txt 2>&1
timeout 300 python tools/float_resize_bench.py >> gpurun_out/r02_resize_modes.txt 2>&1
timeout 300 python tools/resize_sizes_bench.py >> gpurun_out/r02_resize_modes.txt 2>&1
timeout 300 python tools/fused_scales_bench.py > gpurun_out/r02_fused_scales.txt 2>&1
timeout 600 python tools/secondary_bench.py > gpurun_out/r02_secondary_kernels.txt 2>&1
tail -12 gpurun_out/r02_pytest_gpu.log; cat gpurun_out/r02_bench_default.json; cat gpurun_out/r02_pmc_summary.txt; find gpurun_out/prof -name "*kernel_stats.csv" | head -2
