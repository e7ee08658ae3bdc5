#!/bin/bash
# round 4, visit ai: evidence on the final code: the whole GPU suite, the resize table (batched / per frame), the sample chain, SQ counters of the two-chunk
# Lanczos kernel (RGB 1080p -> 416 x 416) and of the march form (Y 1080p -> 720p bilinear), bench default + --extra, smoke
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -n 4 > gpurun_out/r04ai_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04ai_pytest.txt | cut -c1-300
VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\|remap_batch" > gpurun_out/r04ai_resize_batch.txt; grep -c "" gpurun_out/r04ai_resize_batch.txt
timeout 300 python tools/chain_bench.py > gpurun_out/r04ai_chain.txt 2>&1; grep chain gpurun_out/r04ai_chain.txt | cut -c1-330
bash tools/gpu_pmc_resize_batch.sh 1920 1080 416 416 2 > gpurun_out/r04ai_pmc_lanczos_two_chunk_1080_416.txt 2>&1; tail -3 gpurun_out/r04ai_pmc_lanczos_two_chunk_1080_416.txt | cut -c1-200
VPF_PMC_FMT=Y bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 1 > gpurun_out/r04ai_pmc_bilinear_march_Y.txt 2>&1; tail -2 gpurun_out/r04ai_pmc_bilinear_march_Y.txt | cut -c1-200
timeout 600 python bench.py > gpurun_out/r04ai_bench_default.json 2> gpurun_out/r04ai_bench_default.err; cut -c1-400 gpurun_out/r04ai_bench_default.json
timeout 600 python bench.py --extra > gpurun_out/r04ai_bench_extra.json 2> gpurun_out/r04ai_bench_extra.err; cut -c1-200 gpurun_out/r04ai_bench_extra.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
