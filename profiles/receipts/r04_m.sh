#!/bin/bash
# round 4, visit m: the round's evidence for what round 4 did not change in the kernels — headline bench + rocprofv3 kernel stats + PMC traffic of the
# headline kernel, secondary converters, fused scale sweep, sample chain, shard pipeline with the (again) blocking and the asynchronous uploader
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python bench.py --extra > gpurun_out/r04m_bench_extra.json 2> gpurun_out/r04m_bench_extra.err; cut -c1-300 gpurun_out/r04m_bench_extra.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04m_prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/r04m_prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r04m_prof.err"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r04m_prof -name "*kernel_stats.csv" | head -2
bash tools/gpu_pmc.sh > gpurun_out/r04m_pmc.log 2>&1; tail -3 gpurun_out/r04m_pmc.log
timeout 300 python tools/secondary_bench.py > gpurun_out/r04m_secondary.txt 2>&1; tail -30 gpurun_out/r04m_secondary.txt
timeout 300 python tools/fused_scales_bench.py 2>&1 | grep fused > gpurun_out/r04m_fused_scales.txt; cat gpurun_out/r04m_fused_scales.txt
timeout 300 python tools/chain_bench.py > gpurun_out/r04m_chain.txt 2>&1; tail -6 gpurun_out/r04m_chain.txt | cut -c1-300
timeout 300 python tools/shard_pipeline.py --source pageable > gpurun_out/r04m_shard_pageable.txt 2>&1; tail -2 gpurun_out/r04m_shard_pageable.txt | cut -c1-400
timeout 300 python tools/pipeline_bench.py > gpurun_out/r04m_pipeline_blocking.txt 2>&1; tail -8 gpurun_out/r04m_pipeline_blocking.txt | cut -c1-200
timeout 300 python tools/pipeline_bench.py --async > gpurun_out/r04m_pipeline_async.txt 2>&1; tail -8 gpurun_out/r04m_pipeline_async.txt | cut -c1-200
