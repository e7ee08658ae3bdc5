#!/bin/bash
# round 3, visit bd: final evidence on the final code: bench.py default + --extra, rocprofv3 kernel stats of the bench command, secondary converters, box id
mkdir -p gpurun_out
python bench.py > gpurun_out/r03bd_bench_default.json 2> gpurun_out/r03bd_bench_default.err; tail -c 1500 gpurun_out/r03bd_bench_default.json
python bench.py --extra --no-cpu > gpurun_out/r03bd_bench_extra.json 2>/dev/null; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03bd_bench_extra.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
for k, v in d.get("other_configs", {}).items():
    print(" ", k, {a: v[a] for a in v if a in ("value", "unit", "us_per_frame", "frac_of_8TBs", "kernel")})
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03bd_prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --repeats 1 > $GRAFT_REPO_ROOT/gpurun_out/r03bd_prof.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r03bd_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03bd_kernel_stats.csv; head -4 gpurun_out/r03bd_kernel_stats.csv | cut -c1-200; tail -1 gpurun_out/r03bd_prof.log | cut -c1-400
timeout 600 python tools/secondary_bench.py 2>&1 | tail -30 > gpurun_out/r03bd_secondary.txt; tail -5 gpurun_out/r03bd_secondary.txt
timeout 900 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\|remap_batch" > gpurun_out/r03bd_resize_batch.txt; grep -c . gpurun_out/r03bd_resize_batch.txt
timeout 300 python tools/chain_bench.py 2>&1 | tail -12 > gpurun_out/r03bd_chain.txt; tail -6 gpurun_out/r03bd_chain.txt
