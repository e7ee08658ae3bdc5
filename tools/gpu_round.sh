#!/bin/bash
# one GPU-box visit: parity tests, variant sweep, headline bench, kernel trace
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/gpu.txt
lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/gpu.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --sweep --steps 20 --warmup 3 --no-cpu > gpurun_out/bench_sweep.json 2> gpurun_out/bench_sweep.log
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
cd "$GRAFT_REPO_ROOT"; ls -R gpurun_out/prof | head -30
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/bench_sweep.log | tail -40; cat gpurun_out/bench_default.json
