#!/bin/bash
# round 6, visit a: the new 128-frame-dispatch parity tests; the two tools' protocols reconciled on one box; the first tables under bench.sustained
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 1500 python -m pytest tests/test_gpu_large_batches.py -m gpu -q -x 2>&1 | tail -25) > $O/r06_a_pytest_large_batches.txt; tail -3 $O/r06_a_pytest_large_batches.txt
(timeout 600 python tools/protocol_reconcile.py 2>&1 | grep reconcile) > $O/r06_a_protocol_reconcile.txt; cat $O/r06_a_protocol_reconcile.txt
(VPF_BENCH_Y=1 timeout 1200 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch\|remap") > $O/r06_a_resize_batch.txt; cut -c1-260 $O/r06_a_resize_batch.txt
(timeout 900 python tools/fused_scales_bench.py 2>&1 | grep fused) > $O/r06_a_fused_scales.txt; cat $O/r06_a_fused_scales.txt
timeout 600 python bench.py --extra > $O/r06_a_bench_extra.json 2> $O/r06_a_bench_extra.err; tail -c 3000 $O/r06_a_bench_extra.json
