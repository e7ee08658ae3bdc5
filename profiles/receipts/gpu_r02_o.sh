#!/bin/bash
# round 2, visit O: straight-numbered kernels (p16x, planar r16x): suite + same-box A/B against the chunk-per-row forms
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 -n 4 2>&1 | tail -4 ) > gpurun_out/r02_o_pytest.txt
{ for i in 1 2 3; do for v in 37 47; do timeout 300 python bench.py --no-cpu --workload nv12_planar_1080p --ring 128 --variant $v | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1080p NV12->RGB_PLANAR batched, variant', d['config']['variant'], d['value'], 'Gpix/s', d['roofline']['frac'])"; done; done; } > gpurun_out/r02_o_ab.txt 2>&1
cat gpurun_out/r02_o_pytest.txt gpurun_out/r02_o_ab.txt
