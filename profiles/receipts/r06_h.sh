#!/bin/bash
# round 6, visit h: the per-tap fused kernel as a row band (k_convert_resize_band): parity (policy and forced), then the fused table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_batches.py -m gpu -q -n 4 -k "fused or fuzz_resize_and_fused or tuning or convert_resize" 2>&1 | tail -15) > $O/r06_h_pytest.txt; tail -5 $O/r06_h_pytest.txt
(FUSED_VARIANTS=0,40 FUSED_PAIR=0 timeout 900 python tools/fused_scales_bench.py 2>&1 | grep fused) > $O/r06_h_fused_scales.txt; cut -c1-260 $O/r06_h_fused_scales.txt
