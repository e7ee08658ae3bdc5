#!/bin/bash
# round 6, visit c: px4 LDS strips (packed RGB widened to R G B x) in the fused workgroup strips and in the row-band resize: parity, then the two tables
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_batches.py tests/test_gpu_pynvcodec.py -m gpu -q -n 4 -k "resize or fused or fuzz or batch or band or chain or graph" 2>&1 | tail -15) > $O/r06_c_pytest.txt; tail -4 $O/r06_c_pytest.txt
(timeout 900 python tools/fused_scales_bench.py 2>&1 | grep fused) > $O/r06_c_fused_scales.txt; cut -c1-230 $O/r06_c_fused_scales.txt
(VPF_BENCH_ONLY=bilinear timeout 900 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch.*RGB") > $O/r06_c_resize_batch_bilinear_rgb.txt; cut -c1-230 $O/r06_c_resize_batch_bilinear_rgb.txt
