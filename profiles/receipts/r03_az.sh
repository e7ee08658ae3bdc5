#!/bin/bash
# round 3, visit az: "truncate + pack" as v_cvt_pk_u8_f32 under round-toward-zero — probe, parity suite, the benches of every kernel family that packs that way
mkdir -p gpurun_out
timeout 120 tools/lab/probes/probe_rtz_pack 2>&1 | tee gpurun_out/r03_probe_rtz_pack.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r03az_pytest.txt
VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep -v "^$" | tee gpurun_out/r03az_resize_batch_bilinear.txt
timeout 600 python tools/fused_scales_bench.py 2>&1 | tee gpurun_out/r03az_fused_scales.txt
timeout 600 python tools/secondary_bench.py 2>&1 | tee gpurun_out/r03az_secondary.txt
