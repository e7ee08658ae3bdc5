#!/bin/bash
# round 2, visit D: Lanczos with the integer (v_dot2_i32_i16) horizontal pass: parity, per-frame times, kernel durations + SQ counters
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or resize or fuzz_resize" 2>&1 | tail -15 ) > gpurun_out/r02_d_pytest.log 2>&1
timeout 300 python tools/lanczos_bench.py > gpurun_out/r02_d_lanczos.txt 2>&1
bash tools/gpu_pmc_resize.sh 1920 1080 1280 720 2 > gpurun_out/r02_d_pmc_1080_720.txt 2>&1
bash tools/gpu_pmc_resize.sh 1920 1080 3840 2160 2 > gpurun_out/r02_d_pmc_1080_4k.txt 2>&1
cat gpurun_out/r02_d_pytest.log; grep -v amdgpu.ids gpurun_out/r02_d_lanczos.txt; cat gpurun_out/r02_d_pmc_1080_720.txt; cat gpurun_out/r02_d_pmc_1080_4k.txt
