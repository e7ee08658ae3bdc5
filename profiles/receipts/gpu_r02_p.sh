#!/bin/bash
# round 2, visit P: tiled Lanczos with replicated edge margins: parity + the single-frame lines
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pynvcodec.py -q -x -k "lanczos or resize or tiled or fuzz_resize" -n 4 2>&1 | tail -4 > gpurun_out/r02_p_pytest.txt
timeout 300 python tools/lanczos_bench.py > gpurun_out/r02_p_modes.txt 2>&1
VPF_BENCH_MARCH=1 VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | cut -c1-200 >> gpurun_out/r02_p_modes.txt
cat gpurun_out/r02_p_pytest.txt gpurun_out/r02_p_modes.txt
