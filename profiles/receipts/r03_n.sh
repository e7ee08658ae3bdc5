#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma" 2>&1 | grep -E "Error|assert|FAILED|passed|failed" | head -12
python tools/lzm_debug.py 2>&1 | grep -v "^OK\|libvpfhip: launch\|amdgpu.ids" | head -40
