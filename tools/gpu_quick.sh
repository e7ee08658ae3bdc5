#!/bin/bash
# quick GPU visit: all gpu tests + the resize/fused lines of the sweep + default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -25
timeout 300 python - <<'PY'
import sys, torch
sys.argv=['bench']
import bench
dev=torch.device('cuda',0)
for name,ring,steps,mode in (("resize_4k_720p",8,6,"single"),("fused_4k_720p",8,6,"single"),("fused_4k_720p",32,6,"batch"),("fused_4k_720p",64,6,"batch")):
    wl=bench.Workload(name,dev,ring,0,mode)
    _,ev=bench.timed(wl,steps,2,False)
    print(f"[quick] {name:16s} {mode:6s} ring {ring}: {wl.px_per_step*steps/ev/1e9:8.1f} Gpix/s(src) {wl.bytes_per_step*steps/ev/1e9:7.0f} GB/s algorithmic, {ev/steps/ring*1e6:.2f} us/frame")
PY
timeout 300 python bench.py 2>&1 | tail -1
