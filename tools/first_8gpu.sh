#!/bin/bash
# first_8gpu.sh — ONE command for the first node with more than one MI355X.  Nothing in this repo has ever run on a device other
# than 0 or with an RCCL-backed process group; this script is what to run the day such a node appears, and its control flow is rehearsed on CPU so
# that it cannot fail on syntax that day (tests/test_first_8gpu_rehearsal.py: `first_8gpu.sh --rehearse`, gloo ranks, host no-op steps).
#
#   tools/first_8gpu.sh [--rehearse] [--out DIR] [--gpus "1 2 4 8"]
#
# Steps (each appends to DIR/report.json; a failing step is recorded and the script goes on):
#   1. pytest -m gpu tests/test_gpu_multidevice.py tests/test_gpu_multirank.py      per-device parity (every visible device), 2-rank runs
#   2. bench.py --gpus N for N in 1 2 4 8 (torch.distributed.run, backend nccl = RCCL): one JSON line each, with the per-rank arrays
#      (device, PCI address, NUMA node, shader clock, own time) that make a bad curve diagnosable
#   3. tools/shard_pipeline.py at the largest N, frames from pinned and from pageable host memory (BASELINE config 4's runner: upload -> convert)
# The reference's model for all of this is "pass a different gpu_id" (src/PyNvCodec/src/PyNvCodec.cpp:57-111,
# samples/SampleDecodeMultiThread.py:50-115): independent clips, no collective on the data path.
# --rehearse: backend gloo, bench.py --rehearse-host (no GPU touched), steps 1 and 3 reduced to what a CPU can check (test collection,
# argument parsing).  No scaling number comes out of a rehearsal.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
export PYTHONPATH=$ROOT${PYTHONPATH:+:$PYTHONPATH}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
REHEARSE=0; OUT=$ROOT/gpurun_out/first_8gpu; GPUS="1 2 4 8"
while [ $# -gt 0 ]; do
  case "$1" in
    --rehearse) REHEARSE=1 ;;
    --out) OUT=$2; shift ;;
    --gpus) GPUS=$2; shift ;;
    *) echo "usage: $0 [--rehearse] [--out DIR] [--gpus \"1 2 4 8\"]" >&2; exit 2 ;;
  esac
  shift
done
mkdir -p "$OUT"; : > "$OUT/steps.jsonl"
note() { python - "$OUT/steps.jsonl" "$@" <<'PY'
import json, sys
path, step, rc, log = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
rec = {"step": step, "rc": rc, "log": log}
try:
    lines = [l for l in open(log, errors="replace").read().splitlines() if l.startswith("{")]
    if lines:
        rec["json"] = json.loads(lines[-1])
    else:
        rec["tail"] = open(log, errors="replace").read().splitlines()[-5:]
except OSError as e:
    rec["tail"] = [str(e)]
open(path, "a").write(json.dumps(rec) + "\n")
PY
}
port() { python -c 'import socket; s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])'; }
if [ $REHEARSE = 1 ]; then
  BACKEND=gloo; BENCH_EXTRA="--rehearse-host --steps 5 --warmup 2"; NDEV=0
else
  BACKEND=nccl; BENCH_EXTRA=""
  NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
  echo "[first_8gpu] $NDEV device(s) visible"
fi

# ---- 1. per-device parity, 2-rank runs
if [ $REHEARSE = 1 ]; then
  timeout 600 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_multirank.py --collect-only -q > "$OUT/pytest_multi.log" 2>&1
else
  timeout 3000 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_multirank.py -m gpu -q > "$OUT/pytest_multi.log" 2>&1
fi
note "pytest multidevice + multirank" $? "$OUT/pytest_multi.log"

# ---- 2. the scaling runs (the driver's own launch line)
LARGEST=1
for N in $GPUS; do
  if [ $REHEARSE = 0 ] && [ "$N" -gt "$NDEV" ]; then echo "[first_8gpu] skipping N=$N: $NDEV device(s)"; continue; fi
  LARGEST=$N
  if [ "$N" = 1 ]; then
    timeout 1800 python bench.py --gpus 1 $BENCH_EXTRA > "$OUT/bench_n1.log" 2>&1
  else
    timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$(port)" \
      bench.py --gpus "$N" --backend $BACKEND $BENCH_EXTRA > "$OUT/bench_n$N.log" 2>&1
  fi
  note "bench.py --gpus $N ($BACKEND)" $? "$OUT/bench_n$N.log"
done

# ---- 3. upload -> convert over independent clips at the largest N, pinned and pageable sources
for SRC in pinned pageable; do
  if [ $REHEARSE = 1 ]; then
    python tools/shard_pipeline.py --help > "$OUT/shard_$SRC.log" 2>&1 && \
      python - >> "$OUT/shard_$SRC.log" 2>&1 <<PY
import argparse, re, sys
text = open("$ROOT/tools/shard_pipeline.py").read()
for flag in ("--gpus", "--clips", "--frames", "--backend", "--source"):
    assert f'"{flag}"' in text, flag
assert "$SRC" in re.search(r'"--source", choices=\[([^\]]*)\]', text).group(1)
print("arguments ok")
PY
  elif [ "$LARGEST" = 1 ]; then
    timeout 1800 python tools/shard_pipeline.py --gpus 1 --clips 8 --frames 64 --source $SRC > "$OUT/shard_$SRC.log" 2>&1
  else
    timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$LARGEST" --master-addr 127.0.0.1 --master-port "$(port)" \
      tools/shard_pipeline.py --gpus "$LARGEST" --clips $((2 * LARGEST)) --frames 64 --backend $BACKEND --source $SRC > "$OUT/shard_$SRC.log" 2>&1
  fi
  note "shard_pipeline.py --gpus $LARGEST --source $SRC" $? "$OUT/shard_$SRC.log"
done

# ---- one report
python - "$OUT" $REHEARSE <<'PY'
import json, sys, os
out, rehearse = sys.argv[1], sys.argv[2] == "1"
steps = [json.loads(l) for l in open(os.path.join(out, "steps.jsonl"))]
rep = {"rehearsal": rehearse, "steps": steps, "failed": [s["step"] for s in steps if s["rc"] != 0]}
curve = {}
for s in steps:
    j = s.get("json") or {}
    if s["step"].startswith("bench.py") and "n_gpus" in j:
        curve[j["n_gpus"]] = {"value": j.get("value"), "unit": j.get("unit"), "ms_per_step": j.get("ms_per_step"), "per_rank_ms_per_step": j.get("per_rank_ms_per_step"),
                              "ranks": j.get("ranks")}
rep["bench"] = curve
json.dump(rep, open(os.path.join(out, "report.json"), "w"), indent=1)
print(f"[first_8gpu] {'REHEARSAL ' if rehearse else ''}report: {os.path.join(out, 'report.json')}; {len(steps)} steps, failed: {rep['failed'] or 'none'}")
sys.exit(1 if rep["failed"] else 0)
PY
