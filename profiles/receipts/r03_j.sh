#!/bin/bash
# round 3, visit j: asynchronous staged uploads — the whole GPU suite, the clip runner against the stub decoder, and the sharded pipeline
# (config 4's runner) with pinned / pageable sources, one thread round-robin / one thread per clip
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r03j_pytest.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03j_pytest.txt
{ echo "# tools/shard_pipeline.py, 8 4K clips x 64 frames on ONE MI355X (one rank), round 3: staged (pageable) uploads return once the DMA is queued, 4 slots"
  lscpu | grep -E "Model name|^CPU\(s\)|NUMA node\(s\)"; echo "usable CPUs: $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
  for src in pinned pageable; do for thr in "" "--threads"; do
    echo "## --source $src $thr"; timeout 300 python tools/shard_pipeline.py --source $src $thr 2>&1 | tail -1
  done; done
  echo "## VPF_HIP_UPLOAD_SYNC=1 --source pageable --threads (the blocking upload of rounds 1-2)"; VPF_HIP_UPLOAD_SYNC=1 timeout 300 python tools/shard_pipeline.py --source pageable --threads 2>&1 | tail -1
  echo "## tools/pipeline_bench.py"; timeout 300 python tools/pipeline_bench.py 2>&1 | tail -12; timeout 300 python tools/pipeline_bench.py --pinned 2>&1 | tail -8
} > gpurun_out/r03_pipeline.txt 2>&1
cat gpurun_out/r03_pipeline.txt
