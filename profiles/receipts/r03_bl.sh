#!/bin/bash
# round 3, visit bl: tools/shard_pipeline.py (8 4K clips, one rank) with and without the non-temporal staging copy, one thread and one thread per clip
mkdir -p gpurun_out
{ for nt in 0 1; do for thr in "" "--threads"; do
    echo "## VPF_HIP_NT_COPY=$nt --source pageable $thr"; VPF_HIP_NT_COPY=$nt timeout 300 python tools/shard_pipeline.py --source pageable $thr 2>&1 | tail -1
  done; done; } | tee gpurun_out/r03_shard_pipeline_nt_copy.txt | cut -c1-700
