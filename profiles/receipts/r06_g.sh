#!/bin/bash
# round 6, visit g: caller frame buffers page-locked on second sight (HostPinCache): tests, then tools/shard_pipeline.py with pinned and pageable sources (8 clips, threads)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; rm -f $O/r06_g_shard_pipeline_host_memory.txt
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests/test_gpu_pin_cache.py tests/test_gpu_pynvcodec.py tests/test_clip_pipeline.py -m gpu -q 2>&1 | tail -15) > $O/r06_g_pytest.txt; tail -8 $O/r06_g_pytest.txt
for SRC in pinned pageable; do
  for T in "" "--threads"; do
    (timeout 600 python tools/shard_pipeline.py --clips 8 --frames 48 --source $SRC $T 2>&1 | grep runner) >> $O/r06_g_shard_pipeline_host_memory.txt
  done
done
for T in "" "--threads"; do
  (VPF_HIP_PIN_CACHE_MB=0 timeout 600 python tools/shard_pipeline.py --clips 8 --frames 48 --source pageable $T 2>&1 | grep runner | sed 's/^/[VPF_HIP_PIN_CACHE_MB=0] /') >> $O/r06_g_shard_pipeline_host_memory.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_g_shard_pipeline_host_memory.txt"):
    tag = ""
    if l.startswith("["):
        tag, l = l.split("] ", 1); tag += "] "
    j = json.loads(l)
    hm = j["host_memory"]
    print(f"{tag}{hm['source']:8s} {j['threads']:22s} end to end {j['end_to_end']['frames_per_s']:7.0f} frames/s = {j['bytes_per_s_end_to_end']:5.1f} GB/s; host DRAM bytes per frame {hm['host_dram_bytes_per_frame'] / 1e6:5.1f} MB ({hm['staged_rank0']} of {hm['uploads_rank0']} uploads staged); pin cache {hm['pin_cache_rank0']}")
PY
