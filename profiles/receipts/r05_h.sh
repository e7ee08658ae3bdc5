#!/bin/bash
# round 5, visit h: the suite after the scalar row step in the Lanczos staging loads (no spills) and the unified planner model; the full resize table;
# single-frame Lanczos timeline; the shard pipeline's host-memory account (pinned / pageable sources)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests -m gpu -q -x --maxfail=3 2>&1 | tail -6) > $O/r05_h_pytest.txt; tail -3 $O/r05_h_pytest.txt
(VPF_BENCH_Y=1 timeout 600 python tools/resize_batch_bench.py 2>&1 | grep resize_batch) > $O/r05_h_resize_batch.txt; cut -c1-150 $O/r05_h_resize_batch.txt
for spec in "lanczos RGB 3840 2160 1920 1080 --n 1" "lanczos RGB 1920 1080 1280 720 --n 1 --mfma 0x40000" "lanczos NV12 3840 2160 1920 1080 --n 1" "lanczos RGB 3840 2160 1920 1080" "lanczos RGB 1280 720 1920 1080 --n 128"; do
  timeout 120 python tools/wave_times.py $spec 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-400
done > $O/r05_h_wave_times.txt; cat $O/r05_h_wave_times.txt
for src in pinned pageable; do timeout 300 python tools/shard_pipeline.py --clips 8 --frames 48 --source $src --threads 2>/dev/null | tail -1; done > $O/r05_h_shard_pipeline.txt; cut -c1-1500 $O/r05_h_shard_pipeline.txt
