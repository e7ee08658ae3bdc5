#!/bin/bash
# round 6, visit z2: the remaining rows of DESIGN §4.4 that still read round-4 files — the two-chunk / half-tile Lanczos sweeps against the tile kernel and the
# two-chunk kernel's counters — once more under the sustained protocol
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
(SWEEP_R="4" timeout 1500 python tools/lanczos_k2_sweep.py 2>&1 | grep "lz-k2") > $O/r06_z2_lanczos_two_chunk_sweep.txt; cut -c1-200 $O/r06_z2_lanczos_two_chunk_sweep.txt
(SWEEP_THUMBS=1 SWEEP_R="4" timeout 1500 python tools/lanczos_k2_sweep.py 2>&1 | grep "lz-k2") > $O/r06_z2_lanczos_half_tiles_sweep.txt; cut -c1-200 $O/r06_z2_lanczos_half_tiles_sweep.txt
bash tools/gpu_pmc_resize_batch.sh 1920 1080 416 416 2 > $O/r06_z2_pmc_lanczos_two_chunk_1080_416.txt 2>&1; tail -4 $O/r06_z2_pmc_lanczos_two_chunk_1080_416.txt | cut -c1-200
timeout 600 python bench.py > $O/r06_z2_bench_default.json 2> $O/r06_z2_bench_default.err; cut -c1-300 $O/r06_z2_bench_default.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_z2_prof -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu > $O/r06_z2_prof_bench.json 2> $O/r06_z2_prof.err
cd "$GRAFT_REPO_ROOT"; (timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -4) > $O/r06_z2_pytest.txt; tail -1 $O/r06_z2_pytest.txt
