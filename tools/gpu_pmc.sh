#!/bin/bash
# PMC pass: FETCH_SIZE and WRITE_SIZE in SEPARATE runs (TCC has 4 slots; never combined with trace domains other
# than kernel-trace), for the calibration kernels and for the headline bench.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc"
[ -x tools/pmc_calib.bin ] || hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib.bin
./tools/pmc_calib.bin > $OUT/calib_bw.txt 2>&1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -o c -- "$GRAFT_REPO_ROOT/tools/pmc_calib.bin" > /dev/null 2> $OUT/calib_$C.err
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/bench_$C -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu > $OUT/bench_$C.json 2> $OUT/bench_$C.err
done
cd "$GRAFT_REPO_ROOT"; cat $OUT/calib_bw.txt; ls -R $OUT | head -40
# SQ / L2 counters for the headline kernel (separate passes; SQ has 8 slots, TCC 4)
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/bench_SQ -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu > /dev/null 2> $OUT/bench_SQ.err
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/bench_TCC -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu > /dev/null 2> $OUT/bench_TCC.err
cd "$GRAFT_REPO_ROOT"; ls $OUT/bench_SQ $OUT/bench_TCC
