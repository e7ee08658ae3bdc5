#!/bin/bash
# round 6, visit e: the persistent band launch, second form (chunks, static first chunk, prefetched tickets, the next chunk's rows requested before this chunk's last blend): parity, then the knob sweep at 32 frames per dispatch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "band or persistent or march" 2>&1 | tail -15) > $O/r06_e_pytest.txt; tail -6 $O/r06_e_pytest.txt
(SWEEP_PASSES=5 timeout 900 python tools/band_knob_sweep.py 0 0x10000 0x10400 0x10800 0x90100 0x90200 0x90400 0x90800 2>&1 | grep knobs) > $O/r06_e2_band_knobs_persist_v2_static.txt; cat $O/r06_e2_band_knobs_persist_v2_static.txt
