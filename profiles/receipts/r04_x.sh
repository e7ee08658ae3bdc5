#!/bin/bash
# round 4, visit x: the single-frame Lanczos rule in place: whole GPU suite, the resize table (batched / per frame), the sample chain, smoke
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -n 4 > gpurun_out/r04x_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04x_pytest.txt | cut -c1-300
VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r04x_resize_batch.txt; grep -c resize_batch gpurun_out/r04x_resize_batch.txt
timeout 300 python tools/chain_bench.py > gpurun_out/r04x_chain.txt 2>&1; grep chain gpurun_out/r04x_chain.txt | cut -c1-330
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
