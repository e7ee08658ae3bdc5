#!/bin/bash
# round 3, visit af: the blocking tasks poll the stream before they sleep on it: the reference's sample chain as written, upload / download rates, API tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pynvcodec.py tests/test_gpu_reference_PySurface.py -q -x > gpurun_out/r03af_pytest.txt 2>&1; tail -3 gpurun_out/r03af_pytest.txt
timeout 300 python tools/chain_bench.py 2>&1 | grep chain | tee gpurun_out/r03af_chain.txt
VPF_HIP_SYNC_SPIN_US=0 timeout 300 python tools/chain_bench.py 2>&1 | grep chain | sed 's/^/[no spin] /' | tee -a gpurun_out/r03af_chain.txt
timeout 300 python tools/shard_pipeline.py --clips 8 --frames 48 --source pageable 2>&1 | tail -1 | cut -c1-600
timeout 300 python tools/shard_pipeline.py --clips 8 --frames 48 --source pinned 2>&1 | tail -1 | cut -c1-600
timeout 300 python tools/pipeline_bench.py 2>&1 | tail -12
