#!/bin/bash
# round 2, visit A: the N>1 path on one GPU (2 gloo ranks) + the shard pipeline runner + default bench
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multirank.py -q -x 2>&1 | tail -40 > gpurun_out/r02_multirank_pytest.log
{
echo "# two ranks of bench.py sharing ONE MI355X over gloo (the N>1 control flow on hardware; not a scaling measurement)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu 2>&1 | grep -v Gloo | tail -3
echo "# one rank, same box"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu | tail -1
echo "# tools/shard_pipeline.py: 8 4K clips on 2 ranks sharing the GPU, then on 1 rank"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 tools/shard_pipeline.py --gpus 2 --backend gloo --clips 8 --frames 48 2>&1 | grep -v Gloo | tail -2
timeout 300 python tools/shard_pipeline.py --clips 8 --frames 48 | tail -1
} > gpurun_out/r02_two_ranks_one_gpu.txt 2>&1
cat gpurun_out/r02_multirank_pytest.log; cat gpurun_out/r02_two_ranks_one_gpu.txt
