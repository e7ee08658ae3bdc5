"""N>1 path on CPU: world_size-2 gloo processes exercise clip sharding and the sum-units / max-time reduction that
bench.py uses (rendezvous on 127.0.0.1)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from videoprocessingframework_amd import sharding

    assert sharding.init("gloo")
    clips = sharding.assign_clips(8, world, rank)
    sharding.barrier()
    units, secs = sharding.aggregate(units_local=100.0 * len(clips), seconds_local=1.0 + rank)
    tp = sharding.throughput(100.0 * len(clips), 1.0 + rank)
    q.put((rank, clips, units, secs, tp))
    sharding.barrier()
    torch.distributed.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]      # clip s -> rank s mod N, disjoint and complete
    for _, _, units, secs, tp in res:
        assert units == 800.0 and secs == 2.0 and tp == 400.0            # sum of units / MAX time over ranks


def test_single_process_passthrough():
    from videoprocessingframework_amd import sharding

    assert sharding.env_rank()[1] >= 1
    assert sharding.assign_clips(5, 1, 0) == [0, 1, 2, 3, 4]
    assert sharding.aggregate(10.0, 2.0) == (10.0, 2.0)
    with pytest.raises(ValueError):
        sharding.assign_clips(5, 2, 2)
