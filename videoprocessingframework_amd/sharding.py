"""Multi-GPU sharding of the surface path: independent clips / frame rings, one process per GPU, no data-path
collective.  The reference's only notion of multi-GPU is "pass a different gpu_id"
(CudaResMgr, src/PyNvCodec/src/PyNvCodec.cpp:57-111; per-thread pattern samples/SampleDecodeMultiThread.py:50-115);
here ranks meet only to agree on timing.  Works on the gloo backend (CPU tensors) and on nccl (= RCCL) alike.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment; (0, 1, 0) when absent."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def assign_clips(n_clips: int, world: int, rank: int) -> List[int]:
    """Clip s -> rank s mod N (config 4 of BASELINE.json: 8 streams, one per GPU)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_clips, world))


def init(backend: str | None = None, device: torch.device | None = None) -> bool:
    """Initialise the process group when launched with WORLD_SIZE > 1.  Returns True if distributed."""
    rank, world, _ = env_rank()
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return True


def barrier(device: torch.device | None = None) -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def aggregate(units_local: float, seconds_local: float, device: torch.device | None = None) -> Tuple[float, float]:
    """Whole-job (sum of units over ranks, max of elapsed time over ranks)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(units_local), float(seconds_local)
    dev = device if device is not None else torch.device("cpu")
    u = torch.tensor([units_local], dtype=torch.float64, device=dev)
    t = torch.tensor([seconds_local], dtype=torch.float64, device=dev)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(u.item()), float(t.item())


def gather(value_local: float, device: torch.device | None = None) -> List[float]:
    """One float per rank, in rank order (per-rank step times next to the max-over-ranks figure)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value_local)]
    dev = device if device is not None else torch.device("cpu")
    mine = torch.tensor([value_local], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def throughput(units_local: float, seconds_local: float, device: torch.device | None = None) -> float:
    u, t = aggregate(units_local, seconds_local, device)
    return u / t
