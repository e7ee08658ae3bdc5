"""Multi-GPU sharding of the surface path: independent clips / frame rings, one process per GPU, no data-path
collective.  The reference's only notion of multi-GPU is "pass a different gpu_id"
(CudaResMgr, src/PyNvCodec/src/PyNvCodec.cpp:57-111; per-thread pattern samples/SampleDecodeMultiThread.py:50-115);
here ranks meet only to agree on timing.  Works on the gloo backend (CPU tensors) and on nccl (= RCCL) alike.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

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


def gather_objects(obj) -> list:
    """One picklable object per rank, in rank order (the per-rank diagnostics of bench.py's JSON line: device, PCI address, NUMA node, clocks).
    Off the timed path; gloo and nccl (= RCCL: the objects travel as byte tensors on the rank's current device) alike."""
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def throughput(units_local: float, seconds_local: float, device: torch.device | None = None) -> float:
    u, t = aggregate(units_local, seconds_local, device)
    return u / t


# ------------------------------------------------------------------------------------------------
# NUMA placement.  The end-to-end leg of a sharded pipeline is host-memory traffic: software decode writes frames, the uploader
# copies them into page-locked staging buffers, the DMA engine reads those.  On a two-socket 8-GPU node a rank whose threads and
# pinned buffers sit on the other socket pushes every byte across the inter-socket link twice.  One process per GPU makes the
# fix simple: before allocating anything, confine the rank to the CPUs of its GPU's NUMA node (first-touch then places the
# pinned buffers there as well).  The reference has no counterpart (it is one process with a thread per GPU,
# samples/SampleDecodeMultiThread.py:50-115).
# ------------------------------------------------------------------------------------------------
def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of sysfs local_cpulist)."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(cpus))


def pci_address(domain: int, bus: int, device: int, function: int = 0) -> str:
    return f"{domain:04x}:{bus:02x}:{device:02x}.{function:x}"


def gpu_numa_cpus(pci_addr: str, sysfs_pci: str = "/sys/bus/pci/devices") -> Tuple[Optional[int], List[int]]:
    """(NUMA node, CPUs local to it) of the PCI device, from sysfs; (None, []) when the platform does not say (single-node
    machines report node -1, containers may hide sysfs)."""
    base = os.path.join(sysfs_pci, pci_addr)
    try:
        node = int(open(os.path.join(base, "numa_node")).read().strip())
    except (OSError, ValueError):
        return None, []
    try:
        cpus = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
    except (OSError, ValueError):
        cpus = []
    return (node if node >= 0 else None), cpus


def current_sclk_mhz(pci_addr: Optional[str], sysfs_pci: str = "/sys/bus/pci/devices") -> Optional[int]:
    """The shader clock the amdgpu driver reports as current for this device (the starred level of pp_dpm_sclk), None where sysfs does not
    say (containers often hide it).  A rank that starts or ends a timed region on a lower clock than its neighbours explains a slow rank."""
    if not pci_addr:
        return None
    try:
        for line in open(os.path.join(sysfs_pci, pci_addr, "pp_dpm_sclk")).read().splitlines():
            if line.rstrip().endswith("*"):
                return int("".join(ch for ch in line.split(":", 1)[1] if ch.isdigit()))
    except (OSError, ValueError, IndexError):
        pass
    return None


def rank_identity(device_index: Optional[int], sysfs_pci: str = "/sys/bus/pci/devices") -> Dict[str, object]:
    """What a scaling run needs to be diagnosable from its own output: which device this rank drives, where it sits (PCI address, NUMA node)
    and which CPUs the rank may use.  device_index None = a host-only rehearsal."""
    rank, _, local = env_rank()
    info: Dict[str, object] = {"rank": rank, "local_rank": local, "device": device_index, "pci": None, "numa_node": None, "name": None,
                               "cpus": len(os.sched_getaffinity(0)), "pid": os.getpid()}
    if device_index is None:
        return info
    try:
        p = torch.cuda.get_device_properties(device_index)
        info["name"] = p.name
        info["pci"] = pci_address(int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))
        info["numa_node"] = gpu_numa_cpus(info["pci"], sysfs_pci)[0]
    except Exception as e:  # noqa: BLE001
        info["why"] = f"no PCI address: {e}"
    return info


def bind_to_gpu_numa(device_index: int, sysfs_pci: str = "/sys/bus/pci/devices") -> Dict[str, object]:
    """Confine this process to the CPUs local to GPU `device_index` (intersection with the affinity it already has).  Call it first in a
    rank, before pinned buffers are allocated.  sched_setaffinity(0, ...) moves the CALLING thread only, and asking torch for the PCI address
    has started runtime threads by then (HIP, OpenMP, gloo) — so the mask is applied to every thread the process has (/proc/self/task);
    threads created later inherit it.  Returns what was found and done; never raises for a missing sysfs."""
    info: Dict[str, object] = {"device": device_index, "numa_node": None, "bound": False}
    try:
        p = torch.cuda.get_device_properties(device_index)
        addr = pci_address(int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))
    except Exception as e:  # noqa: BLE001  (no GPU, or a torch build without the PCI fields)
        info["why"] = f"no PCI address: {e}"
        return info
    info["pci"] = addr
    node, cpus = gpu_numa_cpus(addr, sysfs_pci)
    info["numa_node"] = node
    allowed = os.sched_getaffinity(0)
    want = sorted(set(cpus) & allowed)
    if node is None or not want or set(want) == set(allowed):
        info["why"] = "platform reports no NUMA locality for this device" if node is None or not cpus else "already confined to the local CPUs"
        return info
    moved = _set_affinity_all_threads(want)
    info.update(bound=True, cpus=len(want), threads=moved)
    return info


def _set_affinity_all_threads(cpus, task_dir: str = "/proc/self/task") -> int:
    """sched_setaffinity for every thread of this process; returns how many were moved (a thread that exits meanwhile is skipped)"""
    try:
        tids = [int(t) for t in os.listdir(task_dir)]
    except (OSError, ValueError):
        tids = []
    moved = 0
    for tid in tids or [0]:
        try:
            os.sched_setaffinity(tid, cpus)
            moved += 1
        except OSError:
            pass
    if not moved:
        os.sched_setaffinity(0, cpus)
        moved = 1
    return moved
