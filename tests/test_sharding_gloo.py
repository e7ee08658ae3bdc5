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


def _run_bench(args, timeout=300):
    import json
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line, from rank 0 only: {lines}"
    return json.loads(lines[0])


@pytest.mark.parametrize("launcher", ["driver", "self"])
def test_bench_multi_rank_control_flow_rehearsal(launcher):
    """bench.py's own N>1 control flow (rendezvous on 127.0.0.1, LOCAL_RANK handling, the barriers around the timed
    region, sum-units / max-time reduction, one JSON line from rank 0) executed for real with 2 gloo ranks; the step is
    a host no-op (--rehearse-host), so no GPU is needed and nothing is measured.  'driver' = the launch line the driver
    uses; 'self' = `python bench.py --gpus 2` re-executing itself under torch.distributed.run."""
    tail = ["--gpus", "2", "--steps", "5", "--warmup", "2", "--rehearse-host"]
    if launcher == "driver":
        out = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), "bench.py", *tail])
    else:
        out = _run_bench(["bench.py", *tail])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["data"] == "rehearsal" and out["value"] is None and out["roofline"] is None
    per_rank = out["per_rank_ms_per_step"]
    assert len(per_rank) == 2 and per_rank[1] > 1.25 * per_rank[0]          # rank 1's step sleeps twice as long (a loaded host stretches both sleeps by the same absolute amount)
    assert out["ms_per_step"] >= per_rank[1] * 0.99                          # MAX over ranks, not the mean ...
    # ... of each rank's OWN launch -> synchronize time: the closing barrier's cost is reported beside it, not inside it.  Rank 1's step
    # takes 4 ms against rank 0's 2, so the slow rank's own time is the job's time to within scheduling noise, and the barrier figure
    # (how long the slowest rank itself waited in the closing collective) stays small against the 20 ms block
    assert out["ms_per_step"] <= per_rank[1] * 1.05 and "barrier_ms" in out and 0.0 <= out["barrier_ms"] < 10.0
    px = 2 * 32 * 3840 * 2160 * 5                                            # both ranks' units are summed
    assert abs(out["rehearsal_units_per_s"] - px / (out["ms_per_step"] * 5e-3)) / out["rehearsal_units_per_s"] < 0.01


def test_bench_eight_rank_rehearsal_carries_per_rank_diagnostics():
    """The first 8-GPU run must be diagnosable from its own output: bench.py's control flow with EIGHT gloo ranks — the
    driver's launch line at the size of a full node — prints one JSON line whose `ranks` array says, per rank, which device it drove, its
    PCI address and NUMA node, the shader clock before and after, its own host-clock time and its HIP-event time per step.  (Host rehearsal:
    device / pci / clocks are null here; the fields and the reduction are what is checked.)"""
    out = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                      "bench.py", "--gpus", "8", "--steps", "4", "--warmup", "1", "--repeats", "3", "--rehearse-host"], timeout=600)
    assert out["n_gpus"] == 8 and out["data"] == "rehearsal" and len(out["per_rank_ms_per_step"]) == 8
    ranks = out["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8)) and len({r["pid"] for r in ranks}) == 8
    for r in ranks:
        for key in ("device", "pci", "numa_node", "sclk_mhz_start", "sclk_mhz_end", "own_ms_per_step", "event_ms_per_step", "cpus", "local_rank"):
            assert key in r, key
        assert r["own_ms_per_step"] >= 2.0 * (1 + r["rank"]) * 0.95          # rank r's step sleeps 2 (1 + r) ms
    slowest = max(r["own_ms_per_step"] for r in ranks)
    assert abs(out["ms_per_step"] - slowest) / slowest < 0.02                  # the job's time IS the slowest rank's own time
    px = 8 * 32 * 3840 * 2160 * 4
    assert abs(out["rehearsal_units_per_s"] - px / (out["ms_per_step"] * 4e-3)) / out["rehearsal_units_per_s"] < 0.01


def test_rank_identity_and_clock_from_a_fake_sysfs(tmp_path):
    from videoprocessingframework_amd import sharding

    d = tmp_path / "0000:c1:00.0"
    d.mkdir()
    (d / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 1270Mhz\n2: 2100Mhz *\n")
    assert sharding.current_sclk_mhz("0000:c1:00.0", str(tmp_path)) == 2100
    assert sharding.current_sclk_mhz("0000:c2:00.0", str(tmp_path)) is None and sharding.current_sclk_mhz(None) is None
    ident = sharding.rank_identity(None)
    assert ident["device"] is None and ident["pci"] is None and ident["cpus"] >= 1 and ident["rank"] == 0
    assert sharding.gather_objects({"a": 1}) == [{"a": 1}]                    # no process group: the rank's own object


def test_bench_refuses_mismatched_world_size():
    import subprocess

    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rehearse-host"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_numa_helpers_read_sysfs_and_bind(tmp_path, monkeypatch):
    """sharding.gpu_numa_cpus / bind_to_gpu_numa against a fake sysfs tree: cpulist parsing, node -1 = no locality, the affinity actually set
    is the intersection with what the process already may use, and a missing tree is a no-op, not an error."""
    import os
    import types

    from videoprocessingframework_amd import sharding as sh

    assert sh.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and sh.parse_cpulist("") == []
    assert sh.pci_address(0, 0xc1, 0) == "0000:c1:00.0"
    dev = tmp_path / "0000:c1:00.0"
    dev.mkdir()
    (dev / "numa_node").write_text("1\n")
    allowed = sorted(os.sched_getaffinity(0))
    local = allowed[: max(1, len(allowed) // 2)]
    (dev / "local_cpulist").write_text(",".join(map(str, local + [4093])) + "\n")   # a CPU this process does not have is ignored
    assert sh.gpu_numa_cpus("0000:c1:00.0", str(tmp_path)) == (1, sorted(local + [4093]))
    assert sh.gpu_numa_cpus("0000:ff:00.0", str(tmp_path)) == (None, [])
    (tmp_path / "0000:c2:00.0").mkdir()
    (tmp_path / "0000:c2:00.0" / "numa_node").write_text("-1\n")
    assert sh.gpu_numa_cpus("0000:c2:00.0", str(tmp_path))[0] is None
    props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0xc1, pci_device_id=0)
    monkeypatch.setattr(sh.torch.cuda, "get_device_properties", lambda i: props)
    # a thread that exists BEFORE the call (what the HIP runtime / OpenMP / gloo threads of a real rank are): it must be moved too —
    # sched_setaffinity(0, ...) alone changes the calling thread only
    import threading
    seen, go, done = {}, threading.Event(), threading.Event()

    def sibling():
        seen["tid"] = threading.get_native_id()
        go.wait(10)
        seen["cpus"] = sorted(os.sched_getaffinity(0))  # (0 = the calling thread)
        done.set()

    th = threading.Thread(target=sibling)
    th.start()
    try:
        info = sh.bind_to_gpu_numa(0, str(tmp_path))
        go.set(); done.wait(10); th.join()
        if len(allowed) > 1:
            assert info["bound"] and info["numa_node"] == 1 and sorted(os.sched_getaffinity(0)) == local
            assert info["threads"] >= 2 and seen["cpus"] == local
        again = sh.bind_to_gpu_numa(0, str(tmp_path))
        assert not again["bound"]                                  # already there
    finally:
        go.set()
        sh._set_affinity_all_threads(allowed)
    props.pci_bus_id = 0xee
    assert sh.bind_to_gpu_numa(0, str(tmp_path))["bound"] is False and sorted(os.sched_getaffinity(0)) == allowed
