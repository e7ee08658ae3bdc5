"""The N>1 path executed on hardware before an 8-GPU node exists: two ranks of bench.py / tools/shard_pipeline.py share the
box's one MI355X through the gloo backend (RCCL refuses two ranks on one device; with gloo the ranks still meet at the same
barriers and run the same reduction).  What this proves: rendezvous on 127.0.0.1, LOCAL_RANK % device_count, per-rank ring
construction, barrier placement and the sum-units / max-time aggregation all run; what it cannot show: scaling."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc, script, *args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script, *args]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"one JSON line from rank 0 only, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_two_ranks_share_one_gpu():
    steps, ring = 5, 8
    out = _torchrun(2, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", str(steps), "--warmup", "2", "--no-cpu", "--ring", str(ring))
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak" and out["data"] == "synthetic"
    assert len(out["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in out["per_rank_ms_per_step"])
    px = 2 * ring * 3840 * 2160 * steps                        # both ranks' pixels ...
    assert abs(out["value"] - px / (out["ms_per_step"] * steps * 1e-3) / 1e9) / out["value"] < 0.01   # ... over the MAX time
    assert out["ms_per_step"] >= 0.95 * max(out["per_rank_ms_per_step"])
    assert 100 < out["value"] < 1800                            # two ranks on ONE GPU cannot beat one GPU's roofline
    assert out["roofline"]["frac"] <= 1.0


def test_shard_pipeline_two_ranks_share_one_gpu():
    """config 4's runner (tools/shard_pipeline.py): clip s -> rank s mod N, host frames -> PyFrameUploader -> PySurfaceConverter
    per rank; every rank checks its last frame of every clip against a checksum of the host data it fed."""
    out = _torchrun(2, os.path.join("tools", "shard_pipeline.py"), "--gpus", "2", "--backend", "gloo", "--clips", "4", "--frames", "6",
                    "--width", "1920", "--height", "1080")
    assert out["n_gpus"] == 2 and out["clips"] == 4 and out["clips_per_rank"] == [[0, 2], [1, 3]]
    assert out["frames_total"] == 24 and out["verified_clips"] == 4
    # (both rates of a 6-frame run are launch-overhead numbers; that the device-resident one is the larger holds on an idle box only — the suite
    # runs this next to three other workers — so only their presence is asserted; tools/shard_pipeline.py's own runs measure them)
    assert out["end_to_end"]["value"] > 0 and out["device_resident"]["value"] > 0
