"""tools/first_8gpu.sh — the one command for the first multi-GPU node — rehearsed on CPU: gloo ranks, bench.py's host no-op
step, test collection instead of test runs.  It cannot fail on syntax the day the hardware appears; no scaling number comes out of this."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_first_8gpu_rehearsal(tmp_path):
    out = str(tmp_path / "f8")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "first_8gpu.sh"), "--rehearse", "--out", out, "--gpus", "1 2 8"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rep = json.load(open(os.path.join(out, "report.json")))
    assert rep["rehearsal"] is True and rep["failed"] == []
    assert [s["step"] for s in rep["steps"]] == ["pytest multidevice + multirank", "bench.py --gpus 1 (gloo)", "bench.py --gpus 2 (gloo)", "bench.py --gpus 8 (gloo)",
                                                 "shard_pipeline.py --gpus 8 --source pinned", "shard_pipeline.py --gpus 8 --source pageable"]
    assert sorted(rep["bench"]) == ["1", "2", "8"]
    for n, b in rep["bench"].items():
        assert len(b["ranks"]) == int(n) and len(b["per_rank_ms_per_step"]) == int(n)   # the per-rank diagnostics travel into the one report
        assert b["value"] is None                                                      # a rehearsal measures nothing
