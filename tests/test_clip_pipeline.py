"""tools/clip_pipeline.py — BASELINE config 1's runner (clip -> libav demux + software decode -> NV12 -> upload -> NV12->RGB) — exercised
against the stub libav (tests/libav_stub): the bindings variant tests/_build/pynvcodec_stubav carries a PyFfmpegDecoder linked against a
stand-in libav that "decodes" synthetic clips, so the runner's control flow, its decode leg and (with -m gpu) its whole chain run here,
and every frame's CRC is compared with the ORACLE (decoded NV12 = the oracle's YUV420 -> NV12 of the stub's planes; RGB = the oracle's
NV12 -> RGB of that).  Real bitstreams need real libav: where it exists the same command runs on tests/test.mp4 of the reference."""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from test_feeder_stub_libav import yuv420_planes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "clip_pipeline.py")


def _stub_pkg():
    from videoprocessingframework_amd import _build_bindings as bb

    base = os.path.join(ROOT, "tests", "_build", "pynvcodec_stubav")
    if os.path.isdir("/root/reference") or not os.path.isdir(base):   # build container: (re)build; GPU box: the prebuilt variant travels
        base = bb.build_stub_libav_variant()
    return base


def _run(args, tmp_path):
    crc = str(tmp_path / "crc.json")
    r = subprocess.run([sys.executable, TOOL] + args + ["--dump-crc", crc], capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stdout[-2000:] + r.stderr[-4000:]
    return r.returncode, json.loads(line[-1]), (json.load(open(crc)) if os.path.exists(crc) else None)


def _nv12(oracle, w, h, i, seed):
    _, p = oracle.convert(oracle.YUV420, oracle.NV12, oracle.BT_601, oracle.MPEG, w, h, yuv420_planes(w, h, i, seed))
    return p


def test_runner_decode_leg_against_the_stub_decoder(oracle, tmp_path):
    w, h, n, seed = 128, 72, 9, 4
    rc, out, crc = _run(["--clip", f"synth:w={w},h={h},n={n},fmt=0,seed={seed},delay=2,audio=2", "--decode-only", "--pynvcodec", _stub_pkg()], tmp_path)
    assert rc == 0 and out["frames"] == n and out["size"] == f"{w}x{h}" and out["cores"] >= 1 and out["decode_only"]["frames_per_s"] > 0
    want = [zlib.crc32(np.concatenate([p.reshape(-1) for p in _nv12(oracle, w, h, i, seed)]).tobytes()) for i in range(n)]
    assert crc["nv12"] == want and crc["rgb"] == []
    rc, out, _ = _run(["--clip", f"synth:w={w},h={h},n={n},fmt=0,seed={seed}", "--decode-only", "--frames", "4", "--pynvcodec", _stub_pkg()], tmp_path)
    assert rc == 0 and out["frames"] == 4


def test_runner_says_so_when_the_build_has_no_libav(tmp_path):
    """the product build of this image has no decoder: the runner reports that (exit code 2), it does not pretend"""
    rc, out, _ = _run(["--clip", "whatever.mp4", "--decode-only"], tmp_path)
    assert rc == 2 and "no libav decoder" in out["error"]


@pytest.mark.gpu
def test_runner_whole_chain_equals_the_oracle_frame_by_frame(oracle, tmp_path):
    w, h, n, seed = 256, 144, 7, 11
    rc, out, crc = _run(["--clip", f"synth:w={w},h={h},n={n},fmt=0,seed={seed},delay=1", "--pynvcodec", _stub_pkg()], tmp_path)
    assert rc == 0 and out["frames"] == n and out["verified_frames"] == n, out
    assert out["end_to_end"]["frames_per_s"] > 0 and out["device_resident"]["frames_per_s"] > 0
    cs, cr = out["colour"]
    for i in range(n):
        nv12 = _nv12(oracle, w, h, i, seed)
        assert crc["nv12"][i] == zlib.crc32(np.concatenate([p.reshape(-1) for p in nv12]).tobytes())
        st, rgb = oracle.convert(oracle.NV12, oracle.RGB, cs, cr, w, h, nv12, oracle.FP32)
        assert st == 0 and crc["rgb"][i] == zlib.crc32(rgb[0].tobytes()), f"frame {i}"
