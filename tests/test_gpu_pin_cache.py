"""-m gpu: the page-locked caller-buffer cache (Tasks.hpp HostPinCache) is OPT-IN, and its cases (tests/cases_gpu_pin_cache.py) run in
a process of their own: with the cache on, whole-suite runs on this image's ROCm aborted inside later pageable hipMemcpy calls of torch / the test
harness (2 of 9 single-process runs; 0 of 16 with it off or on the round-5 tree) — a process that registers caller memory is not
shared with 650 other tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_cache_is_off_by_default():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import PyNvCodec as nvc\n"
            "up = nvc.PyFrameUploader(1920, 1080, nvc.PixelFormat.NV12, 0); f = np.zeros(1920 * 1080 * 3 // 2, np.uint8)\n"
            "[up.UploadSingleFrame(f) for _ in range(4)]\n"
            "s = nvc.PinCacheStats(); assert int(s['registered']) == 0 and int(s['in_place']) == 0, dict(s); print('OFF-OK')\n") % os.path.join(ROOT, "videoprocessingframework_amd")
    env = {k: v for k, v in os.environ.items() if k != "VPF_HIP_PIN_CACHE_MB"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OFF-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_pin_cache_cases_in_a_process_of_their_own():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    env = dict(os.environ, VPF_HIP_PIN_CACHE_MB="1024")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "cases_gpu_pin_cache.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
