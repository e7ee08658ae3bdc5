#!/usr/bin/env python3
"""Builds tools/lab/libvpfhip_lab.so (gfx950): the measurement lab of the NV12 -> RGB converter — round 1's experimental kernel
forms and bandwidth probes (tools/lab/k_lab.hip).  NOT product: `videoprocessingframework_amd._build` does not build it and
nothing in the package loads it; `bench.py --sweep` and tests/test_gpu_lab.py do."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "videoprocessingframework_amd", "csrc")
OUT = os.path.join(HERE, "libvpfhip_lab.so")
SRC = os.path.join(HERE, "k_lab.hip")


def build(force=False):
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("k_yuv2rgb_tasks.h", "vpf_device.h", "vpf_internal.h", "vpf_coef.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm",
           "-amdgpu-kernarg-preload-count=16", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
           SRC, "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("lab build failed")
    return OUT


if __name__ == "__main__":
    print(build("--force" in sys.argv))
