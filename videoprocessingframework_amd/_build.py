"""In-tree build of the native parts (gfx950 only).

  libvpfhip.so      — the C-ABI kernel library (include/vpf_hip.h), hipcc --offload-arch=gfx950
  _PyNvCodec*.so    — pybind11 module over the C++ Task layer (added by build_bindings())

Everything is built next to the sources so the .so files travel with the repo snapshot to the GPU
box (they are git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libvpfhip.so")

KERNEL_TUS = ["vpf_abi.hip", "k_yuv2rgb.hip", "k_relayout.hip", "k_rgb2yuv.hip", "k_resize.hip", "k_lanczos_mfma.hip", "k_remap.hip", "k_convert_resize.hip"]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-fvisibility=hidden",
             "-Wall", "-Wno-unused-function", f"-I{INC}", f"-I{CSRC}"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith(".h")]
    for d, _, fs in os.walk(CSRC):
        hs += [os.path.join(d, f) for f in fs if f.endswith((".h", ".hpp"))]
    return hs


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} {cmd[-1]}")
    return r


def build_kernels(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for tu in KERNEL_TUS:
        src = os.path.join(CSRC, tu)
        obj = os.path.join(OBJ, tu.replace(".hip", ".o"))
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC, *HIP_FLAGS, "-c", src, "-o", obj])
    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(_run, jobs))
    objs = [os.path.join(OBJ, tu.replace(".hip", ".o")) for tu in KERNEL_TUS]
    if force or jobs or _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


def build_all(force: bool = False):
    out = [build_kernels(force)]
    try:
        from . import _build_bindings  # optional until the Task layer lands
    except ImportError:
        return out
    out += _build_bindings.build(force)
    return out


if __name__ == "__main__":
    print("\n".join(build_all("--force" in sys.argv)))
