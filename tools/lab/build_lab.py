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


# ---- the kernel FORMS no policy selects: libvpfhip.so's own sources compiled once more with -DVPF_LAB_FORMS —
#   the persistent launch of the band kernels (k_planes_mp_persist; VPF_TUNE_RESIZE_BAND | 0x10000 [| 0x40000 | 0x80000]),
#   the two-role Lanczos form (LanczosPairTask; VPF_TUNE_RESIZE_MFMA | 0x20000),
#   the fused kernel's per-wave strips of rounds 2-4 (k_convert_strip; VPF_TUNE_NV12_RGB_VARIANT = 47).
# The product library contains none of them and refuses those knob values; tests/conftest.py's `capi_forms` and the sweep tools
# (SWEEP_LIB=tools/lab/libvpfhip_forms.so) load this build.  Same C ABI, same pixels.
FORMS_OUT = os.path.join(HERE, "libvpfhip_forms.so")
FORMS_TUS = ["vpf_abi.hip", "k_resize.hip", "k_lanczos_mfma.hip", "k_convert_resize.hip"]  # the translation units VPF_LAB_FORMS changes


def build_forms(force=False):
    sys.path.insert(0, ROOT)
    from videoprocessingframework_amd import _build
    import concurrent.futures as cf
    _build.build_kernels()  # the other translation units' objects are the product's
    obj_dir = os.path.join(HERE, "build_forms")
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = _build._headers()
    jobs, objs = [], []
    for tu in _build.KERNEL_TUS:
        if tu not in FORMS_TUS:
            objs.append(os.path.join(_build.OBJ, tu.replace(".hip", ".o")))
            continue
        src, obj = os.path.join(CSRC, tu), os.path.join(obj_dir, tu.replace(".hip", ".o"))
        objs.append(obj)
        if force or _build._newer(obj, [src] + hdrs):
            jobs.append([_build.HIPCC, *_build.HIP_FLAGS, "-DVPF_LAB_FORMS", "-c", src, "-o", obj])
    with cf.ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(_build._run, jobs))
    if force or jobs or _build._newer(FORMS_OUT, objs):
        _build._run([_build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", FORMS_OUT, *objs])
    return FORMS_OUT


if __name__ == "__main__":
    print(build("--force" in sys.argv))
    print(build_forms("--force" in sys.argv))
