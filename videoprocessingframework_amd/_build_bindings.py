"""Builds the C++ Task layer (csrc/tc) + pybind11 bindings (csrc/bindings) into PyNvCodec/_PyNvCodec*.so,
linked against the in-tree libvpfhip.so (rpath $ORIGIN/..)."""
from __future__ import annotations

import concurrent.futures as cf
import os
import sysconfig

import pybind11

from . import _build as B

TC = os.path.join(B.CSRC, "tc")
BND = os.path.join(B.CSRC, "bindings")
OUT_DIR = os.path.join(B.PKG, "PyNvCodec")
SOURCES = [os.path.join(TC, "MemoryInterfaces.cpp"), os.path.join(TC, "Tasks.cpp"), os.path.join(BND, "PyNvCodec.cpp")]


def module_path() -> str:
    return os.path.join(OUT_DIR, "_PyNvCodec" + sysconfig.get_config_var("EXT_SUFFIX"))


def have_libav() -> bool:
    """True when a program using libavformat/libavcodec/libavutil compiles AND links here (no pkg-config in the image)."""
    import subprocess
    import tempfile

    src = ('extern "C" {\n#include <libavformat/avformat.h>\n#include <libavcodec/avcodec.h>\n}\n'
           "int main() { return avformat_version() && avcodec_version() ? 0 : 1; }\n")
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "p.cpp")
        open(f, "w").write(src)
        r = subprocess.run(["g++", f, "-o", os.path.join(d, "p"), "-lavformat", "-lavcodec", "-lavutil"], capture_output=True)
        return r.returncode == 0


def build(force: bool = False):
    os.makedirs(B.OBJ, exist_ok=True)
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = B._headers()
    libav = os.environ.get("VPF_WITH_LIBAV", "auto")
    libav = have_libav() if libav == "auto" else libav not in ("0", "no", "off")
    sources = list(SOURCES) + ([os.path.join(B.CSRC, "feeder", "FfmpegFeeder.cpp")] if libav else [])
    flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-x", "hip", "--offload-arch=gfx950",
             f"-I{B.INC}", f"-I{TC}", f"-I{os.path.join(B.CSRC, 'feeder')}", f"-I{pybind11.get_include()}",
             f"-I{sysconfig.get_paths()['include']}"] + (["-DVPF_WITH_LIBAV"] if libav else [])
    jobs, objs = [], []
    for src in sources:
        obj = os.path.join(B.OBJ, "tc_" + os.path.basename(src).replace(".cpp", ".o"))
        objs.append(obj)
        if force or B._newer(obj, [src] + hdrs):
            jobs.append([B.HIPCC, *flags, "-c", src, "-o", obj])
    with cf.ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(B._run, jobs))
    out = module_path()
    if force or jobs or B._newer(out, objs + [B.LIB]):
        B._run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, f"-L{B.PKG}", "-lvpfhip", "-Wl,-rpath,$ORIGIN/.."] +
               (["-lavformat", "-lavcodec", "-lavutil"] if libav else []))
    return [out]
