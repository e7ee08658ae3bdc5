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


def _build_module(out_dir: str, obj_prefix: str, sources, extra_flags, link_tail, rpath: str, force: bool):
    os.makedirs(B.OBJ, exist_ok=True)
    os.makedirs(out_dir, exist_ok=True)
    hdrs = B._headers()
    flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-x", "hip", "--offload-arch=gfx950",
             f"-I{B.INC}", f"-I{TC}", f"-I{os.path.join(B.CSRC, 'feeder')}", f"-I{pybind11.get_include()}",
             f"-I{sysconfig.get_paths()['include']}"] + list(extra_flags)
    jobs, objs = [], []
    for src in sources:
        if src.endswith(".o"):  # a prebuilt object (the C stub of libav in the test variant)
            objs.append(src)
            continue
        obj = os.path.join(B.OBJ, obj_prefix + os.path.basename(src).replace(".cpp", ".o"))
        objs.append(obj)
        if force or B._newer(obj, [src] + hdrs):
            jobs.append([B.HIPCC, *flags, "-c", src, "-o", obj])
    with cf.ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(B._run, jobs))
    out = os.path.join(out_dir, "_PyNvCodec" + sysconfig.get_config_var("EXT_SUFFIX"))
    if force or jobs or B._newer(out, objs + [B.LIB]):
        B._run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, f"-L{B.PKG}", "-lvpfhip", f"-Wl,-rpath,{rpath}"] + list(link_tail))
    return out


def build(force: bool = False):
    libav = os.environ.get("VPF_WITH_LIBAV", "auto")
    libav = have_libav() if libav == "auto" else libav not in ("0", "no", "off")
    sources = list(SOURCES) + ([os.path.join(B.CSRC, "feeder", "FfmpegFeeder.cpp")] if libav else [])
    return [_build_module(OUT_DIR, "tc_", sources, ["-DVPF_WITH_LIBAV"] if libav else [], ["-lavformat", "-lavcodec", "-lavutil"] if libav else [],
                          "$ORIGIN/..", force)]


def build_stub_libav_variant(force: bool = False) -> str:
    """TEST BUILD (tests/test_feeder_stub_libav.py): the same bindings with the optional PyFfmpegDecoder section compiled in
    (-DVPF_WITH_LIBAV) and linked against tests/libav_stub/stub_libav.c instead of FFmpeg, into tests/_build/pynvcodec_stubav/PyNvCodec/.
    Returns the directory to put on sys.path.  Never part of the product build."""
    import shutil
    import subprocess

    stub = os.path.join(B.ROOT, "tests", "libav_stub")
    base = os.path.join(B.ROOT, "tests", "_build", "pynvcodec_stubav")
    pkg = os.path.join(base, "PyNvCodec")
    os.makedirs(pkg, exist_ok=True)
    shutil.copyfile(os.path.join(OUT_DIR, "__init__.py"), os.path.join(pkg, "__init__.py"))
    stub_o = os.path.join(base, "stub_libav.o")
    if force or B._newer(stub_o, [os.path.join(stub, "stub_libav.c")]):
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-fPIC", f"-I{stub}", "-c", os.path.join(stub, "stub_libav.c"), "-o", stub_o])
    _build_module(pkg, "tcstub_", list(SOURCES) + [os.path.join(B.CSRC, "feeder", "FfmpegFeeder.cpp"), stub_o], ["-DVPF_WITH_LIBAV", f"-I{stub}"], [], B.PKG, force)
    return base
