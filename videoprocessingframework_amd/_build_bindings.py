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


def build(force: bool = False):
    os.makedirs(B.OBJ, exist_ok=True)
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = B._headers()
    flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-x", "hip", "--offload-arch=gfx950",
             f"-I{B.INC}", f"-I{TC}", f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}"]
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(B.OBJ, "tc_" + os.path.basename(src).replace(".cpp", ".o"))
        objs.append(obj)
        if force or B._newer(obj, [src] + hdrs):
            jobs.append([B.HIPCC, *flags, "-c", src, "-o", obj])
    with cf.ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(B._run, jobs))
    out = module_path()
    if force or jobs or B._newer(out, objs + [B.LIB]):
        B._run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, f"-L{B.PKG}", "-lvpfhip", "-Wl,-rpath,$ORIGIN/.."])
    return [out]
