"""One HIP runtime per process.

PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the same SONAME as /opt/rocm's) and
link it under the name "libamdhip64.so".  If our libraries (linked against libamdhip64.so.7) are loaded BEFORE torch,
the loader maps /opt/rocm's copy first and then a second copy for torch: two HIP runtimes in one process, and the one
that initialises second reports "no ROCm-capable device".  Pre-loading torch's copy (when torch is installed) makes both
resolve to the same object whatever the import order.  No torch import happens here.
"""
import ctypes
import importlib.util
import os

_done = False


def preload() -> None:
    global _done
    if _done:
        return
    _done = True
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    lib = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(lib):
        try:
            ctypes.CDLL(lib, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass
