"""Token/Task state machine: ours (videoprocessingframework_amd/csrc/tc/TC_CORE.hpp) against the REFERENCE's own
TC_CORE compiled from /root/reference/src/TC/TC_CORE by oracle/Makefile into oracle/_ref/libtc_core_ref.so — the one
part of the reference that builds without CUDA/NPP/libav.  The same C shim (oracle/ref_shim.cpp) is compiled against
both headers and driven through identical random call sequences."""
import ctypes as C
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libtc_core_ref.so")
OURS = os.path.join(ROOT, "tests", "_build", "libtc_core_ours.so")


def _load(path):
    L = C.CDLL(path)
    L.ref_token_new.restype = C.c_void_p
    L.ref_token_del.argtypes = [C.c_void_p]
    L.ref_task_new.restype = C.c_void_p
    L.ref_task_new.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.c_int]
    L.ref_task_del.argtypes = [C.c_void_p]
    for f in ("ref_task_set_input", "ref_task_set_output"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    for f in ("ref_task_get_input", "ref_task_get_output"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_uint32]
        getattr(L, f).restype = C.c_void_p
    for f in ("ref_task_clear_inputs", "ref_task_clear_outputs", "ref_task_execute", "ref_task_runs"):
        getattr(L, f).argtypes = [C.c_void_p]
    for f in ("ref_task_num_inputs", "ref_task_num_outputs"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_uint64
    for f in ("ref_task_clear_inputs", "ref_task_clear_outputs", "ref_task_del", "ref_token_del"):
        getattr(L, f).restype = None
    L.ref_task_name.argtypes = [C.c_void_p]
    L.ref_task_name.restype = C.c_char_p
    return L


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(REF):
        import oracle
        oracle.build()
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    os.makedirs(os.path.dirname(OURS), exist_ok=True)
    src = os.path.join(ROOT, "oracle", "ref_shim.cpp")
    inc = os.path.join(ROOT, "videoprocessingframework_amd", "csrc", "tc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", f"-I{inc}", src, "-o", OURS])
    return _load(REF), _load(OURS)


class Driver:
    """Runs one library; records every observable as token INDICES (pointers differ between the two libs)."""

    def __init__(self, L, ni, no, with_sync, run_ret):
        self.L = L
        self.counter = C.c_int(0)
        self.toks = [L.ref_token_new() for _ in range(6)]
        self.task = L.ref_task_new(b"TestTask", ni, no, C.byref(self.counter) if with_sync else None, run_ret)

    def idx(self, p):
        return None if not p else self.toks.index(p)

    def op(self, name, a=0, b=0):
        L, t = self.L, self.task
        if name == "set_in":
            return L.ref_task_set_input(t, self.toks[a] if a >= 0 else None, b)
        if name == "set_out":
            return L.ref_task_set_output(t, self.toks[a] if a >= 0 else None, b)
        if name == "get_in":
            return self.idx(L.ref_task_get_input(t, b))
        if name == "get_out":
            return self.idx(L.ref_task_get_output(t, b))
        if name == "clear_in":
            L.ref_task_clear_inputs(t)
            return None
        if name == "clear_out":
            L.ref_task_clear_outputs(t)
            return None
        if name == "exec":
            return (L.ref_task_execute(t), L.ref_task_runs(t), self.counter.value)
        if name == "state":
            ni, no = L.ref_task_num_inputs(t), L.ref_task_num_outputs(t)
            return (ni, no, L.ref_task_name(t), [self.idx(L.ref_task_get_input(t, i)) for i in range(ni + 2)],
                    [self.idx(L.ref_task_get_output(t, i)) for i in range(no + 2)])
        raise ValueError(name)

    def close(self):
        self.L.ref_task_del(self.task)
        for t in self.toks:
            self.L.ref_token_del(t)


@pytest.mark.parametrize("seed", range(12))
def test_random_sequences_match_reference(libs, seed):
    ref, ours = libs
    rnd = random.Random(seed)
    ni, no = rnd.randint(0, 4), rnd.randint(0, 4)
    with_sync, run_ret = rnd.random() < 0.6, int(rnd.random() < 0.3)
    a, b = Driver(ref, ni, no, with_sync, run_ret), Driver(ours, ni, no, with_sync, run_ret)
    assert a.op("state") == b.op("state")
    for _ in range(300):
        name = rnd.choice(["set_in", "set_out", "get_in", "get_out", "clear_in", "clear_out", "exec", "state"])
        x, y = rnd.randint(-1, 5), rnd.randint(0, 6)  # token index (-1 = nullptr), slot (may be out of range)
        assert a.op(name, x, y) == b.op(name, x, y), (name, x, y)
    a.close()
    b.close()


def test_execute_semantics(libs):
    """Execute = Run then the sync callback, only when both callback and args are set (Task.cpp:50-57)."""
    for L in libs:
        d = Driver(L, 2, 1, True, 0)
        assert d.op("exec") == (0, 1, 1) and d.op("exec") == (0, 2, 2)
        d.close()
        d = Driver(L, 2, 1, False, 1)
        assert d.op("exec") == (1, 1, 0)  # failing Run, no sync registered
        assert d.op("set_in", 0, 2) == 0 and d.op("set_in", 0, 1) == 1 and d.op("get_in", 0, 1) == 0
        d.close()
