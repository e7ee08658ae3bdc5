"""Host side of the persistent launches (csrc/vpf_persist.h, compiled here with g++: no HIP, no GPU): a stream keeps ONE slot of work counters
(launches of a stream run in order, so they may share counters; launches of two streams may run at the same time, so they may not); a full
table hands on only slots whose stream has drained; the XCDs' shares of an item list are contiguous, complete and even.  Plus a model of
the device side's ticket protocol: whatever the interleaving, every item is taken exactly once and every counter is back at zero."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pst():
    from conftest import native_test_build
    extra, out = native_test_build()
    so = os.path.join(out, "libpersist_capi.so")
    src = os.path.join(ROOT, "tests", "c", "persist_capi.cpp")
    hdr = os.path.join(ROOT, "videoprocessingframework_amd", "csrc")
    deps = [src, os.path.join(hdr, "vpf_persist.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Werror", *extra, "-I", hdr, src, "-o", so, "-pthread"], check=True)
    L = C.CDLL(so)
    L.pst_new.restype = C.c_void_p
    L.pst_free.argtypes = [C.c_void_p]
    L.pst_busy.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    L.pst_slot.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    L.pst_used.argtypes = [C.c_void_p]
    L.pst_shares.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
    return L


def test_a_stream_keeps_its_slot_and_a_full_table_hands_on_drained_slots_only(pst):
    t = pst.pst_new()
    try:
        a, b = pst.pst_slot(t, 0, 0x1000), pst.pst_slot(t, 0, 0x2000)
        assert a != b and a >= 0 and b >= 0
        assert pst.pst_slot(t, 0, 0x1000) == a and pst.pst_slot(t, 0, 0x2000) == b      # the same stream: the same counters
        assert pst.pst_slot(t, 1, 0x1000) not in (a, b)                                   # the same handle on another device is another stream
        slots = {pst.pst_slot(t, 0, 0x10000 + i) for i in range(61)}
        assert len(slots) == 61 and pst.pst_used(t) == 64                                  # 64 streams: the table is full
        for s in (0x1000, 0x2000):
            pst.pst_busy(t, s, 1)
        pst.pst_busy(t, 0x1000, 1)
        for i in range(61):
            pst.pst_busy(t, 0x10000 + i, 1)
        assert pst.pst_slot(t, 0, 0x99999) == -1                                          # every stream still has work queued: not persistent (never a shared slot)
        pst.pst_busy(t, 0x2000, 0)                                                        # stream 0x2000 has drained — but it is not the least recently used
        assert pst.pst_slot(t, 0, 0x99999) == -1                                          # (only the oldest slot is asked about: one stream query per launch)
        assert pst.pst_slot(t, 0, 0x1000) == a                                            # 0x1000 is used again: 0x2000 becomes the oldest
        got = pst.pst_slot(t, 0, 0x99999)
        assert got == b and pst.pst_slot(t, 0, 0x99999) == b                              # ... and its slot is handed on
    finally:
        pst.pst_free(t)


def test_shares_are_contiguous_complete_and_even(pst):
    lo = (C.c_uint32 * 9)()
    for total in (0, 1, 7, 8, 9, 63, 64, 1000, 17280, (1 << 22) - 1):
        pst.pst_shares(total, lo)
        sizes = [lo[i + 1] - lo[i] for i in range(8)]
        assert lo[0] == 0 and lo[8] == total and max(sizes) - min(sizes) <= 1 and all(s >= 0 for s in sizes)


@pytest.mark.parametrize("seed", range(20))
def test_ticket_protocol_model(pst, seed):
    """k_planes_mp_persist's protocol replayed with a random scheduler: W waves, each starting at its own XCD's counter, fetch-and-add until
    the ticket is past the share (one failing fetch per counter and wave), the drawer of ticket n_x + W - 1 stores 0.  -> every item exactly
    once, every counter zero at the end, for any interleaving — including the reset racing with nobody (it is the counter's last access)."""
    rng = np.random.default_rng(seed)
    total, W = int(rng.integers(0, 400)), int(rng.integers(1, 40))
    lo = (C.c_uint32 * 9)()
    pst.pst_shares(total, lo)
    ctr = [0] * 8
    taken = []
    waves = [{"x0": int(rng.integers(0, 8)), "hop": 0, "done": False} for _ in range(W)]
    accesses_after_reset = 0
    reset_done = [False] * 8
    while not all(w["done"] for w in waves):
        w = waves[int(rng.integers(0, W))]
        if w["done"]:
            continue
        x = (w["x0"] + w["hop"]) & 7
        if reset_done[x]:
            accesses_after_reset += 1
        t = ctr[x]; ctr[x] += 1
        nx = lo[x + 1] - lo[x]
        if t >= nx:
            if t == nx + W - 1:
                ctr[x] = 0; reset_done[x] = True
            w["hop"] += 1
            w["done"] = w["hop"] == 8
        else:
            taken.append(lo[x] + t)
    assert sorted(taken) == list(range(total)) and ctr == [0] * 8 and all(reset_done) and accesses_after_reset == 0
