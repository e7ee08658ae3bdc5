"""Host side of the persistent launches (csrc/vpf_persist.h, compiled here with g++: no HIP, no GPU): a stream keeps ONE slot of work counters
(launches of a stream run in order, so they may share counters; launches of two streams may run at the same time, so they may not); a full
table hands on only slots whose stream has drained; the XCDs' shares of an item list are contiguous, complete and even.  Plus a model of
the device side's ticket protocol: whatever the interleaving, every chunk is taken exactly once; the two counter sets of a slot take turns."""
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
    L.pst_take.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_int)]
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


def test_a_slot_hands_out_its_two_counter_sets_in_turn(pst):
    """a launch draws from one set of eight counters and zeroes the other; the slot's next launch draws from that other one — also when the
    slot has changed hands in between (its last launch zeroed the set that is next)"""
    t = pst.pst_new()
    try:
        st = C.c_int(-1)
        seen = []
        for _ in range(5):
            assert pst.pst_take(t, 0, 0x1000, C.byref(st)) == 0
            seen.append(st.value)
        assert seen == [0, 1, 0, 1, 0]
        assert pst.pst_take(t, 0, 0x2000, C.byref(st)) == 1 and st.value == 0            # another stream: another slot, its own turn
        for i in range(62):
            pst.pst_take(t, 0, 0x10000 + i, C.byref(st))
        assert pst.pst_used(t) == 64
        s = pst.pst_take(t, 0, 0x99999, C.byref(st))                                      # full table, everybody idle: the oldest slot (0x1000's) moves on
        assert s == 0 and st.value == 1                                                   # ... and keeps its turn: 0x1000's fifth launch drew from set 0 and zeroed set 1
    finally:
        pst.pst_free(t)


@pytest.mark.parametrize("seed", range(30))
def test_ticket_protocol_model(pst, seed):
    """k_planes_mp_persist's protocol (round 6) replayed with a random scheduler.  G workgroups of 4 waves; workgroup b names counter b & 7; a
    wave's first chunk is static (index (b >> 3) * 4 + w of its counter's share, if the share is that long); tickets hand out the rest of a
    share; the ticket for the chunk after the next is drawn when a chunk starts (one outstanding draw per wave: `tn`); a dry counter sends the
    wave on to hops - 1 neighbours, each looked at before it is drawn from.  -> with hops = 1 and G >= 8, or hops = 8 and any G, every chunk
    is taken exactly once, for any interleaving; a second launch on the other counter set sees zeros after the first one zeroed them."""
    rng = np.random.default_rng(seed)
    total = int(rng.integers(0, 600))
    hops = 1 if seed % 2 else 8
    G = int(rng.integers(8 if hops == 1 else 1, 48))
    lo = (C.c_uint32 * 9)()
    pst.pst_shares(total, lo)
    sets = [[0] * 8, [5] * 8]                                                             # set 1 still holds the counts of some earlier launch
    for launch in range(2):
        ctr, other = sets[launch & 1], sets[(launch & 1) ^ 1]
        for i in range(8):
            other[i] = 0                                                                  # workgroup 0's first eight lanes
        def statics(x):
            nx = lo[x + 1] - lo[x]
            w = ((G - x + 7) >> 3) * 4 if G > x else 0
            return min(w, nx)
        taken = []
        waves = []
        for b in range(G):
            for w in range(4):
                x = b & 7
                st = {"xcc": x, "hop": 0, "x": x, "lo": lo[x] + statics(x), "left": lo[x + 1] - lo[x] - statics(x), "tn": None, "queue": [], "done": False}
                st["tn"] = ("pending", x)                                                 # first(): draw()
                idx = (b >> 3) * 4 + w
                if idx < lo[x + 1] - lo[x]:
                    st["queue"].append(lo[x] + idx)
                waves.append(st)
        def land(st):                                                                     # the outstanding atomicAdd performs now
            if st["tn"] is not None and st["tn"][0] == "pending":
                x = st["tn"][1]
                st["tn"] = ("value", ctr[x]); ctr[x] += 1
        while not all(w["done"] for w in waves):
            st = waves[int(rng.integers(0, len(waves)))]
            if st["done"]:
                continue
            if rng.random() < 0.5:
                land(st); continue                                                        # the memory system makes progress on its own
            if st["queue"]:
                taken.append(st["queue"].pop()); continue                                 # the static chunk is worked through
            # next()
            while True:
                if st["hop"] >= hops:
                    st["done"] = True; break
                land(st)
                t = st["tn"][1]
                if t < st["left"]:
                    taken.append(st["lo"] + t)
                    st["tn"] = ("pending", st["x"])
                    break
                moved = False
                while True:
                    st["hop"] += 1
                    if st["hop"] >= hops:
                        st["done"] = True; break
                    x = (st["xcc"] + st["hop"]) & 7
                    st["x"], st["lo"], st["left"] = x, lo[x] + statics(x), lo[x + 1] - lo[x] - statics(x)
                    if ctr[x] < st["left"]:
                        moved = True; break
                if st["done"]:
                    break
                st["tn"] = ("pending", st["x"])
        assert sorted(taken) == list(range(total)), (total, G, hops, launch)
