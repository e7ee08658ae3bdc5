"""Host-side planning of the matrix-core Lanczos launch (csrc/vpf_lzm_plan.h, compiled here with g++: no HIP, no GPU):
  * the launch-shape planner against the measured sweeps in profiles/ — the cost model must keep picking shapes close to the best measured;
  * the weight-table cache against its contract (what makes "a table in use can never change" and "a stream never reads a table whose
    build it is not ordered behind" true)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lzp():
    from conftest import native_test_build
    extra, out = native_test_build()
    so = os.path.join(out, "liblzm_plan_capi.so")
    src = os.path.join(ROOT, "tests", "c", "lzm_plan_capi.cpp")
    hdr = os.path.join(ROOT, "videoprocessingframework_amd", "csrc")
    deps = [src, os.path.join(hdr, "vpf_lzm_plan.h"), os.path.join(hdr, "vpf_plan_bounds.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Werror", *extra, "-I", hdr, src, "-o", so, "-lm", "-pthread"], check=True)
    L = C.CDLL(so)
    L.lzp_plan.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.lzp_cache_new.restype = C.c_void_p
    L.lzp_cache_new.argtypes = [C.c_uint64]
    L.lzp_cache_free.argtypes = [C.c_void_p]
    L.lzp_cache_launch.restype = C.c_uint32
    L.lzp_cache_launch.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int] + [C.c_uint32] * 5 + [C.c_uint64]
    L.lzp_cache_launch2.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32)]
    L.lzp_cache_used.restype = C.c_uint64
    L.lzp_cache_used.argtypes = [C.c_void_p, C.c_int]
    L.lzp_cache_entries.restype = C.c_uint32
    L.lzp_cache_entries.argtypes = [C.c_void_p, C.c_int]
    L.lzp_sync_complete.argtypes = [C.c_void_p, C.c_uint64]
    L.lzp_sync_next.restype = C.c_uint64
    L.lzp_sync_next.argtypes = [C.c_void_p]
    L.lzp_sync_live.argtypes = [C.c_void_p]
    L.lzp_sync_reset_device.argtypes = [C.c_void_p, C.c_int]
    L.lzp_sync_stream_states.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.lzp_sync_log.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.lzp_ws_get.restype = C.c_uint32
    L.lzp_ws_get.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.c_int, C.c_int] + [C.c_uint32] * 5 + [C.c_uint64, C.POINTER(C.c_uint32)]
    L.lzp_ws_n.restype = C.c_uint32
    L.lzp_ws_n.argtypes = [C.POINTER(C.c_uint64)]
    L.lzp_table_bytes_bound.restype = C.c_uint64
    L.lzp_table_bytes_bound.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    return L


def plan(L, planes, n, forced=0, tables=True, up2=True):
    a = (C.c_uint32 * (5 * len(planes)))(*[v for p in planes for v in p])
    out = (C.c_uint32 * 10)()
    L.lzp_plan(len(planes), a, n, forced, int(tables) | (0 if up2 else 2), out)
    return dict(ok=bool(out[0]), nt=out[1], r=out[2], span=out[3], pitch=out[4], wave_lds=out[5], group_lds=out[6], kc=out[7], rts=out[8], up2=bool(out[9]))


def planes_of(fmt, sw, sh, dw, dh):
    if fmt == "RGB":
        return [(3, sw, sh, dw, dh)]
    if fmt == "Y":
        return [(1, sw, sh, dw, dh)]
    if fmt == "NV12":
        return [(1, sw, sh, dw, dh), (2, sw // 2, sh // 2, dw // 2, dh // 2)]
    return [(1, sw, sh, dw, dh), (1, sw // 2, sh // 2, dw // 2, dh // 2), (1, sw // 2, sh // 2, dw // 2, dh // 2)]


def sweep_lines(name):
    for line in open(os.path.join(ROOT, "profiles", name)):
        m = re.match(r"\[lzm-sweep\] (\w+)\s+(\d+)x(\d+)->(\d+)x(\d+) n=(\d+):", line)
        if m:
            fmt, (sw, sh, dw, dh, nn) = m.group(1), (int(v) for v in m.groups()[1:])
            yield fmt, sw, sh, dw, dh, nn, {(int(a), int(b)): float(c) for a, b, c in re.findall(r"nt(\d) r(\d+)=([\d.]+)", line)}


def sweep_cases():
    """the round-3 sweeps, without their up-scale lines: those kernels were replaced in round 5 (the ring of two; test_planner_on_up_scales)"""
    for n in (32, 8, 1):
        for c in sweep_lines(f"r03_lanczos_shape_sweep_n{n}.txt"):
            if c[1] > c[3]:
                yield c


def pick_time(p, res):
    """the planner's pick read off a sweep at the measured band heights either side of it (outside them: the nearest one)"""
    rs = sorted(r for (nt, r) in res if nt == p["nt"])
    lo = max([r for r in rs if r <= p["r"]], default=rs[0])
    hi = min([r for r in rs if r >= p["r"]], default=rs[-1])
    return res[(p["nt"], lo)] if lo == hi else float(np.interp(p["r"], [lo, hi], [res[(p["nt"], lo)], res[(p["nt"], hi)]]))


def test_planner_stays_close_to_the_best_measured_shape(lzp):
    """For every down-scale case of the three sweeps (RGB / NV12 / YUV420 x two size pairs x 32 / 8 / 1 frames per dispatch) the planner's pick — read
    off the sweep at the measured band heights either side of it — is within 20 % of the best measured shape, 6 % on average.  A change of the
    cost model that loses more than that shows up here, without a GPU."""
    regrets = []
    for fmt, sw, sh, dw, dh, n, res in sweep_cases():
        p = plan(lzp, planes_of(fmt, sw, sh, dw, dh), n)
        assert p["ok"] and p["nt"] in (4, 8) and 2 <= p["r"] <= 64
        rs = sorted(r for (nt, r) in res if nt == p["nt"])
        lo = max([r for r in rs if r <= p["r"]], default=rs[0])      # (outside the measured band heights: the nearest one)
        hi = min([r for r in rs if r >= p["r"]], default=rs[-1])
        t = res[(p["nt"], lo)] if lo == hi else np.interp(p["r"], [lo, hi], [res[(p["nt"], lo)], res[(p["nt"], hi)]])
        regrets.append(t / min(res.values()) - 1.0)
        assert regrets[-1] <= 0.20, (fmt, sw, dw, n, p, t, min(res.values()))
    assert len(regrets) == 18 and float(np.mean(regrets)) <= 0.06, np.mean(regrets)


def test_planner_against_the_round_5_down_scale_sweeps(lzp):
    """The same question asked of sweeps taken with the round's FINAL kernels (profiles/r05_lanczos_shape_sweep_down_n*.txt: RGB / NV12 /
    YUV420 / Y x 1080p -> 720p, 4K -> 1080p, 1080p -> 900p, 4K -> 1440p at 32 / 8 / 1 frames per dispatch — two size pairs and one format the
    round-3 fit never saw): mean regret <= 5 %, worst <= 30 % (Y 4K -> 1440p, where the 0.45 of a 1-channel 4-tile strip is too cheap).  A
    refit of the five constants to these 48 cases gains 0.3 points and loses 3 on the up-scales (tools/lab/fit_lzm_down.py): they stand."""
    regrets = []
    for n in (32, 8, 1):
        for fmt, sw, sh, dw, dh, nn, res in sweep_lines(f"r05_lanczos_shape_sweep_down_n{n}.txt"):
            p = plan(lzp, planes_of(fmt, sw, sh, dw, dh), nn)
            assert p["ok"] and not p["up2"]
            regrets.append(pick_time(p, res) / min(res.values()) - 1.0)
            assert regrets[-1] <= 0.30, (fmt, sw, dw, nn, p, min(res.values()))
    assert len(regrets) == 48 and float(np.mean(regrets)) <= 0.05, (len(regrets), np.mean(regrets))


def test_planner_on_up_scales(lzp):
    """The ring-of-two kernels (round 5) against their own sweeps (profiles/r05_lanczos_shape_sweep_up_n*.txt: RGB / NV12 / YUV420 / Y x five
    up-scales at 32 / 8 / 1 frames per dispatch, four at 64 / 128): 2 x up-scales take 8-tile strips (three workgroups per CU), 1.5 x ones
    4-tile strips — except large launches of them (1440p -> 4K x 32), which take the WIDE 8-tile strips (LzMfma8uw, rows of up to 256 B) —
    within 15 % of the best measured shape everywhere, 4 % on average; and the planner says which kernel it planned for."""
    regrets = []
    for n in (32, 8, 1, 64, 128):
        for fmt, sw, sh, dw, dh, nn, res in sweep_lines(f"r05_lanczos_shape_sweep_up_n{n}.txt"):
            if fmt == "RGB" and nn > 32 and 3 * (sw * sh + dw * dh) > 7_000_000:
                continue   # (frames of that size are dispatched 32 at a time by the ABI: not a launch of nn frames)
            p = plan(lzp, planes_of(fmt, sw, sh, dw, dh), nn)
            assert p["ok"] and p["up2"], (fmt, sw, dw, nn, p)
            regrets.append(pick_time(p, res) / min(res.values()) - 1.0)
            assert regrets[-1] <= 0.15, (fmt, sw, dw, nn, p, min(res.values()))
    assert len(regrets) == 86 and float(np.mean(regrets)) <= 0.04, (len(regrets), np.mean(regrets))
    # which candidates are the ring of two: 2 x -> both strip widths; 1.5 x packed RGB -> the 4-tile strips only; down-scales, and the knob -> none
    assert plan(lzp, planes_of("RGB", 1920, 1080, 3840, 2160), 32)["nt"] == 8
    assert plan(lzp, planes_of("RGB", 1280, 720, 1920, 1080), 32)["nt"] == 4
    assert plan(lzp, planes_of("RGB", 1280, 720, 1920, 1080), 32, forced=8 << 8)["up2"] is False   # (wide 8-tile strips: large launches only)
    big = plan(lzp, planes_of("RGB", 2560, 1440, 3840, 2160), 32)
    assert big["nt"] == 8 and big["up2"] and 128 < big["span"] <= 256
    assert plan(lzp, planes_of("RGB", 2560, 1440, 3840, 2160), 8)["nt"] == 4
    assert plan(lzp, planes_of("RGB", 1280, 720, 1920, 1080), 32, forced=4 << 8)["up2"] is True
    assert plan(lzp, planes_of("RGB", 1920, 1080, 1280, 720), 32)["up2"] is False
    assert plan(lzp, planes_of("RGB", 1920, 1080, 3840, 2160), 32, up2=False)["up2"] is False
    assert plan(lzp, [(3, 1920, 1080, 3840, 1000)], 32)["up2"] is False   # (an up-scale in x only)


def test_planner_on_one_up_scaled_frame_per_dispatch(lzp):
    """Round 6: ONE frame per dispatch (an unmodified PySurfaceResizer.Execute()).  A lone launch is one round of waves and its time is that of the
    fullest CU, so the best band height sits just under a multiple of 256 workgroups; for the ring-of-two kernels (up-scales) on launches whose
    planes have equal strip counts (RGB, Y, NV12) the planner keeps its strip width and re-chooses the band height with an occupancy term fitted to
    the n = 1 sweep (tools/lab/fit_lzm_lone.py): 20 cases, mean regret <= 2.5 %, worst <= 9 % (before: 3.2 % / 13 %); the picks it changes were
    measured band height by band height (profiles/r06_m_lone_upscales.txt).  YUV420 (the first form of the rule lost 9-14 % there), down-scales,
    forced band heights and launches of more than one frame are untouched."""
    regrets = []
    for fmt, sw, sh, dw, dh, nn, res in sweep_lines("r05_lanczos_shape_sweep_up_n1.txt"):
        assert nn == 1
        p = plan(lzp, planes_of(fmt, sw, sh, dw, dh), 1)
        assert p["ok"] and p["up2"] and 2 <= p["r"] <= 64
        regrets.append(pick_time(p, res) / min(res.values()) - 1.0)
        assert regrets[-1] <= 0.09, (fmt, sw, dw, p, min(res.values()))
    assert len(regrets) == 20 and float(np.mean(regrets)) <= 0.025, (len(regrets), np.mean(regrets))
    # the case that motivated it: bands that fill whole multiples of the 256 CUs
    assert plan(lzp, planes_of("RGB", 1920, 1080, 3840, 2160), 1)["r"] == 8      # 45 x 17 = 765 workgroups = 3 per CU (7 tiles: 900; 6: a second round)
    # untouched: a forced band height; launches of more than one frame: the sweep tests above and below (the ring of four at n = 1: the next test)
    assert plan(lzp, planes_of("RGB", 1920, 1080, 3840, 2160), 1, forced=(4 << 8) | 5)["r"] == 5
    assert plan(lzp, planes_of("NV12", 2560, 1440, 3840, 2160), 1)["r"] == 4 and plan(lzp, planes_of("YUV420", 1920, 1080, 3840, 2160), 1)["r"] == 4   # (YUV420: the model's own pick)


def lone_sweep_lines(name):
    """tools/lab/ab/lone_lanczos.py files: one frame per dispatch, knob (nt << 8 | band tiles) -> us; knob 0 (policy) left out"""
    cur, out = None, {}
    for line in open(os.path.join(ROOT, "profiles", name)):
        m = re.match(r"\[lone\] \S+\s+(\w+)\s+(\d+)x(\d+)->(\d+)x(\d+)", line)
        if m:
            cur = (m.group(1),) + tuple(int(v) for v in m.groups()[1:])
            out[cur] = {}
            continue
        m = re.match(r"\[lone\]\s+(0x[0-9a-f]+): ([\d.]+) us", line)
        if m and int(m.group(1), 16):
            out[cur][(int(m.group(1), 16) >> 8, int(m.group(1), 16) & 255)] = float(m.group(2))
    return out


def test_planner_on_one_down_scaled_frame_per_dispatch(lzp):
    """Round 6, the ring of four at n = 1: where the launch fills the chip (more than 384 workgroups as first picked) strip width and band height are
    re-chosen by a model with a ring-fill term, fitted to two boxes' sweeps and cross-validated between them (tools/lab/fit_lzm_lone.py).  Against
    the round-5 sweep (16 cases): mean regret <= 1.5 %, worst <= 7 % (before: 2.8 % / 18 %); against the final tree's own sweep of the 4K cases
    (profiles/r06_p_lone_downscales.txt, every band height measured, 7 cases): mean <= 3.5 %, worst <= 13 % (before: 7.7 % / 21 %)."""
    regrets = []
    for fmt, sw, sh, dw, dh, nn, res in sweep_lines("r05_lanczos_shape_sweep_down_n1.txt"):
        p = plan(lzp, planes_of(fmt, sw, sh, dw, dh), 1)
        assert p["ok"] and not p["up2"] and p["kc"] == 1
        regrets.append(pick_time(p, res) / min(res.values()) - 1.0)
        assert regrets[-1] <= 0.07, (fmt, sw, dw, p, min(res.values()))
    assert len(regrets) == 16 and float(np.mean(regrets)) <= 0.015, np.mean(regrets)
    regrets = []
    for (fmt, sw, sh, dw, dh), res in lone_sweep_lines("r06_p_lone_downscales.txt").items():
        if fmt == "Y" and dw == 1920:
            continue   # (one such plane per dispatch goes to the tile kernel by policy: launch_resize, not this planner)
        p = plan(lzp, planes_of(fmt, sw, sh, dw, dh), 1)
        regrets.append(res[(p["nt"], p["r"])] / min(res.values()) - 1.0)   # (the pick itself was measured)
        assert regrets[-1] <= 0.13, (fmt, sw, dw, p)
    assert len(regrets) == 7 and float(np.mean(regrets)) <= 0.035, np.mean(regrets)
    rgb = plan(lzp, planes_of("RGB", 3840, 2160, 1920, 1080), 1)
    assert (rgb["nt"], rgb["r"]) == (4, 4) and rgb["group_lds"] <= 80 * 1024 and rgb["span"] <= 256   # 23 x 17 = 391 workgroups: 14.2 us against 15.2
    assert plan(lzp, planes_of("YUV420", 1920, 1080, 1600, 900), 1)["r"] == 2      # 323 workgroups: below the gate, the pick stands (6.9 us; bands of 3: 7.6)
    # untouched: any forced shape, launches without tables, more than one frame
    assert plan(lzp, planes_of("RGB", 3840, 2160, 1920, 1080), 1, forced=8 << 8)["nt"] == 8
    assert (plan(lzp, planes_of("RGB", 3840, 2160, 1920, 1080), 1, tables=False)["nt"], plan(lzp, planes_of("RGB", 3840, 2160, 1920, 1080), 2)["nt"]) == (8, 8)


@pytest.mark.parametrize("n,mean_max,worst_max", [(64, 0.06, 0.15), (128, 0.05, 0.15)])
def test_planner_beyond_32_frames_per_dispatch(lzp, n, mean_max, worst_max):
    """Round 5 (up to 128 frames per dispatch): the cost model's large-launch branch (continuous rounds + a tail of one and a half wave lives)
    against the sweeps at 64 and 128 frames (RGB / NV12 / YUV420 / Y x three size pairs each): the pick within 15 % of the best measured
    shape, 5-6 % on average — the round-3 model, asked about such launches, put NV12 and Y 1080p -> 720p on whole-column 4-tile strips
    (46 % / 27 % off the best)."""
    regrets = []
    for fmt, sw, sh, dw, dh, nn, res in sweep_lines(f"r05_lanczos_shape_sweep_n{n}.txt"):
        if sw < dw:
            continue   # (the up-scale lines were measured with the kernels the ring of two replaced: test_planner_on_up_scales)
        p = plan(lzp, planes_of(fmt, sw, sh, dw, dh), nn)
        assert p["ok"] and nn == n
        rs = sorted(r for (nt, r) in res if nt == p["nt"])
        lo = max([r for r in rs if r <= p["r"]], default=rs[0])
        hi = min([r for r in rs if r >= p["r"]], default=rs[-1])
        t = res[(p["nt"], lo)] if lo == hi else np.interp(p["r"], [lo, hi], [res[(p["nt"], lo)], res[(p["nt"], hi)]])
        regrets.append(t / min(res.values()) - 1.0)
        assert regrets[-1] <= worst_max, (fmt, sw, dw, p, t, min(res.values()))
    assert len(regrets) == 8 and float(np.mean(regrets)) <= mean_max, np.mean(regrets)


def test_planner_limits_and_forced_shapes(lzp):
    """What the launcher relies on: the LDS of a workgroup never exceeds 80 KB, the staged row fits the variant's loads, a forced shape is
    honoured when it fits, and shapes outside the kernel's windows are refused (the gather kernel takes them)."""
    rng = np.random.default_rng(5)
    n_ok = n_no = 0
    for _ in range(400):
        ch = int(rng.choice([1, 2, 3]))
        sw, sh = int(rng.integers(16, 4000)), int(rng.integers(8, 2200))
        dw, dh = max(2, int(sw / rng.uniform(0.3, 14.0))), max(2, int(sh / rng.uniform(0.3, 9.0)))   # (refused: beyond ~10 across / ~6 down)
        p = plan(lzp, [(ch, sw, sh, dw, dh)], int(rng.choice([1, 4, 32])))
        if not p["ok"]:
            n_no += 1
            continue
        n_ok += 1
        assert (p["kc"], p["nt"]) in ((1, 8), (1, 4), (2, 4), (3, 2))
        assert p["group_lds"] <= 80 * 1024 and p["span"] <= (8 if p["kc"] >= 2 else 5 if p["nt"] == 8 else 4) * 64 and p["pitch"] >= p["span"] and p["pitch"] % 64 == 32
        if p["kc"] >= 2:  # multi-chunk windows are the fallbacks: no cheaper shape holds this plane's taps
            assert p["pitch"] in (288, 416, 544)
        assert p["group_lds"] == 4 * p["wave_lds"] + 16384 and p["wave_lds"] >= 16 * p["pitch"] + 16 * (16 * p["nt"] + 16)
        assert p["rts"] in (3, 4) and 1 <= p["r"] <= (dh + (1 << p["rts"]) - 1) >> p["rts"]
    assert n_ok > 150 and n_no > 30
    base = [(3, 1920, 1080, 1280, 720)]
    assert plan(lzp, base, 32, (8 << 8) | 5)["r"] == 5 and plan(lzp, base, 32, (8 << 8) | 5)["nt"] == 8
    assert plan(lzp, base, 32, (4 << 8) | 64)["r"] == 45 and plan(lzp, base, 32, (4 << 8))["nt"] == 4   # more tiles than the picture has: one band
    assert plan(lzp, base, 32, 7)["r"] == 7
    net = plan(lzp, [(3, 1920, 1080, 416, 416)], 32)                            # 4.6 x 2.6: the taps of 16 destination bytes do not fit a 64-B window,
    assert net["ok"] and net["kc"] == 2 and net["nt"] == 4 and net["span"] <= 384   # they fit a 128-B one: pass 1 with two K chunks
    assert plan(lzp, base, 32)["kc"] == 1
    res = plan(lzp, [(1, 1920, 1080, 224, 224), (1, 960, 540, 112, 112), (1, 960, 540, 112, 112)], 32)   # the reference's sample: YUV420 1080p -> 224 x 224,
    assert res["ok"] and res["kc"] == 3 and res["nt"] == 2 and res["rts"] == 3 and res["span"] <= 384       # 8.6 x 4.8: 192-B windows, 2-tile strips, half tiles
    assert not plan(lzp, [(3, 1920, 1080, 224, 224)], 32)["ok"] and plan(lzp, [(3, 1920, 1080, 224, 224)], 32, (4 << 8) | 2)["kc"] == 3   # packed RGB: by force only (the tile kernel is faster)
    assert plan(lzp, [(1, 3840, 2160, 416, 416), (2, 1920, 1080, 208, 208)], 32)["kc"] == 3
    assert not plan(lzp, [(3, 1920, 1080, 120, 120)], 32)["ok"]                 # 16 x 9: out of every window
    thumb = plan(lzp, [(3, 1920, 1080, 480, 270)], 32)                          # 4 x 4: a 16-row tile needs more than four source tiles, 8 rows do not
    assert thumb["ok"] and thumb["rts"] == 3 and thumb["kc"] == 2 and plan(lzp, base, 32)["rts"] == 4
    assert plan(lzp, [(1, 1280, 720, 224, 224), (1, 640, 360, 112, 112), (1, 640, 360, 112, 112)], 32)["rts"] == 3   # YUV420 720p -> 224 x 224 (5.7 x 3.2)
    assert not plan(lzp, base + [(1, 1920, 1080, 120, 120)], 32)["ok"]          # one plane out -> the launch is out
    mixed = plan(lzp, [(1, 1920, 1080, 640, 480), (2, 960, 540, 320, 240)], 8)  # NV12 3 x 2.25: both planes on two-chunk windows
    assert mixed["ok"] and mixed["kc"] == 2
    with_t, without = plan(lzp, base, 1, 0, True), plan(lzp, base, 1, 0, False)
    assert without["r"] >= with_t["r"]                                          # a bigger fixed cost per wave never asks for shorter bands


B = 0x80000000


def _log(L, c):
    buf = C.create_string_buffer(1 << 16)
    L.lzp_sync_log(c, buf, len(buf))
    return buf.value.decode().split()


def test_table_cache_contract(lzp):
    """the fallback arena (vpf_lzm_plan.h: LzmTableCache) against a recording stand-in for HIP events: what makes "a launch never reads a table
    whose build it is not ordered behind" and "a table is never overwritten under a kernel that may still read it" true"""
    c = lzp.lzp_cache_new(1 << 20)
    try:
        go = lambda st, dev=0, cap=0, kind=0, k=(3, 1920, 1280, 8), nbytes=100_000: lzp.lzp_cache_launch(c, st, dev, cap, kind, *k, nbytes)
        a = go(0x1000)
        assert a & B and (a & ~B) == 16                      # first use: allocated after the 256 unused bytes, build on this stream
        assert _log(lzp, c) == ["R1@4096d0"]                 # an event behind the build; the launch itself costs no HIP call (the entry remembers its stream)
        assert go(0x1000) == 16                              # same stream again: ordered behind its own build by the stream itself
        assert _log(lzp, c) == []
        assert go(0x2000) == 16                              # another stream: no second build — it WAITS for the build's event
        assert _log(lzp, c) == ["W1@8192"]
        lzp.lzp_sync_complete(c, 1)                          # the build is seen complete once ...
        assert go(0x3000) == 16 and _log(lzp, c) == []       # ... and nobody waits for it again
        assert go(0x2000) == 16 and go(0x1000) == 16
        _log(lzp, c)
        # a capturing stream always queues its own build (a captured build has not run), records nothing (an event recorded there would
        # become a node of the graph), and pins the entry: a graph captured with it replays whenever it likes
        assert go(0x1000, cap=1) == (16 | B) and go(0x1000, cap=1) == (16 | B) and _log(lzp, c) == []
        fresh = go(0x3000, cap=1, k=(1, 640, 320, 4), nbytes=5000)
        assert fresh & B and go(0x3000, k=(1, 640, 320, 4), nbytes=5000) & B   # first met under capture: its build never ran — the plain call builds
        b = go(0x1000, kind=1, k=(1080, 720, 368, 0), nbytes=24_576)
        assert b & B and (b & ~B) == 16 + (100_000 + 255) // 256 * 16 + (5000 + 255) // 256 * 16   # first fit in 256-B steps
        assert go(0x1000, dev=1) == (16 | B)                 # arenas are per device
        used = lzp.lzp_cache_used(c, 0)
        assert go(0x1000, k=(3, 9999, 9999, 8), nbytes=(1 << 20) + 1) == 0      # larger than the arena: no table, no build (the kernel evaluates its weights)
        assert lzp.lzp_cache_used(c, 0) == used               # ... and nothing was taken or evicted for it
        assert go(0x1000, dev=64) == 0 and go(0x1000, dev=-1) == 0
    finally:
        lzp.lzp_cache_free(c)
    z = lzp.lzp_cache_new(0)                                   # VPF_HIP_LANCZOS_TABLE_KB=0
    assert lzp.lzp_cache_launch(z, 1, 0, 0, 0, 3, 1920, 1280, 8, 1000) == 0
    lzp.lzp_cache_free(z)


def test_table_cache_evicts_least_recently_used_behind_its_last_use(lzp):
    """a full arena hands the least recently used entry's space to the new shape — after making the BUILDING stream wait for the event the
    last launch left on that entry (unless that launch is known to have finished) — and never an entry of the launch in progress"""
    c = lzp.lzp_cache_new(256 + 3 * 4096)                      # room for three 4-KiB tables
    try:
        go = lambda st, k, nbytes=4096: lzp.lzp_cache_launch(c, st, 0, 0, 0, k, 1, 1, 8, nbytes)
        offs = [go(0x10, k) & ~B for k in (1, 2, 3)]
        assert offs == [16, 16 + 256, 16 + 512] and lzp.lzp_cache_entries(c, 0) == 3
        go(0x10, 1)                                            # shape 1 is used again: shape 2 is now the oldest
        _log(lzp, c)                                           # (events so far: the three builds = 1, 2, 3)
        d = go(0x20, 4)                                        # a fourth shape, from another stream
        assert d == (offs[1] | B)                              # it took shape 2's place
        lg = _log(lzp, c)
        # stream 0x20 waits for an event recorded just now on the stream that used shape 2 (= behind its last launch) AND for its build
        assert lg[0] == "R4@16d0" and lg[1] == "W4@32" and "W2@32" in lg
        assert go(0x10, 2) & B                                 # shape 2 is gone: it is built again (evicting the oldest: shape 3)
        assert lzp.lzp_cache_entries(c, 0) == 3
        lzp.lzp_sync_complete(c, lzp.lzp_sync_next(c))         # everything queued so far has finished:
        _log(lzp, c)
        lg = _log(lzp, c) if go(0x30, 5) & B else None
        builds_waited = [x for x in lg if x.startswith("W") and int(x[1:].split("@")[0]) <= 6]
        assert lg is not None and not builds_waited           # an eviction then waits for no build any more (only for its readers' streams)
        # an entry that has met more streams than it remembers gets one event per launch instead
        for st in (0x41, 0x42, 0x43, 0x44):
            go(st, 5)
        _log(lzp, c)
        go(0x45, 5)
        assert [x for x in _log(lzp, c) if x.startswith("R") and "@69d0" in x]
        # two tables of ONE launch never evict each other: with room for three, a launch that needs two new ones keeps both
        out = (C.c_uint32 * 2)()
        ka, kb = (C.c_uint32 * 5)(0, 77, 1, 1, 8), (C.c_uint32 * 5)(1, 77, 1, 1, 0)
        lzp.lzp_cache_launch2(c, 0x10, 0, ka, 4096, kb, 4096, out)
        assert out[0] & B and out[1] & B and (out[0] & ~B) != (out[1] & ~B)
        big = (C.c_uint32 * 5)(0, 78, 1, 1, 8)
        lzp.lzp_cache_launch2(c, 0x10, 0, kb, 4096, big, 2 * 4096 + 1, out)     # kb's table is in use by this launch: it may not make room for the big one ...
        assert out[0] == (out[0] & ~B) and out[0] != 0 and out[1] == 0           # ... it is still there, untouched; the big one gets no table (weights in the kernel)
        # 10 000 distinct shapes through an arena that holds three: every one of them gets a table (the arena never "fills up for good")
        for k in range(1000, 11000):
            assert go(0x10, k) & B
        assert lzp.lzp_cache_entries(c, 0) <= 3
        # a device reset: the host forgets the device's tables (static device memory is re-initialised) — the next launch builds again
        last = go(0x10, 10999)
        assert last == (last & ~B)
        lzp.lzp_sync_reset_device(c, 0)
        assert go(0x10, 10999) & B and lzp.lzp_cache_entries(c, 0) == 1
    finally:
        live_before = lzp.lzp_sync_live(c)
        lzp.lzp_cache_free(c)
    assert live_before <= 2 * 3 + 1                            # events are released as entries go: no leak over 10 000 evictions


def test_table_cache_eviction_leaves_foreign_streams_alone_when_it_must(lzp):
    """the streams an entry remembers are somebody else's handles from earlier launches.  At eviction a reader stream that is under
    capture is not recorded on (the record would become a node of that graph): its entry is not a victim for this launch; a handle the
    runtime no longer knows (destroyed stream) is dropped without a record"""
    c = lzp.lzp_cache_new(256 + 2 * 4096)                      # room for two 4-KiB tables
    try:
        go = lambda st, k: lzp.lzp_cache_launch(c, st, 0, 0, 0, k, 1, 1, 8, 4096)
        a, b = go(0x10, 1) & ~B, go(0x20, 2) & ~B              # shape 1 read on stream 0x10 (the older entry), shape 2 on 0x20
        _log(lzp, c)
        lzp.lzp_sync_stream_states(c, 0x10, 0)                 # stream 0x10 has begun a capture since
        d = go(0x30, 3)
        lg = _log(lzp, c)
        assert d == (b | B)                                    # the older entry is left alone: shape 2's place is taken instead ...
        assert not [x for x in lg if x.startswith("R") and "@16d" in x]   # ... and nothing was recorded on the capturing stream
        assert go(0x10, 1) == a                                # shape 1 is still there
        lzp.lzp_sync_stream_states(c, 0, 0x10)                 # capture over, and stream 0x10 has been destroyed
        go(0x30, 3)                                            # (shape 3 is now the most recent: shape 1 is the victim)
        _log(lzp, c)
        e = go(0x40, 4)
        lg = _log(lzp, c)
        assert e == (a | B) and not [x for x in lg if x.startswith("R") and "@16d" in x]   # no record on a handle that names no stream
        # both entries read by capturing streams: no victim — the launch gets no table (weights in the kernel) instead of a wrong one
        lzp.lzp_sync_stream_states(c, 0x30, 0)
        go(0x30, 4)                                            # both live entries (3, 4) have 0x30 among their readers
        assert go(0x50, 5) == 0 and lzp.lzp_cache_entries(c, 0) == 2
        lzp.lzp_sync_stream_states(c, 0, 0)
        assert go(0x50, 5) & B
    finally:
        lzp.lzp_cache_free(c)


def test_workspace_record(lzp):
    """the caller-owned workspace (vpf_workspace.opaque = LzmWorkspace): stream-ordered, no events.  Same shape on the same stream -> no
    rebuild; another stream / a capturing stream / a new shape -> rebuild; no room -> everything is dropped and the region reused from its
    start, except that the entries of the launch in progress are never dropped (the fallback arena serves that table instead)"""
    ws = (C.c_uint64 * 40)()
    region = 256 + 3 * 4096

    def get(st, k, nbytes=4096, cap=0, dev=0, kind=0, launch=None):
        """one lookup; `launch` = the mask of a launch in progress (a fresh one per call otherwise: every lookup its own launch)"""
        m = launch if launch is not None else C.c_uint32(0)
        return lzp.lzp_ws_get(ws, region, st, dev, cap, kind, k, 1, 1, 8, nbytes, C.byref(m))

    assert get(0x10, 1) == (16 | B) and get(0x10, 1) == 16 and get(0x10, 1, kind=1) == ((16 + 256) | B)
    assert get(0x10, 1, cap=1) == (16 | B)                     # captured: build again
    assert get(0x20, 1) == (16 | B) and lzp.lzp_ws_n(ws) == 1  # another stream: nothing in the region is ordered for it — it starts over
    assert get(0x20, 2) & B and get(0x20, 3) & B and lzp.lzp_ws_n(ws) == 3
    assert get(0x20, 4) == (16 | B) and lzp.lzp_ws_n(ws) == 1  # full: reuse from the start (stream order keeps the old readers in front)
    m = C.c_uint32(0)
    assert get(0x20, 5, launch=m) & B and get(0x20, 6, launch=m) & B and m.value == 0b110
    assert get(0x20, 7, launch=m) == 0                         # full AND the entries in it belong to this very launch: not here
    assert get(0x20, 8, nbytes=region) == 0                    # larger than the region: never
    # a launch that HIT an old entry and then misses on a full record must not be handed the hit table's bytes a second time
    m = C.c_uint32(0)
    assert get(0x20, 4, launch=m) == 16 and m.value == 1       # hit: the first entry, at offset 16
    assert get(0x20, 9, launch=m) == 0 and lzp.lzp_ws_n(ws) == 3  # miss on the full record: the fallback arena, nothing dropped
    assert get(0x20, 4, launch=m) == 16 and get(0x20, 5, launch=m) == 16 + 256  # ... and the tables it was given are still there
    assert get(0x20, 9) == (16 | B) and lzp.lzp_ws_n(ws) == 1  # the NEXT launch may start over
    # the same with the entry count as the limit (eight small tables) instead of the bytes
    ws2, m = (C.c_uint64 * 40)(), C.c_uint32(0)
    g2 = lambda k, launch: lzp.lzp_ws_get(ws2, 1 << 20, 0x30, 0, 0, 0, k, 1, 1, 8, 256, C.byref(launch))
    offs = [g2(k, C.c_uint32(0)) & ~B for k in range(8)]
    assert len(set(offs)) == 8 and lzp.lzp_ws_n(ws2) == 8
    assert g2(3, m) == offs[3] and g2(100, m) == 0 and lzp.lzp_ws_n(ws2) == 8 and g2(3, m) == offs[3]
    assert get(0x20, 1, dev=1) == (16 | B)                     # another device: a fresh record


def test_workspace_bound_covers_every_plan(lzp):
    """vpf_resize_workspace_bytes adds lzm_table_bytes_bound over the planes: it must cover what ANY launch shape the planner can pick needs
    (column tables: strips x nt x 2 KiB; row tables: bands x groups per band x 8 KiB)"""
    rng = np.random.default_rng(3)
    for _ in range(300):
        ch = int(rng.choice([1, 2, 3]))
        dw, dh = int(rng.integers(2, 4000)), int(rng.integers(2, 2200))
        bound = lzp.lzp_table_bytes_bound(ch, dw, dh)
        for nt in (4, 8):
            strips = (dw * ch + 16 * nt - 1) // (16 * nt)
            for tiles in (1, 2, 3, 5, 8, 23, 64):
                tiles = min(tiles, (dh + 15) // 16)  # (the planner never asks for more tiles per band than the picture has)
                rows = 16 * tiles
                need = (strips * nt * 2048 + 255) // 256 * 256 + (((dh + rows - 1) // rows) * ((rows + 63) // 64) * 8192 + 255) // 256 * 256
                assert need <= bound, (ch, dw, dh, nt, tiles)
