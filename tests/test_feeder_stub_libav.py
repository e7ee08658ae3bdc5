"""SURVEY §8(f) N3 — the libav feeder RUNS: `csrc/feeder/FfmpegFeeder.cpp` (demux + software decode -> real NV12; reference:
src/TC/src/FfmpegSwDecoder.cpp:141-168,254-287,332-360) is compiled and linked against tests/libav_stub/stub_libav.c, a stand-in
implementation of the dozen libav entry points it uses that "decodes" synthetic clips with padded linesizes, a reorder delay,
interleaved audio packets, and switchable failure modes.  Executed here on the CPU: stream selection, the send / receive loop, the
YUV420P -> NV12 repack (checked against the oracle's YUV420 -> NV12, TasksColorCvt.cpp:945-975), the NV12 pass-through, end of
stream, colour tags, and every error path.  What this cannot cover is real bitstream decoding (no FFmpeg in the image); where
libav exists `_build_bindings.have_libav()` builds the same file against the real thing."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "videoprocessingframework_amd", "csrc")
STUB = os.path.join(ROOT, "tests", "libav_stub")
OUT = os.path.join(ROOT, "tests", "_build", "feeder_stub")
A, B, Cc, D = (3, 1, 3), (5, 2, 1), (7, 11, 13), (1, 3, 5)   # the stub's content formula (stub_libav.c)


def plane(p, w, h, i, seed):
    y, x = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    return ((A[p] * x + B[p] * y + Cc[p] * i + seed * D[p]) & 0xFF).astype(np.uint8)


def yuv420_planes(w, h, i, seed):
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return [plane(0, w, h, i, seed), plane(1, cw, ch, i, seed), plane(2, cw, ch, i, seed)]


@pytest.fixture(scope="module")
def feeder():
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "libfeeder_stub.so")
    srcs = [os.path.join(CSRC, "feeder", "FfmpegFeeder.cpp"), os.path.join(STUB, "feeder_capi.cpp"), os.path.join(STUB, "stub_libav.c")]
    deps = srcs + [os.path.join(CSRC, "feeder", "FfmpegFeeder.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        obj = os.path.join(OUT, "stub_libav.o")
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-fPIC", "-Wall", "-Wextra", "-Werror", f"-I{STUB}", "-c", srcs[2], "-o", obj])
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-Werror", f"-I{STUB}", f"-I{os.path.join(CSRC, 'tc')}",
                               f"-I{os.path.join(CSRC, 'feeder')}", f"-I{os.path.join(ROOT, 'include')}", srcs[0], srcs[1], obj, "-Wl,-z,defs", "-o", so])
    L = C.CDLL(so)
    L.feeder_open.restype = C.c_void_p
    L.feeder_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.feeder_close.argtypes = [C.c_void_p]
    L.feeder_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.feeder_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_int]
    L.feeder_next.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_char_p, C.c_int]
    L.feeder_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_int]
    return L


class Clip:
    def __init__(self, L, url):
        self.L, self.err = L, C.create_string_buffer(512)
        self.h = L.feeder_open(url.encode(), self.err, 512)

    def info(self):
        out = (C.c_longlong * 7)()
        self.L.feeder_info(self.h, out)
        return list(out)

    def decode(self, nbytes, cap=None):
        buf = np.full(nbytes + 64, 0xAB, np.uint8)   # 64 guard bytes behind the frame
        rc = self.L.feeder_decode(self.h, buf.ctypes.data, nbytes if cap is None else cap, self.err, 512)
        assert (buf[nbytes:] == 0xAB).all(), "the feeder wrote past the destination"
        return rc, buf[:nbytes]

    def close(self):
        if self.h:
            self.L.feeder_close(self.h)
            self.h = None


@pytest.mark.parametrize("w,h,fmt", [(64, 32, 0), (62, 30, 12), (33, 17, 0), (848, 464, 0)])
def test_yuv420p_clips_come_out_as_real_nv12(feeder, oracle, w, h, fmt):
    o = oracle
    n, seed = 7, 5
    c = Clip(feeder, f"synth:w={w},h={h},n={n},fmt={fmt},seed={seed},delay=2,audio=3")
    assert c.h, c.err.value
    cw, ch = (w + 1) // 2, (h + 1) // 2
    nbytes = w * h + 2 * cw * ch
    info = c.info()
    assert info[:2] == [w, h] and info[4] == 3 and info[6] == 29970            # NV12 = Pixel_Format 3; 30000/1001 fps
    assert info[5] == nbytes   # FrameBytes() of the announced size: the buffer DecodeNextFrame asks for, odd sizes included (33 x 17: 867, not 841)
    for i in range(n):
        rc, got = c.decode(nbytes)
        assert rc == 1, (i, c.err.value)
        st, want = o.convert(o.YUV420, o.NV12, o.BT_601, o.MPEG, w, h, yuv420_planes(w, h, i, seed))   # the oracle's C4 re-layout
        assert st == 0
        assert np.array_equal(got, np.concatenate([want[0].reshape(-1), want[1].reshape(-1)])), f"frame {i}"
    assert c.decode(nbytes)[0] == 0 and c.decode(nbytes)[0] == 0               # end of stream, and it stays there
    c.close()


def test_nv12_frames_pass_through(feeder):
    w, h, n, seed = 96, 40, 4, 9
    c = Clip(feeder, f"synth:w={w},h={h},n={n},fmt=23,seed={seed}")
    for i in range(n):
        rc, got = c.decode(w * h * 3 // 2)
        assert rc == 1
        y, u, v = yuv420_planes(w, h, i, seed)
        uv = np.stack([u, v], -1).reshape(h // 2, w)
        assert np.array_equal(got, np.concatenate([y.reshape(-1), uv.reshape(-1)]))
    assert c.decode(w * h * 3 // 2)[0] == 0
    c.close()


def test_colour_tags_follow_the_reference_mapping(feeder):
    """FfmpegSwDecoder.cpp:437-465: BT709 -> BT_709, BT470BG / SMPTE170M -> BT_601, else UNSPEC; MPEG / JPEG / else UDEF"""
    for cs, want in ((1, 1), (5, 0), (6, 0), (2, 2), (0, 2)):
        for cr, wantr in ((1, 0), (2, 1), (0, 2)):
            c = Clip(feeder, f"synth:cs={cs},cr={cr}")
            assert c.info()[2:4] == [want, wantr]
            c.close()


def test_open_errors(feeder):
    for url, msg in (("/no/such/file.mp4", b"can't open"), ("synth:novideo=1", b"no video stream"), ("synth:nocodec=1", b"no software decoder")):
        c = Clip(feeder, url)
        assert not c.h and msg in c.err.value, (url, c.err.value)


def test_decode_errors_surface_as_exceptions_and_never_overrun(feeder):
    w, h = 64, 32
    c = Clip(feeder, f"synth:w={w},h={h},n=6,fail_at=2")                         # the decoder fails on the third picture
    assert c.decode(3072)[0] == 1 and c.decode(3072)[0] == 1
    rc, _ = c.decode(3072)
    assert rc == -1 and b"decode error" in c.err.value
    c.close()
    c = Clip(feeder, f"synth:w={w},h={h},n=3,fmt=4")                              # 4:2:2 pictures: not a feeder for the NV12 path
    rc, _ = c.decode(3072)
    assert rc == -1 and b"not 8-bit 4:2:0" in c.err.value
    c.close()
    c = Clip(feeder, f"synth:w={w},h={h},n=3")
    rc, _ = c.decode(3072, cap=3071)                                             # destination one byte short
    assert rc == -1 and b"destination holds 3071" in c.err.value
    assert c.decode(3072)[0] == 1                                                # the stream goes on with the next picture
    c.close()


def test_mid_stream_resolution_change_is_sized_from_the_frame(feeder, oracle):
    """capacity used to be checked against the container's announced size while the copy loops used the frame's own —
    a larger frame overflowed the caller's buffer.  Now the frame decides: the one-call form refuses, the two-step form resizes."""
    o = oracle
    w, h = 64, 32
    c = Clip(feeder, f"synth:w={w},h={h},n=5,change_at=2,seed=1")
    assert c.decode(3072)[0] == 1 and c.decode(3072)[0] == 1
    rc, _ = c.decode(3072)                                                       # frame 2 is 128 x 64 = 12288 B
    assert rc == -1 and b"128x64" in c.err.value and b"12288" in c.err.value
    dims = (C.c_longlong * 3)()
    assert feeder.feeder_next(c.h, dims, c.err, 512) == 1 and list(dims) == [128, 64, 12288]     # frame 3, two-step form
    buf = np.zeros(12288, np.uint8)
    assert feeder.feeder_copy(c.h, buf.ctypes.data, buf.size, c.err, 512) == 1
    st, want = o.convert(o.YUV420, o.NV12, o.BT_601, o.MPEG, 128, 64, yuv420_planes(128, 64, 3, 1))
    assert np.array_equal(buf, np.concatenate([want[0].reshape(-1), want[1].reshape(-1)]))
    assert c.info()[:2] == [128, 64]                                             # Width() / Height() follow the stream
    c.close()


# ------------------------------------------------------------------------------------------------------------------------
# the Python binding of the feeder (PyFfmpegDecoder, reference: src/PyNvCodec/src/PyFFMpegDecoder.cpp:37-70,220-268): the bindings are
# built a second time with -DVPF_WITH_LIBAV against the stub (tests/_build/pynvcodec_stubav) and driven in a subprocess (a process
# can hold only one module named _PyNvCodec)
# ------------------------------------------------------------------------------------------------------------------------
_DECODER_SCRIPT = r"""
import sys, json, zlib
import numpy as np
sys.path.insert(0, {base!r})
import PyNvCodec as nvc
assert nvc._native.HAVE_LIBAV and nvc.PyFfmpegDecoder is nvc._native.PyFfmpegDecoder
res = {{}}
try:
    nvc.PyFfmpegDecoder("/no/such/file.mp4", {{}}, 0)
    res["bad_url"] = "no exception"
except RuntimeError as e:
    res["bad_url"] = str(e)
d = nvc.PyFfmpegDecoder("synth:w=64,h=32,n=5,seed=3,change_at=3,cs=5,cr=2", {{"threads": "1"}}, 0)
res["info"] = [d.Width(), d.Height(), int(d.ColorSpace()), int(d.ColorRange()), int(d.Format()), round(d.Framerate(), 2)]
frames = []
frame = np.zeros(1, np.uint8)
while d.DecodeSingleFrame(frame):
    frames.append([int(frame.size), zlib.crc32(frame.tobytes())])
res["frames"] = frames
if {gpu}:
    d = nvc.PyFfmpegDecoder("synth:w=64,h=32,n=4,seed=3,change_at=2", {{}}, 0)
    surf = []
    while True:
        s = d.DecodeSingleSurface()
        if s.Empty():
            break
        out = np.zeros(1, np.uint8)
        assert nvc.PySurfaceDownloader(s.Width(), s.Height(), nvc.PixelFormat.NV12, 0).DownloadSingleSurface(s, out)
        surf.append([s.Width(), s.Height(), int(s.Format()), zlib.crc32(out.tobytes())])
    res["surfaces"] = surf
print("RESULT " + json.dumps(res))
"""


def _run_decoder(gpu):
    import json
    import sys
    import zlib  # noqa: F401

    from videoprocessingframework_amd import _build_bindings as bb

    base = os.path.join(ROOT, "tests", "_build", "pynvcodec_stubav")
    if os.path.isdir("/root/reference") or not os.path.isdir(base):   # build container: (re)build; GPU box: the prebuilt variant travels
        base = bb.build_stub_libav_variant()
    r = subprocess.run([sys.executable, "-c", _DECODER_SCRIPT.format(base=base, gpu=gpu)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])


def _nv12_crc(oracle, w, h, i, seed):
    import zlib

    _, want = oracle.convert(oracle.YUV420, oracle.NV12, oracle.BT_601, oracle.MPEG, w, h, yuv420_planes(w, h, i, seed))
    return zlib.crc32(np.concatenate([want[0].reshape(-1), want[1].reshape(-1)]).tobytes())


def test_pyffmpegdecoder_binding_decodes_frames_through_the_stub(oracle):
    res = _run_decoder(gpu=False)
    assert "can't open" in res["bad_url"]
    assert res["info"] == [64, 32, 0, 1, 3, 29.97]                 # BT470BG -> BT_601, JPEG range, NV12
    want = [[64 * 32 * 3 // 2, _nv12_crc(oracle, 64, 32, i, 3)] for i in range(3)] + [[128 * 64 * 3 // 2, _nv12_crc(oracle, 128, 64, i, 3)] for i in (3, 4)]
    assert res["frames"] == want                                     # the array is resized when the stream changes resolution


@pytest.mark.gpu
def test_pyffmpegdecoder_decodes_straight_into_surfaces(oracle):
    """decode (stub) -> NV12 repack -> PyFrameUploader (pinned staging, copy stream) -> download: byte-equal to the oracle's NV12"""
    res = _run_decoder(gpu=True)
    want = [[64, 32, 3, _nv12_crc(oracle, 64, 32, i, 3)] for i in range(2)] + [[128, 64, 3, _nv12_crc(oracle, 128, 64, i, 3)] for i in (2, 3)]
    assert res["surfaces"] == want
