"""ctypes front end of the CPU oracle (oracle/vpf_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``videoprocessingframework_amd``) never does.  PARITY UNPINNED — see
oracle/vpf_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# `make sanitize` (tools/sanitize.sh): VPF_TEST_CFLAGS / VPF_TEST_BUILD_TAG build and load an instrumented copy next to the plain one
_TAG = os.environ.get("VPF_TEST_BUILD_TAG", "")
_LIB_PATH = os.path.join(_HERE, f"libvpforacle_{_TAG}.so" if _TAG else "libvpforacle.so")

EXACT, FP32 = 0, 1

# reference Pixel_Format values (src/TC/inc/MemoryInterfaces.hpp:30-49)
Y, RGB, NV12, YUV420, RGB_PLANAR, BGR, YCBCR, YUV444, RGB_32F, RGB_32F_PLANAR = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
P10, P12 = 12, 13
BT_601, BT_709 = 0, 1
MPEG, JPEG = 0, 1
NEAREST, LINEAR, LANCZOS3 = 0, 1, 2


class Plane(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("pitch", C.c_uint32), ("reserved", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile the oracle (and oracle/_ref when /root/reference exists) with oracle/Makefile."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("vpf_oracle.c", "vpf_oracle.h", "Makefile"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:  # pytest-xdist workers must not link the same file at the same time
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
                subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH), "EXTRA_CFLAGS=" + os.environ.get("VPF_TEST_CFLAGS", "")], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/TC/TC_CORE/src"):
        ref = os.path.join(_HERE, "_ref")
        shim_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ref_shim.cpp", "ref_tc_shim.cpp", "Makefile"))
        for target, so in (("ref", "libtc_core_ref.so"), ("ref_tc", "libtc_ref.so")):  # the reference's own sources
            so = os.path.join(ref, so)
            if force or not os.path.exists(so) or os.path.getmtime(so) < shim_m:
                subprocess.check_call(["make", "-C", _HERE, target], stdout=subprocess.DEVNULL)
        # the reference's converter Task layer for the GPU: CUDA driver calls over HIP, nppi*_Ctx forwarding to libvpfhip
        vpflib = os.path.join(os.path.dirname(_HERE), "videoprocessingframework_amd", "libvpfhip.so")
        if os.path.exists(vpflib):
            so = os.path.join(ref, "libtc_ref_hip.so")
            dep_m = max([os.path.getmtime(vpflib), os.path.getmtime(os.path.join(_HERE, "ref_tc_hip_shim.cpp")), os.path.getmtime(os.path.join(_HERE, "Makefile")),
                         os.path.getmtime(os.path.join(_HERE, "ref_shim_hip", "npp_over_vpf.h")), os.path.getmtime(os.path.join(_HERE, "ref_tasks_stubs.py"))])
            if force or not os.path.exists(so) or os.path.getmtime(so) < dep_m:
                subprocess.check_call(["make", "-C", _HERE, "ref_tc_hip"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        try:
            build()  # no-op when up to date
        except Exception:
            if not os.path.exists(_LIB_PATH):
                raise
        L = C.CDLL(_LIB_PATH)
        P3 = C.POINTER(Plane)
        L.vpfo_convert.argtypes = [C.c_int] * 5 + [C.c_uint32, C.c_uint32, P3, P3]
        L.vpfo_convert_supported.argtypes = [C.c_int] * 4
        L.vpfo_resize.argtypes = [C.c_int] * 3 + [C.c_uint32, C.c_uint32, P3, C.c_uint32, C.c_uint32, P3]
        L.vpfo_remap.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, P3, C.c_void_p, C.c_uint32, C.c_void_p,
                                 C.c_uint32, C.c_uint32, C.c_uint32, P3]
        L.vpfo_convert_resize.argtypes = [C.c_int] * 5 + [C.c_uint32, C.c_uint32, P3, C.c_uint32, C.c_uint32, P3]
        L.vpfo_yuv2rgb_px.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_uint8)]
        L.vpfo_rgb2yuv_px.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_uint8)]
        L.vpfo_yuv2rgb_exhaustive.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.vpfo_rgb2yuv_exhaustive.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
        L.vpfo_set_threads.argtypes = [C.c_int]
        L.vpfo_set_assumption.argtypes = [C.c_int, C.c_int]
        L.vpfo_get_assumption.argtypes = [C.c_int]
        L.vpfo_version.restype = C.c_char_p
        _lib = L
    return _lib


A2_CHROMA_UPSAMPLE, A6_CHROMA_DECIMATE, A8_RESIZE_COORDS, A10_LANCZOS_MINIFY = 2, 6, 8, 10


class assume:
    """with oracle.assume(A8_RESIZE_COORDS, 1): ...  — EXACT mode under a non-default convention (vpf_oracle.h)."""

    def __init__(self, key: int, value: int):
        self.key, self.value = key, value

    def __enter__(self):
        self.prev = lib().vpfo_set_assumption(self.key, self.value)
        if self.prev < 0:
            raise ValueError(f"assumption {self.key} has no value {self.value}")
        return self

    def __exit__(self, *exc):
        lib().vpfo_set_assumption(self.key, self.prev)


def set_threads(n: int) -> int:
    return lib().vpfo_set_threads(n)


def release_threads() -> None:
    lib().vpfo_release_threads()


# ------------------------------------------------------------------------------------------------
# plane geometry (tight host layout); mirrors the reference's per-format plane shapes
# (MemoryInterfaces.cpp:811-913 NV12, :915-1062 YUV420, :1361-1519 RGB/BGR, :1521-1637 planar)
# ------------------------------------------------------------------------------------------------
def plane_shapes(fmt: int, w: int, h: int):
    """[(rows, row_bytes, dtype)] per plane for an image of w x h pixels."""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    if fmt == Y:
        return [(h, w, np.uint8)]
    if fmt in (RGB, BGR):
        return [(h, 3 * w, np.uint8)]
    if fmt == NV12:
        return [(h, w, np.uint8), (ch, 2 * cw, np.uint8)]
    if fmt in (YUV420, YCBCR):
        return [(h, w, np.uint8), (ch, cw, np.uint8), (ch, cw, np.uint8)]
    if fmt in (YUV444, RGB_PLANAR):
        return [(h, w, np.uint8)] * 3
    if fmt == RGB_32F:
        return [(h, 3 * w, np.float32)]
    if fmt == RGB_32F_PLANAR:
        return [(h, w, np.float32)] * 3
    if fmt in (P10, P12):
        return [(h, w, np.uint16), (ch, 2 * cw, np.uint16)]
    raise ValueError(f"format {fmt}")


def alloc(fmt: int, w: int, h: int, fill=None):
    """List of C-contiguous numpy planes for `fmt`."""
    out = []
    for rows, rb, dt in plane_shapes(fmt, w, h):
        a = np.zeros((rows, rb), dtype=dt) if fill is None else np.full((rows, rb), fill, dtype=dt)
        out.append(a)
    return out


def _planes(arrs):
    p = (Plane * 3)()
    for i, a in enumerate(arrs):
        assert a.flags["C_CONTIGUOUS"] or a.strides[1] == a.itemsize
        p[i].ptr = a.ctypes.data
        p[i].pitch = a.strides[0]
    return p


def convert(src_fmt, dst_fmt, cs, cr, w, h, src, mode=FP32, dst=None):
    """Run the oracle converter on numpy planes; returns (status, dst planes)."""
    if dst is None:
        dst = alloc(dst_fmt, w, h)
    st = lib().vpfo_convert(mode, src_fmt, dst_fmt, cs, cr, w, h, _planes(src), _planes(dst))
    return st, dst


def supported(src_fmt, dst_fmt, cs, cr) -> bool:
    return bool(lib().vpfo_convert_supported(src_fmt, dst_fmt, cs, cr))


def resize(fmt, interp, sw, sh, src, dw, dh, mode=FP32, dst=None):
    if dst is None:
        dst = alloc(fmt, dw, dh)
    st = lib().vpfo_resize(mode, fmt, interp, sw, sh, _planes(src), dw, dh, _planes(dst))
    return st, dst


def lanczos_taps(S: int, D: int):
    """FP32-mode Lanczos-3 taps of one axis: (i0[D], q[6 D]) — floor of the source coordinate and the six Q14 weights per sample"""
    i0, q = np.zeros(D, np.int32), np.zeros(6 * D, np.int32)
    L = lib()
    L.vpfo_lanczos_taps_q14.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    if L.vpfo_lanczos_taps_q14(S, D, i0.ctypes.data, q.ctypes.data) != 0:
        raise ValueError("vpfo_lanczos_taps_q14")
    return i0, q


def remap(fmt, sw, sh, src, xmap, ymap, mode=FP32, dst=None):
    dh, dw = xmap.shape
    xmap = np.ascontiguousarray(xmap, dtype=np.float32)
    ymap = np.ascontiguousarray(ymap, dtype=np.float32)
    if dst is None:
        dst = alloc(fmt, dw, dh)
    st = lib().vpfo_remap(mode, fmt, sw, sh, _planes(src), xmap.ctypes.data, xmap.strides[0], ymap.ctypes.data,
                          ymap.strides[0], dw, dh, _planes(dst))
    return st, dst


def convert_resize(src_fmt, dst_fmt, cs, cr, sw, sh, src, dw, dh, mode=FP32, dst=None):
    if dst is None:
        dst = alloc(dst_fmt, dw, dh)
    st = lib().vpfo_convert_resize(mode, src_fmt, dst_fmt, cs, cr, sw, sh, _planes(src), dw, dh, _planes(dst))
    return st, dst


def yuv2rgb_px(cs, cr, y, u, v, mode=EXACT):
    out = (C.c_uint8 * 3)()
    st = lib().vpfo_yuv2rgb_px(mode, cs, cr, y, u, v, out)
    if st:
        raise ValueError("unsupported colour space / range")
    return tuple(out)


def rgb2yuv_px(cr, r, g, b, mode=EXACT):
    out = (C.c_uint8 * 3)()
    st = lib().vpfo_rgb2yuv_px(mode, cr, r, g, b, out)
    if st:
        raise ValueError("unsupported colour range")
    return tuple(out)


def yuv2rgb_exhaustive(cs, cr):
    n = C.c_uint64(0)
    m = lib().vpfo_yuv2rgb_exhaustive(cs, cr, C.byref(n))
    return m, n.value


def rgb2yuv_exhaustive(cr):
    n = C.c_uint64(0)
    m = lib().vpfo_rgb2yuv_exhaustive(cr, C.byref(n))
    return m, n.value


# ------------------------------------------------------------------------------------------------
# synthetic inputs — exact recipe of SURVEY.md §8(d)
# ------------------------------------------------------------------------------------------------
def synth(fmt: int, w: int, h: int, seed: int, dist: str = "A"):
    """Distribution A: uniform 0..255; B: legal video (Y 16..235, chroma 16..240); C: ramps."""
    rng = np.random.default_rng(seed)
    planes = alloc(fmt, w, h)
    for i, p in enumerate(planes):
        rows, rb = p.shape
        if p.dtype == np.float32:
            p[...] = rng.random(p.shape, dtype=np.float32)
        elif p.dtype == np.uint16:
            p[...] = rng.integers(0, 65536, p.shape, dtype=np.uint16)
        elif dist == "A":
            p[...] = rng.integers(0, 256, p.shape, dtype=np.uint8)
        elif dist == "B":
            lo, hi = (16, 236) if i == 0 else (16, 241)
            p[...] = rng.integers(lo, hi, p.shape, dtype=np.uint8)
        else:  # C structured
            xx = np.arange(rb, dtype=np.int64)[None, :]
            yy = np.arange(rows, dtype=np.int64)[:, None]
            if i == 0:
                p[...] = ((xx + 3 * yy) % 256).astype(np.uint8)
            else:
                p[...] = ((xx // 2 + yy // 2 + 85 * i) % 256).astype(np.uint8)
    return planes
