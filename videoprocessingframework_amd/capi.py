"""ctypes binding of the C ABI in include/vpf_hip.h (libvpfhip.so).

This is the thinnest possible Python view of the drop-in boundary: it passes raw device pointers,
pitches and a hipStream_t, exactly as the reference's Task layer hands them to NPP
(src/TC/src/TasksColorCvt.cpp:122-182).  There is no CPU fallback: if libvpfhip.so is missing or a
launch fails, these functions raise.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libvpfhip.so")

# Pixel_Format (reference: src/TC/inc/MemoryInterfaces.hpp:30-49)
UNDEFINED, Y, RGB, NV12, YUV420, RGB_PLANAR, BGR, YCBCR, YUV444, RGB_32F, RGB_32F_PLANAR, YUV422, P10, P12 = range(14)
YUV444_10bit, YUV420_10bit, NV12_PLANAR, GRAY12 = 14, 15, 16, 17
BT_601, BT_709, CS_UNSPEC = 0, 1, 2
MPEG, JPEG, CR_UDEF = 0, 1, 2
INTERP_NEAREST, INTERP_LINEAR, INTERP_LANCZOS3 = 0, 1, 2
OK, ERR_UNSUPPORTED, ERR_BAD_ARG, ERR_LAUNCH, ERR_NO_DEVICE = range(5)
TUNE_NV12_RGB_VARIANT = 1
TUNE_RESIZE_TILE = 2
TUNE_RESIZE_BAND = 3
TUNE_RESIZE_MFMA = 5

EXPORTS = [
    "vpf_convert", "vpf_convert_batch", "vpf_convert_supported", "vpf_resize", "vpf_remap", "vpf_convert_resize",
    "vpf_convert_resize_batch", "vpf_resize_batch", "vpf_remap_batch", "vpf_resize_ws", "vpf_resize_batch_ws", "vpf_resize_workspace_bytes",
    "vpf_status_string", "vpf_version", "vpf_device_count", "vpf_set_tuning", "vpf_trace_push", "vpf_trace_pop",
]


class Workspace(C.Structure):
    """vpf_workspace: caller-owned scratch region for per-shape filter tables (include/vpf_hip.h); `opaque` starts zeroed"""
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_uint64), ("opaque", C.c_uint64 * 40)]


class Plane(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("pitch", C.c_uint32), ("reserved", C.c_uint32)]


class Size(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32)]


class Exec(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32), ("stream", C.c_void_p)]


class FrameIO(C.Structure):
    _fields_ = [("src", Plane * 3), ("dst", Plane * 3)]


class VpfError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        super().__init__(f"{what}: {status_string(status)} (vpf_status {status})")


_lib = None


def lib() -> C.CDLL:
    """Load libvpfhip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m videoprocessingframework_amd._build` "
                "(there is no CPU fallback for the conversion path)")
        from ._hip_runtime import preload

        preload()  # one HIP runtime per process, whatever the import order relative to torch
        L = C.CDLL(LIB_PATH)
        PP, PE, PF = C.POINTER(Plane), C.POINTER(Exec), C.POINTER(FrameIO)
        L.vpf_convert.argtypes = [PE, C.c_int, C.c_int, C.c_int, C.c_int, Size, PP, PP]
        L.vpf_convert_batch.argtypes = [PE, C.c_int, C.c_int, C.c_int, C.c_int, Size, C.c_uint32, PF]
        L.vpf_convert_supported.argtypes = [C.c_int] * 4
        L.vpf_resize.argtypes = [PE, C.c_int, C.c_int, Size, PP, Size, PP]
        L.vpf_remap.argtypes = [PE, C.c_int, Size, PP, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, Size, PP]
        L.vpf_convert_resize.argtypes = [PE, C.c_int, C.c_int, C.c_int, C.c_int, Size, PP, Size, PP]
        L.vpf_convert_resize_batch.argtypes = [PE, C.c_int, C.c_int, C.c_int, C.c_int, Size, Size, C.c_uint32, PF]
        L.vpf_resize_batch.argtypes = [PE, C.c_int, C.c_int, Size, Size, C.c_uint32, PF]
        L.vpf_remap_batch.argtypes = [PE, C.c_int, Size, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, Size, C.c_uint32, PF]
        L.vpf_resize_ws.argtypes = [PE, C.c_int, C.c_int, Size, PP, Size, PP, C.POINTER(Workspace)]
        L.vpf_resize_batch_ws.argtypes = [PE, C.c_int, C.c_int, Size, Size, C.c_uint32, PF, C.POINTER(Workspace)]
        L.vpf_resize_workspace_bytes.argtypes = [C.c_int, C.c_int, Size, Size]
        L.vpf_resize_workspace_bytes.restype = C.c_uint64
        L.vpf_status_string.argtypes = [C.c_int]
        L.vpf_status_string.restype = C.c_char_p
        L.vpf_version.restype = C.c_char_p
        L.vpf_set_tuning.argtypes = [C.c_int, C.c_int]
        _lib = L
    return _lib


def status_string(s: int) -> str:
    return lib().vpf_status_string(s).decode()


def version() -> str:
    return lib().vpf_version().decode()


def device_count() -> int:
    return lib().vpf_device_count()


def set_tuning(key: int, value: int) -> int:
    return lib().vpf_set_tuning(key, value)


def convert_supported(src_fmt, dst_fmt, cs, cr) -> bool:
    return bool(lib().vpf_convert_supported(src_fmt, dst_fmt, cs, cr))


EXEC_DST_REUSED = 1


def make_exec(stream: int = 0, device: int = -1, flags: int = 0) -> Exec:
    return Exec(device, flags, stream or None)


def planes(desc) -> "C.Array[Plane]":
    """desc: iterable of (device_ptr, pitch_bytes), at most 3; an array built earlier is passed through (callers in a
    per-frame loop build their descriptors once)."""
    if isinstance(desc, Plane * 3):
        return desc
    p = (Plane * 3)()
    for i, (ptr, pitch) in enumerate(desc):
        p[i].ptr, p[i].pitch = ptr, pitch
    return p


def _check(st: int, what: str):
    if st != OK:
        raise VpfError(st, what)


def convert(ex: Exec, src_fmt, dst_fmt, cs, cr, w, h, src, dst, check=True) -> int:
    st = lib().vpf_convert(C.byref(ex), src_fmt, dst_fmt, cs, cr, Size(w, h), planes(src), planes(dst))
    if check:
        _check(st, "vpf_convert")
    return st


def make_batch(frames) -> "C.Array[FrameIO]":
    """frames: list of (src_desc, dst_desc) with desc as in planes()."""
    arr = (FrameIO * len(frames))()
    for i, (s, d) in enumerate(frames):
        s, d = planes(s), planes(d)  # accepts (ptr, pitch) lists and prebuilt Plane arrays alike
        for k in range(3):
            arr[i].src[k].ptr, arr[i].src[k].pitch = s[k].ptr, s[k].pitch
            arr[i].dst[k].ptr, arr[i].dst[k].pitch = d[k].ptr, d[k].pitch
    return arr


def convert_batch(ex: Exec, src_fmt, dst_fmt, cs, cr, w, h, batch, n=None, check=True) -> int:
    st = lib().vpf_convert_batch(C.byref(ex), src_fmt, dst_fmt, cs, cr, Size(w, h), len(batch) if n is None else n, batch)
    if check:
        _check(st, "vpf_convert_batch")
    return st


def resize(ex: Exec, fmt, interp, sw, sh, src, dw, dh, dst, check=True) -> int:
    st = lib().vpf_resize(C.byref(ex), fmt, interp, Size(sw, sh), planes(src), Size(dw, dh), planes(dst))
    if check:
        _check(st, "vpf_resize")
    return st


def resize_batch(ex: Exec, fmt, interp, sw, sh, dw, dh, batch, n=None, check=True) -> int:
    """batch: FrameIO array from make_batch(); every plane of every frame in as few dispatches as possible"""
    st = lib().vpf_resize_batch(C.byref(ex), fmt, interp, Size(sw, sh), Size(dw, dh), len(batch) if n is None else n, batch)
    if check:
        _check(st, "vpf_resize_batch")
    return st


def resize_workspace_bytes(fmt, interp, sw, sh, dw, dh) -> int:
    return int(lib().vpf_resize_workspace_bytes(fmt, interp, Size(sw, sh), Size(dw, dh)))


def make_workspace(ptr: int, nbytes: int) -> Workspace:
    """a zeroed vpf_workspace over `nbytes` of device memory at `ptr` (256-B aligned), which the caller keeps alive"""
    ws = Workspace()
    ws.ptr, ws.bytes = ptr, nbytes
    return ws


def resize_ws(ex: Exec, fmt, interp, sw, sh, src, dw, dh, dst, ws, check=True) -> int:
    st = lib().vpf_resize_ws(C.byref(ex), fmt, interp, Size(sw, sh), planes(src), Size(dw, dh), planes(dst), C.byref(ws) if ws is not None else None)
    if check:
        _check(st, "vpf_resize_ws")
    return st


def resize_batch_ws(ex: Exec, fmt, interp, sw, sh, dw, dh, batch, ws, n=None, check=True) -> int:
    st = lib().vpf_resize_batch_ws(C.byref(ex), fmt, interp, Size(sw, sh), Size(dw, dh), len(batch) if n is None else n, batch,
                                   C.byref(ws) if ws is not None else None)
    if check:
        _check(st, "vpf_resize_batch_ws")
    return st


def remap_batch(ex: Exec, fmt, sw, sh, xmap_ptr, xmap_pitch, ymap_ptr, ymap_pitch, dw, dh, batch, n=None, check=True) -> int:
    st = lib().vpf_remap_batch(C.byref(ex), fmt, Size(sw, sh), xmap_ptr, xmap_pitch, ymap_ptr, ymap_pitch, Size(dw, dh), len(batch) if n is None else n, batch)
    if check:
        _check(st, "vpf_remap_batch")
    return st


def remap(ex: Exec, fmt, sw, sh, src, xmap_ptr, xmap_pitch, ymap_ptr, ymap_pitch, dw, dh, dst, check=True) -> int:
    st = lib().vpf_remap(C.byref(ex), fmt, Size(sw, sh), planes([src]), xmap_ptr, xmap_pitch, ymap_ptr, ymap_pitch,
                         Size(dw, dh), planes([dst]))
    if check:
        _check(st, "vpf_remap")
    return st


def convert_resize(ex: Exec, src_fmt, dst_fmt, cs, cr, sw, sh, src, dw, dh, dst, check=True) -> int:
    st = lib().vpf_convert_resize(C.byref(ex), src_fmt, dst_fmt, cs, cr, Size(sw, sh), planes(src), Size(dw, dh),
                                  planes(dst))
    if check:
        _check(st, "vpf_convert_resize")
    return st


def convert_resize_batch(ex: Exec, src_fmt, dst_fmt, cs, cr, sw, sh, dw, dh, batch, n=None, check=True) -> int:
    st = lib().vpf_convert_resize_batch(C.byref(ex), src_fmt, dst_fmt, cs, cr, Size(sw, sh), Size(dw, dh),
                                        len(batch) if n is None else n, batch)
    if check:
        _check(st, "vpf_convert_resize_batch")
    return st
