"""PyNvCodec — drop-in Python API of VPF's surface path on MI355X.

Same role as the reference's src/PyNvCodec/__init__.py:15-17 (re-export of the compiled `_PyNvCodec`): every
class on the conversion path keeps its reference name and signature — PixelFormat, ColorSpace, ColorRange,
ColorspaceConversionContext, Surface, SurfacePlane, CudaBuffer, PySurfaceConverter, PySurfaceResizer,
PySurfaceRemaper, PyFrameUploader, PySurfaceDownloader, PyBufferUploader, PyCudaBufferDownloader, GetNumGpus.

    import sys; sys.path.insert(0, "<repo>/videoprocessingframework_amd")
    import PyNvCodec as nvc            # or: from videoprocessingframework_amd import PyNvCodec as nvc

NVDEC / NVENC classes (PyNvDecoder, PyNvEncoder) are NVIDIA fixed-function hardware and do not exist here;
constructing them raises with an explanation.  FFmpeg demux / software decode is a host-side feeder that needs
libav, which this image does not ship.
"""
import importlib.util as _ilu
import os as _os


def _preload_hip_runtime():
    """See videoprocessingframework_amd/_hip_runtime.py (duplicated: this package is also importable top-level)."""
    import ctypes

    try:
        spec = _ilu.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        lib = _os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if _os.path.exists(lib):
            try:
                ctypes.CDLL(lib, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


_preload_hip_runtime()
try:
    from ._PyNvCodec import *  # noqa: F401,F403
    from ._PyNvCodec import _UseHostAllocator  # noqa: F401
    from . import _PyNvCodec as _native
except ImportError as e:  # no silent fallback: the native module IS the product
    raise ImportError(
        "PyNvCodec: the native module _PyNvCodec is not built; run "
        "`python -m videoprocessingframework_amd._build` (needs hipcc, gfx950)") from e


class _NotPortable:
    _why = ""

    def __init__(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is not available on MI355X: {self._why}")


class PyNvDecoder(_NotPortable):
    _why = "NVDEC is NVIDIA fixed-function hardware (reference: src/TC/src/NvDecoder.cpp); decode on the host and upload with PyFrameUploader"


class PyNvEncoder(_NotPortable):
    _why = "NVENC is NVIDIA fixed-function hardware (reference: src/TC/src/NvEncoder.cpp)"


class PyFFmpegDemuxer(_NotPortable):
    _why = "needs libavformat, which is not present in this image (host-side feeder, out of the conversion path)"


if getattr(_native, "HAVE_LIBAV", False):  # built where libav exists (csrc/feeder, _build_bindings.py probes for it)
    PyFfmpegDecoder = _native.PyFfmpegDecoder
else:

    class PyFfmpegDecoder(_NotPortable):
        _why = "needs libavcodec, which is not present in this image (host-side feeder, out of the conversion path)"
