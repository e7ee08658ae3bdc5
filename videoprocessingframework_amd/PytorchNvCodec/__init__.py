"""PytorchNvCodec — pitched device memory <-> torch.Tensor on torch-ROCm.

Function surface of the reference's src/PytorchNvCodec/src/PytorchNvCodec.cpp:141-258:
  makefromDevicePtrUint8 / DptrToTensor (ptr, width, height, pitch, elem_size[, stream]) -> torch.uint8 [height, width]
  TensorToDptr (tensor, ptr, width, height, pitch, elem_size[, stream])
The reference allocates a tensor and cudaMemcpy2D's into it (:36-87) — a copy.  Here the pitched plane is first
exposed to torch as a ZERO-COPY strided view (through __cuda_array_interface__, strides = (pitch, 1)), and the
reference-named functions are that view plus one strided D2D copy on the requested stream; `view_plane` /
`view_surface_planar` hand out the zero-copy view itself (config 5 of BASELINE.json), removing 2 x 3 B/px of traffic.

torch is plumbing here (allocation, streams); no pixel arithmetic happens in this module.
"""
from __future__ import annotations

import torch


class _DevMem:
    """Minimal __cuda_array_interface__ carrier for a pitched uint8 region that someone else owns."""

    def __init__(self, ptr: int, height: int, width: int, pitch: int, owner=None):
        self.owner = owner  # keeps the Surface alive while torch holds the view
        self.__cuda_array_interface__ = {
            "shape": (height, width), "typestr": "|u1", "data": (int(ptr), False), "strides": (int(pitch), 1), "version": 2,
        }


def _check(ptr, elem_size, fn):
    if elem_size != 1:
        raise RuntimeError(f"{fn}: only torch.uint8 data type is supported")  # PytorchNvCodec.cpp:40-45
    if not ptr:
        raise RuntimeError(f"{fn}: Video frame has void device ptr.")


def view_plane(ptr: int, width: int, height: int, pitch: int, owner=None, device=None) -> torch.Tensor:
    """Zero-copy torch.uint8 view [height, width] with strides (pitch, 1) of pitched device memory."""
    _check(ptr, 1, "view_plane")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return torch.as_tensor(_DevMem(ptr, height, width, pitch, owner), device=dev)


def view_surface_planar(surface, gpu_id: int | None = None) -> torch.Tensor:
    """Zero-copy [3, H, W] uint8 view of an RGB_PLANAR / YUV444 Surface (one W x 3H allocation, plane i at
    base + i*H*pitch — reference layout MemoryInterfaces.cpp:1593-1600)."""
    p = surface.PlanePtr()
    h3, w, pitch = p.Height(), p.Width(), p.Pitch()
    flat = view_plane(p.GpuMem(), w, h3, pitch, owner=surface, device=None if gpu_id is None else f"cuda:{gpu_id}")
    return flat.view(3, h3 // 3, w) if pitch == w else flat.as_strided((3, h3 // 3, w), ((h3 // 3) * pitch, pitch, 1))


def _on_stream(stream: int):
    if not stream:
        return torch.cuda.stream(torch.cuda.current_stream())
    return torch.cuda.stream(torch.cuda.ExternalStream(int(stream)))


def DptrToTensor(ptr: int, width: int, height: int, pitch: int, elem_size: int, stream: int = 0) -> torch.Tensor:
    """New contiguous torch.uint8 tensor [height, width] holding a copy of the pitched plane."""
    _check(ptr, elem_size, "makefromDevicePtrUint8")
    if not stream:
        # The plane was usually written a moment ago by a converter on ITS stream (the per-GPU stream of the gpu_id
        # constructors is non-blocking, like the reference's: PyNvCodec.cpp:107).  The reference's stream-less overload is a
        # blocking cudaMemcpy2D that nothing orders after that stream; here the device is drained first, so the copy can
        # never read a half-written surface.  Pass `stream` (the converter's) to stay asynchronous.
        torch.cuda.synchronize()
    with _on_stream(stream):
        out = view_plane(ptr, width, height, pitch).contiguous().clone() if pitch == width else view_plane(ptr, width, height, pitch).contiguous()
    if not stream:
        torch.cuda.current_stream().synchronize()  # the reference's stream-less overload is cudaMemcpy2D (blocking)
    return out


makefromDevicePtrUint8 = DptrToTensor


def TensorToDptr(tensor: torch.Tensor, ptr: int, width: int, height: int, pitch: int, elem_size: int, stream: int = 0) -> None:
    """Copy a torch.uint8 tensor of width*height elements into pitched device memory."""
    _check(ptr, elem_size, "copytoDevicePtrUint8")
    if tensor.dtype != torch.uint8 or not tensor.is_cuda:
        raise RuntimeError("copytoDevicePtrUint8: need a CUDA/HIP torch.uint8 tensor")
    if tensor.numel() != width * height:
        raise RuntimeError("copytoDevicePtrUint8: tensor has the wrong number of elements")
    if not stream:
        torch.cuda.synchronize()  # earlier readers / writers of the destination surface on other streams (see DptrToTensor)
    with _on_stream(stream):
        view_plane(ptr, width, height, pitch, device=tensor.device).copy_(tensor.reshape(height, width))
    if not stream:
        torch.cuda.current_stream().synchronize()


def surface_from_tensor(tensor: torch.Tensor, fmt=None):
    """Wrap a contiguous CUDA/HIP uint8 tensor as a non-owning PyNvCodec.Surface — the converter then writes straight into
    the tensor (`PySurfaceConverter.ExecuteBatch([src], [surface_from_tensor(t)])`), no copy at all.
      [3, H, W] -> RGB_PLANAR (default) or YUV444;   [H, W, 3] -> RGB (default) or BGR;   [H, W] -> Y
    The tensor is kept alive by the returned Surface.  Stream ordering is the caller's, exactly as with the reference:
    kernels that produced / will consume the tensor on torch's stream are not ordered against a converter running on its
    own stream — build the converter on `torch.cuda.current_stream().cuda_stream` or synchronise in between."""
    try:
        import PyNvCodec as nvc
    except ImportError:  # package-relative import when used as videoprocessingframework_amd.PytorchNvCodec
        from .. import PyNvCodec as nvc
    if tensor.dtype != torch.uint8 or not tensor.is_cuda or not tensor.is_contiguous():
        raise RuntimeError("surface_from_tensor: need a contiguous CUDA/HIP torch.uint8 tensor")
    PF = nvc.PixelFormat
    if tensor.dim() == 3 and tensor.shape[0] == 3:
        f, (h, w), pitch = (fmt or PF.RGB_PLANAR), tensor.shape[1:], tensor.shape[2]
    elif tensor.dim() == 3 and tensor.shape[2] == 3:
        f, (h, w), pitch = (fmt or PF.RGB), tensor.shape[:2], 3 * tensor.shape[1]
    elif tensor.dim() == 2:
        f, (h, w), pitch = (fmt or PF.Y), tensor.shape, tensor.shape[1]
    else:
        raise RuntimeError("surface_from_tensor: expected [3,H,W], [H,W,3] or [H,W]")
    s = nvc.Surface.Wrap(f, int(w), int(h), int(pitch), tensor.data_ptr())
    s._owner = tensor
    return s
