"""videoprocessingframework_amd — MI355X-native surface conversion behind VPF's API.

  capi            ctypes view of the C ABI (include/vpf_hip.h, libvpfhip.so)
  PyNvCodec       drop-in Python API (pybind11 over the C++ Task layer in csrc/tc)
  PytorchNvCodec  pitched device memory <-> torch tensors (zero-copy views)
  sharding        one-process-per-GPU clip sharding helpers
"""
from ._hip_runtime import preload as _preload

_preload()
