"""tools/lab/libvpfhip_lab.so — the measurement lab (round 1's experimental NV12 -> RGB kernel forms and the bandwidth probes), kept
OUT of libvpfhip.so.  `bench.py --sweep` quotes their rates, so the non-probe forms must still be real conversions: each is checked
bit for bit against the oracle here.  The product never loads this library (tests/test_abi_cpu.py::test_product_does_not_touch_lab)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from gpu_util import DevPlanes, assert_planes_equal, stream_handle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "lab", "libvpfhip_lab.so")
CONVERSIONS = [1, 2, 3, 5, 6, 7, 10, 11, 13, 14, 16, 17, 18, 19, 20, 21, 27, 28, 29, 31, 32, 33, 34, 35, 36, 38, 41, 42, 43, 45, 46]
PROBES = [15, 22, 23, 24, 25, 26]


@pytest.fixture(scope="module")
def lab(capi):
    capi.lib()
    assert os.path.exists(LAB), "build it with `python tools/lab/build_lab.py`"
    L = C.CDLL(LAB)
    L.vpf_lab_nv12_rgb.argtypes = [C.POINTER(capi.Exec), C.c_int, C.c_int, C.c_int, C.c_int, capi.Size, C.c_uint32, C.POINTER(capi.FrameIO)]
    return L


@pytest.mark.parametrize("variant", CONVERSIONS)
def test_lab_conversion_forms_write_the_product_pixels(lab, capi, oracle, variant):
    assert lab.vpf_lab_is_conversion(variant) == 1
    n_ran = 0
    for dst in ("RGB", "BGR", "RGB_PLANAR"):
        for (w, h) in [(1920, 32), (3840, 8), (848, 464)]:
            src = oracle.synth(oracle.NV12, w, h, 1001)
            s, d = DevPlanes(src), DevPlanes(oracle.alloc(getattr(oracle, dst), w, h))
            io = capi.make_batch([(s.desc(), d.desc())])
            ex = capi.make_exec(stream_handle())
            rc = lab.vpf_lab_nv12_rgb(C.byref(ex), variant, getattr(capi, dst), 1, 0, capi.Size(w, h), 1, io)
            assert rc in (0, 1), f"variant {variant} {dst} {w}x{h}: rc {rc}"
            if rc == 1:      # this form does not apply to this shape / output class (the lab never falls back)
                continue
            torch.cuda.synchronize()
            got, intact = d.download()
            assert intact
            _, want = oracle.convert(oracle.NV12, getattr(oracle, dst), 1, 0, w, h, src, oracle.FP32)
            assert_planes_equal(got, want, f"lab variant {variant} {dst} {w}x{h}")
            n_ran += 1
    assert n_ran >= 2


def test_lab_knows_which_variants_are_probes(lab):
    assert all(lab.vpf_lab_is_conversion(v) == 0 for v in PROBES)
    assert all(lab.vpf_lab_is_conversion(v) == -1 for v in (0, 4, 8, 9, 12, 30, 37, 40, 44, 99))  # (the lab's own 45 / 46 are the prototype of the product's p16x)   # product kernels are not in the lab
