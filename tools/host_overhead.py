"""Host-side cost of one PySurfaceConverter.Execute (pybind -> Task -> C ABI -> launch) on frames small enough that the GPU
is never the limit."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
import PyNvCodec as nvc
from videoprocessingframework_amd import capi
PF = nvc.PixelFormat
for (w, h) in ((320, 180), (1920, 1080)):
    up = nvc.PyFrameUploader(w, h, PF.NV12, 0)
    s = up.UploadSingleFrame(np.zeros(w * h * 3 // 2, np.uint8))
    conv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.RGB, 0)
    cc = nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_709, nvc.ColorRange.MPEG)
    for _ in range(100): conv.Execute(s, cc)
    torch.cuda.synchronize()
    n = 5000
    t0 = time.perf_counter()
    for _ in range(n): conv.Execute(s, cc)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"[host] PySurfaceConverter.Execute {w}x{h}: {(t1 - t0) / n * 1e6:.2f} us/call issue, {(t2 - t0) / n * 1e6:.2f} us/call incl. drain")
    # raw C ABI through ctypes for comparison
    ex = capi.make_exec(nvc.GetStream(0), 0)
    src = [(s.PlanePtr(0).GpuMem(), s.Pitch(0)), (s.PlanePtr(1).GpuMem(), s.Pitch(1))]
    out = conv.Execute(s, cc)
    dst = [(out.PlanePtr(0).GpuMem(), out.Pitch(0))]
    t0 = time.perf_counter()
    for _ in range(n): capi.convert(ex, capi.NV12, capi.RGB, 1, 0, w, h, src, dst)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"[host] ctypes vpf_convert           {w}x{h}: {(t1 - t0) / n * 1e6:.2f} us/call issue")
