"""host time of each stage of the reference's sample chain through the Python API, on small frames (640 x 360 -> 224 x 224: the GPU work per stage is below the launch cost):
us per Execute() call issued (the resizer with SetAsync(True): no wait), then the whole chain"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
import PyNvCodec as nvc
PF = nvc.PixelFormat
w, h, dw, dh = 640, 360, 224, 224
cc = nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_601, nvc.ColorRange.MPEG)
up = nvc.PyFrameUploader(w, h, PF.NV12, 0)
s = up.UploadSingleFrame(np.zeros(w * h * 3 // 2, np.uint8))
to_yuv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.YUV420, 0)
rs = nvc.PySurfaceResizer(dw, dh, PF.YUV420, 0); rs.SetAsync(True)
to_rgb = nvc.PySurfaceConverter(dw, dh, PF.YUV420, PF.RGB, 0)
to_pln = nvc.PySurfaceConverter(dw, dh, PF.RGB, PF.RGB_PLANAR, 0)
a = to_yuv.Execute(s, cc); b = rs.Execute(a); c = to_rgb.Execute(b, cc); d = to_pln.Execute(c, cc)
assert not (a.Empty() or b.Empty() or c.Empty() or d.Empty())
torch.cuda.synchronize()
n = 20000
def t(fn, name):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"[host-chain] {name:34s} {(t1 - t0) / n * 1e6:5.2f} us/call issued", flush=True)
t(lambda: to_yuv.Execute(s, cc), "NV12 -> YUV420 Execute")
t(lambda: rs.Execute(a), "resize YUV420 Execute (async)")
t(lambda: to_rgb.Execute(b, cc), "YUV420 -> RGB Execute")
t(lambda: to_pln.Execute(c, cc), "RGB -> RGB_PLANAR Execute")
t(lambda: to_pln.Execute(to_rgb.Execute(rs.Execute(to_yuv.Execute(s, cc)), cc), cc), "the chain (4 stages, async resize)")
rs.SetAsync(False)
t(lambda: rs.Execute(a), "resize YUV420 Execute (blocking)")
t(lambda: to_pln.Execute(to_rgb.Execute(rs.Execute(to_yuv.Execute(s, cc)), cc), cc), "the chain (blocking resize)")
t(lambda: to_pln.Execute(None, cc), "Execute(None): pybind + empty surface")
t(lambda: c.Clone(0), "Surface.Clone (owning copy on the GPU)")
t(lambda: c.PlanePtr(0), "PlanePtr (a pybind object return)")
t(lambda: nvc.GetStream(0), "GetStream")
t(lambda: None, "empty lambda")
t(lambda: s.Width(), "a pybind getter")
