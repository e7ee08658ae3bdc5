"""The reference's ResNet sample chain (samples/SampleTorchResnet.py:1073-1138) through the drop-in Python API, device-resident:
NV12 1080p -> YUV420 -> PySurfaceResizer(224 x 224, the reference's Lanczos filter = the default, and bilinear) -> RGB -> RGB_PLANAR.
(a) one Execute() per stage and frame, as the sample is written; (b) the additive ExecuteBatch of every stage over 32 frames;
(c) the additive one-pass PySurfaceConvertResizer NV12 -> bilinear -> RGB_PLANAR (no Lanczos form of it exists).
Frames per second of the whole chain; the ring (64 frames) is larger than what a stage leaves in the Infinity Cache."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
import PyNvCodec as nvc

PF, GPU = nvc.PixelFormat, 0
w, h, tw, th, ring, B = 1920, 1080, 224, 224, 64, 32
cc = nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_601, nvc.ColorRange.MPEG)
rng = np.random.default_rng(1)
up = nvc.PyFrameUploader(w, h, PF.NV12, GPU)
src = [up.UploadSingleFrame(rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)).Clone(GPU) for _ in range(ring)]


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for interp, name in ((2, "lanczos3 (the reference's filter, the default)"), (1, "bilinear (SetInterpolation(1))")):
    to_yuv = nvc.PySurfaceConverter(w, h, PF.NV12, PF.YUV420, GPU)
    rs = nvc.PySurfaceResizer(tw, th, PF.YUV420, GPU); rs.SetInterpolation(interp)
    to_rgb = nvc.PySurfaceConverter(tw, th, PF.YUV420, PF.RGB, GPU)
    to_pln = nvc.PySurfaceConverter(tw, th, PF.RGB, PF.RGB_PLANAR, GPU)

    def per_frame():
        for s in src:
            to_pln.Execute(to_rgb.Execute(rs.Execute(to_yuv.Execute(s, cc)), cc), cc)

    yuv = [nvc.Surface.Make(PF.YUV420, w, h, GPU) for _ in range(B)]
    small = [nvc.Surface.Make(PF.YUV420, tw, th, GPU) for _ in range(B)]
    rgb = [nvc.Surface.Make(PF.RGB, tw, th, GPU) for _ in range(B)]
    pln = [nvc.Surface.Make(PF.RGB_PLANAR, tw, th, GPU) for _ in range(ring)]

    def batched():
        for i in range(0, ring, B):
            assert to_yuv.ExecuteBatch(src[i:i + B], yuv, cc) and rs.ExecuteBatch(yuv, small)
            assert to_rgb.ExecuteBatch(small, rgb, cc) and to_pln.ExecuteBatch(rgb, pln[i:i + B], cc)

    t1 = timed(per_frame)
    rs.SetAsync(True)   # additive: the resizer stops waiting for the stream inside every Execute() (the reference's, and our default, does)
    t1a = timed(per_frame)
    rs.SetAsync(False)
    t2 = timed(batched)
    print(f"[chain] 1080p NV12 -> YUV420 -> resize 224x224 {name} -> RGB -> RGB_PLANAR: Execute() per stage and frame {t1 / ring * 1e6:6.2f} us/frame "
          f"({ring / t1:8.0f} frames/s), with resizer.SetAsync(True) {t1a / ring * 1e6:6.2f} us/frame ({ring / t1a:8.0f} frames/s) | "
          f"ExecuteBatch per stage {t2 / ring * 1e6:6.2f} us/frame ({ring / t2:8.0f} frames/s)", flush=True)

nvc.SetExtendedColorspaces(True)  # NV12 -> RGB under BT.601 + MPEG is refused like the reference refuses it unless the caller opts in
fused = nvc.PySurfaceConvertResizer(w, h, PF.NV12, tw, th, PF.RGB_PLANAR, GPU)
out = [nvc.Surface.Make(PF.RGB_PLANAR, tw, th, GPU) for _ in range(ring)]


def fused_batched():
    for i in range(0, ring, B):
        assert fused.ExecuteBatch(src[i:i + B], out[i:i + B], cc)


t3 = timed(fused_batched)
print(f"[chain] 1080p NV12 -> bilinear 224x224 -> RGB_PLANAR in ONE pass (PySurfaceConvertResizer.ExecuteBatch): {t3 / ring * 1e6:6.2f} us/frame ({ring / t3:8.0f} frames/s)")
