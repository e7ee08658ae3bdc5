"""The reference's own Surface test (tests/test_PySurface.py:67-195) transplanted: same three test cases, same
ground-truth constants (:55-64), same assertions.  Two substitutions, both forced by the platform: the 96 frames come
from a seeded synthetic 848x464 NV12 clip through PyFrameUploader instead of NVDEC decoding tests/test.mp4 (NVDEC is
NVIDIA fixed-function hardware and this image has no H.264 decoder), and the pitched 2-D device copies that the reference
does with pycuda.Memcpy2D are done with SurfacePlane.Export (the same hipMemcpy2DAsync underneath)."""
import os
import sys
import unittest

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no GPU visible", allow_module_level=True)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "videoprocessingframework_amd"))
import PyNvCodec as nvc  # noqa: E402

# Ground truth information about input video (tests/test_PySurface.py:55-64)
gt_width = 848
gt_height = 464
gt_pix_fmt = nvc.PixelFormat.NV12
gt_num_frames = 96
gt_color_space = nvc.ColorSpace.BT_709
gt_color_range = nvc.ColorRange.MPEG


class SyntheticDecoder:
    """Stands in for PyNvDecoder: DecodeSingleSurface() hands out surfaces that belong to an internal pool (here the
    uploader's two slots), an Empty() surface at end of stream; DecodeSingleFrame() the same frames as host arrays."""

    def __init__(self, gpu_id):
        self.up = nvc.PyFrameUploader(gt_width, gt_height, gt_pix_fmt, gpu_id)
        self.i = self.j = 0

    @staticmethod
    def frame(i):
        return np.random.default_rng(4000 + i).integers(0, 256, gt_width * gt_height * 3 // 2, dtype=np.uint8)

    def Width(self): return gt_width
    def Height(self): return gt_height
    def Format(self): return gt_pix_fmt

    def DecodeSingleSurface(self):
        if self.i >= gt_num_frames:
            return nvc.PySurfaceConverter(gt_width, gt_height, gt_pix_fmt, nvc.PixelFormat.RGB, 0).Execute(None, None)  # Empty()
        self.i += 1
        return self.up.UploadSingleFrame(self.frame(self.i - 1))

    def DecodeSingleFrame(self, out):
        if self.j >= gt_num_frames:
            return False
        f = self.frame(self.j)
        self.j += 1
        out.resize(f.size, refcheck=False)
        out[:] = f
        return True


class TestSurfaceHip(unittest.TestCase):
    def setUp(self):
        self.gpu_id = 0
        self.stream = torch.cuda.Stream()
        self.ctx, self.str = nvc.GetContext(self.gpu_id), self.stream.cuda_stream
        self.nvDec = SyntheticDecoder(self.gpu_id)
        self.nvDwn = nvc.PySurfaceDownloader(self.nvDec.Width(), self.nvDec.Height(), self.nvDec.Format(), self.ctx, self.str)

    def test_memcpy_Surface_Surface(self):  # reference: test_pycuda_memcpy_Surface_Surface (:86-122)
        n = 0
        while True:
            surf_src = self.nvDec.DecodeSingleSurface()
            if surf_src.Empty():
                break
            n += 1
            src_plane = surf_src.PlanePtr()
            surf_dst = nvc.Surface.Make(self.nvDec.Format(), self.nvDec.Width(), self.nvDec.Height(), self.gpu_id)
            self.assertFalse(surf_dst.Empty())
            dst_plane = surf_dst.PlanePtr()
            self.assertEqual((src_plane.Width(), src_plane.Height()), (gt_width, gt_height * 3 // 2))  # raw plane: W x 1.5H
            src_plane.Export(dst_plane.GpuMem(), dst_plane.Pitch(), self.ctx, self.str)
            frame_src = np.ndarray(shape=(0), dtype=np.uint8)
            if not self.nvDwn.DownloadSingleSurface(surf_src, frame_src):
                self.fail("Failed to download decoded surface")
            frame_dst = np.ndarray(shape=(0), dtype=np.uint8)
            if not self.nvDwn.DownloadSingleSurface(surf_dst, frame_dst):
                self.fail("Failed to download decoded surface")
            if not np.array_equal(frame_src, frame_dst):
                self.fail("Video frames are not equal")
        self.assertEqual(n, gt_num_frames)

    def test_memcpy_Surface_Tensor(self):  # reference: test_pycuda_memcpy_Surface_Tensor (:124-161)
        while True:
            surf_src = self.nvDec.DecodeSingleSurface()
            if surf_src.Empty():
                break
            src_plane = surf_src.PlanePtr()
            surface_tensor = torch.zeros(src_plane.Height(), src_plane.Width(), 1, dtype=torch.uint8,
                                         device=torch.device(f"cuda:{self.gpu_id}"))
            torch.cuda.synchronize()  # torch's zero-fill runs on torch's stream, the copy on ours: order them (the
            # reference test has this latent race too; it copies on a pycuda stream right after torch.zeros)
            src_plane.Export(surface_tensor.data_ptr(), self.nvDec.Width(), self.ctx, self.str)
            frame_src = np.ndarray(shape=(0), dtype=np.uint8)
            if not self.nvDwn.DownloadSingleSurface(surf_src, frame_src):
                self.fail("Failed to download decoded surface")
            frame_dst = surface_tensor.to("cpu").numpy().reshape((src_plane.Height() * src_plane.Width()))
            if not np.array_equal(frame_src, frame_dst):
                self.fail("Video frames are not equal")

    def test_list_append(self):  # reference: test_list_append (:163-195)
        dec_frames = []
        nvDec = SyntheticDecoder(0)
        while True:
            surf = nvDec.DecodeSingleSurface()
            if not surf or surf.Empty():
                break
            # surfaces returned by the decoder belong to its internal pool: clone them
            dec_frames.append(surf.Clone(self.gpu_id))
        self.assertEqual(len(dec_frames), gt_num_frames)
        nvDec = SyntheticDecoder(0)
        nvDwn = nvc.PySurfaceDownloader(nvDec.Width(), nvDec.Height(), nvDec.Format(), self.gpu_id)
        for surf in dec_frames:
            dec_frame = np.ndarray(shape=(0), dtype=np.uint8)
            svd_frame = np.ndarray(shape=(0), dtype=np.uint8)
            nvDwn.DownloadSingleSurface(surf, svd_frame)
            nvDec.DecodeSingleFrame(dec_frame)
            self.assertTrue(np.array_equal(dec_frame, svd_frame))
