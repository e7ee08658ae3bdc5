import ctypes, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "videoprocessingframework_amd"))
import torch
import PyNvCodec as nvc
print("mallopt", ctypes.CDLL(None).mallopt(-3, 128 * 1024), "env", {k: v for k, v in os.environ.items() if "MALLOC" in k or "GLIBC" in k or "LD_PRELOAD" in k})
rng = np.random.default_rng(5)
N = 1920 * 1080 * 3 // 2
for i in range(3):
    a = rng.integers(0, 256, N, dtype=np.uint8)
    p = a.ctypes.data
    print(i, hex(p), p % 4096, hex(ctypes.c_size_t.from_address(p - 8).value) if p % 4096 >= 8 else None)
up = nvc.PyFrameUploader(1920, 1080, nvc.PixelFormat.NV12, 0)
nvc.PinCacheSetBudgetMB(1024)
for i in range(3):
    up.UploadSingleFrame(a)
    print(dict(nvc.PinCacheStats()))
