"""-m gpu: caller frame buffers that come back are page-locked where they lie and DMA'd from there (Tasks.hpp HostPinCache) —
and a buffer that was freed and reallocated at the same address is a NEW buffer, never served from a stale registration.
Reference: PyFrameUploader copies straight from the caller's numpy buffer (src/TC/src/Tasks.cpp:625-662)."""
import ctypes
import gc
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no GPU visible", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
import PyNvCodec as nvc  # noqa: E402

# Only buffers that own their pages are page-locked (glibc serves them by mmap: PyNvCodec.cpp owns_its_pages).  glibc raises its mmap threshold
# whenever a mapped chunk is freed, so in a process that has freed large arrays later ones are cut from the heap; a process that wants its frame
# pool taken allocates it up front — or pins the threshold, like here (M_MMAP_THRESHOLD = -3).
assert ctypes.CDLL(None).mallopt(-3, 128 * 1024) == 1
_parked = []


def owns_its_pages(a):
    """what PyNvCodec.cpp's owns_its_pages() asks of a buffer: a glibc chunk served by mmap (user pointer 16 bytes into a page, IS_MMAPPED in the size word)"""
    p = a.ctypes.data
    return p % 4096 == 16 and (ctypes.c_size_t.from_address(p - 8).value & 2) != 0


def mapped(make):
    """an array whose memory is a mapping of its own.  Even above the mmap threshold malloc first looks for a free chunk on the heap (a long-lived
    process has some): heap-cut results are parked — they use those chunks up — until the allocator has to map"""
    for _ in range(200):
        a = make()
        if owns_its_pages(a):
            return a
        _parked.append(a)
    raise AssertionError("the allocator never mapped a chunk")


PF = nvc.PixelFormat
W, H = 1920, 1080
N = W * H * 3 // 2


def download(surf):
    dl = nvc.PySurfaceDownloader(surf.Width(), surf.Height(), surf.Format(), 0)
    out = np.zeros(1, np.uint8)
    assert dl.DownloadSingleSurface(surf, out)
    return out


def delta(before):
    now = nvc.PinCacheStats()
    return {k: int(now[k]) - int(before[k]) for k in now}


def test_a_buffer_seen_twice_is_registered_and_read_in_place():
    nvc.PinCacheDrop()
    up = nvc.PyFrameUploader(W, H, PF.NV12, 0)
    rng = np.random.default_rng(5)
    frame = mapped(lambda: rng.integers(0, 256, N, dtype=np.uint8))   # owns its data (somebody to vouch for the memory) and its pages
    s0 = nvc.PinCacheStats()
    assert np.array_equal(download(up.UploadSingleFrame(frame)), frame)
    d = delta(s0)
    assert d["staged"] == 1 and d["in_place"] == 0 and d["registered"] == 0          # first sight: the staged copy
    for k in range(3):                                                                # a decoder refilling its buffer
        frame[:] = rng.integers(0, 256, N, dtype=np.uint8)
        assert np.array_equal(download(up.UploadSingleFrame(frame)), frame), k
    d = delta(s0)
    assert d["registered"] == 1 and d["in_place"] == 3 and d["staged"] == 1 and d["bytes"] == N
    view = frame[:]                                                                   # a view: the same owner, the same memory
    assert np.array_equal(download(up.UploadSingleFrame(view)), frame) and delta(s0)["in_place"] == 4
    del view, frame
    gc.collect()
    d = delta(s0)
    assert d["registered"] == 0 and d["bytes"] == 0                                   # the owner died: nothing stays page-locked on its behalf


def test_freed_and_reallocated_at_the_same_address_is_a_new_buffer():
    nvc.PinCacheDrop()
    up = nvc.PyFrameUploader(W, H, PF.NV12, 0)
    rng = np.random.default_rng(6)
    a = mapped(lambda: rng.integers(0, 256, N, dtype=np.uint8))
    for _ in range(2):
        assert np.array_equal(download(up.UploadSingleFrame(a)), a)
    assert int(nvc.PinCacheStats()["registered"]) == 1
    addr = a.ctypes.data
    want_a = a.copy()
    del a
    gc.collect()
    assert int(nvc.PinCacheStats()["registered"]) == 0
    same = None
    keep = []
    for _ in range(64):                                                               # glibc hands the mapping back for the same size, usually at once
        b = np.empty(N, np.uint8)
        if b.ctypes.data == addr:
            same = b
            break
        keep.append(b)
    del keep
    if same is None:
        pytest.skip("the allocator did not return the old address")
    same[:] = rng.integers(0, 256, N, dtype=np.uint8)
    assert not np.array_equal(same, want_a)
    s0 = nvc.PinCacheStats()
    got = download(up.UploadSingleFrame(same))
    assert np.array_equal(got, same)                                                  # the NEW pages' bytes, not the old registration's
    d = delta(s0)
    assert d["staged"] == 1 and d["in_place"] == 0                                    # ... and a first sight again
    assert np.array_equal(download(up.UploadSingleFrame(same)), same) and delta(s0)["in_place"] == 1


def test_memory_nobody_vouches_for_keeps_the_staged_copy():
    nvc.PinCacheDrop()
    up = nvc.PyFrameUploader(W, H, PF.NV12, 0)
    raw = bytearray(os.urandom(N))
    f = np.frombuffer(raw, dtype=np.uint8)                                            # the root of the .base chain is not a numpy array that owns its data
    s0 = nvc.PinCacheStats()
    for _ in range(3):
        assert np.array_equal(download(up.UploadSingleFrame(f)), f)
    d = delta(s0)
    assert d["registered"] == 0 and d["in_place"] == 0
    pinned = nvc.AllocPinned(N)                                                       # page-locked already: DMA'd in place without the cache
    pinned[:] = f
    for _ in range(3):
        assert np.array_equal(download(up.UploadSingleFrame(pinned)), f)
    assert delta(s0)["registered"] == 0


def test_heap_cut_buffers_keep_the_staged_copy():
    """a buffer cut from the heap shares its first and last page with its neighbours, and unregistering works on whole pages (the GPU fault of
    profiles/r06_pin_cache_fault.txt): such buffers are never page-locked"""
    nvc.PinCacheDrop()
    libc = ctypes.CDLL(None)
    assert libc.mallopt(-3, 16 << 20) == 1                                            # M_MMAP_THRESHOLD (glibc caps it at 32 MiB): the next arrays come from the heap
    try:
        a = np.zeros(N, np.uint8)
        a[:] = 7
    finally:
        assert libc.mallopt(-3, 128 * 1024) == 1
    up = nvc.PyFrameUploader(W, H, PF.NV12, 0)
    s0 = nvc.PinCacheStats()
    for _ in range(3):
        assert np.array_equal(download(up.UploadSingleFrame(a)), a)
    d = delta(s0)
    assert d["registered"] == 0 and d["in_place"] == 0
    b = mapped(lambda: np.zeros(N, np.uint8))                                         # a chunk of whole pages of its own
    assert b.ctypes.data % 4096 == 16
    for _ in range(3):
        assert np.array_equal(download(up.UploadSingleFrame(b)), b)
    assert delta(s0)["registered"] == 1


def test_least_recently_used_buffers_leave_a_full_cache():
    nvc.PinCacheDrop()
    w, h = 640, 360
    n = w * h * 3 // 2                                                                # 345 600 B: above the cache's 256-KiB floor
    up = nvc.PyFrameUploader(w, h, PF.NV12, 0)
    rng = np.random.default_rng(7)
    pool = [mapped(lambda: rng.integers(0, 256, n, dtype=np.uint8)) for _ in range(70)]   # more buffers than the cache has entries (64)
    s0 = nvc.PinCacheStats()
    for rnd in range(2):
        for i, f in enumerate(pool):
            for rep in range(2):  # twice in a row: page-locked the second time (a plain cycle through more buffers than entries would never see one twice)
                assert np.array_equal(download(up.UploadSingleFrame(f)), f), (rnd, i, rep)
    d = delta(s0)
    assert 0 < d["registered"] <= 64 and d["evictions"] > 0
    del pool, f
    gc.collect()
    assert int(nvc.PinCacheStats()["registered"]) == 0


def test_async_uploads_keep_the_staged_copy():
    """SetAsync(True): the call returns with the copy only QUEUED and its contract lets the caller reuse an ordinary frame at once — so such uploads are
    never read in place: not registered on their own, and copied out first when a blocking uploader had the buffer page-locked earlier"""
    nvc.PinCacheDrop()
    rng = np.random.default_rng(8)
    frame = mapped(lambda: rng.integers(0, 256, N, dtype=np.uint8))
    blocking = nvc.PyFrameUploader(W, H, PF.NV12, 0)
    for _ in range(2):
        assert np.array_equal(download(blocking.UploadSingleFrame(frame)), frame)
    assert int(nvc.PinCacheStats()["registered"]) == 1                                # page-locked by the blocking uploader
    up = nvc.PyFrameUploader(W, H, PF.NV12, 0)
    up.SetAsync(True)
    s0 = nvc.PinCacheStats()
    wants, surfs = [], []
    for k in range(6):
        frame[:] = rng.integers(0, 256, N, dtype=np.uint8)
        wants.append(frame.copy())
        surfs.append(up.UploadSingleFrame(frame).Clone(0))                            # ... and the frame is overwritten right away, next iteration
    torch.cuda.synchronize()
    for k in range(6):
        assert np.array_equal(download(surfs[k]), wants[k]), k                        # every upload saw ITS bytes: they were copied out before the call returned
    d = delta(s0)
    assert d["in_place"] == 0 and d["registered"] == 0
    fresh = [mapped(lambda: rng.integers(0, 256, N, dtype=np.uint8)) for _ in range(2)]
    for rnd in range(3):
        for f in fresh:
            up.UploadSingleFrame(f)
    torch.cuda.synchronize()
    assert delta(s0)["registered"] == 0                                               # asynchronous uploads do not register buffers
    up.SetAsync(True, in_place=True)                                                  # ... unless the caller promises to leave frames alone until they are consumed
    for rnd in range(3):
        for f in fresh:
            last = up.UploadSingleFrame(f).Clone(0)
    torch.cuda.synchronize()
    assert delta(s0)["registered"] == 2 and delta(s0)["in_place"] >= 4 and np.array_equal(download(last), fresh[-1])
    del last
    del frame, fresh, f, surfs
    gc.collect()
    assert int(nvc.PinCacheStats()["registered"]) == 0
