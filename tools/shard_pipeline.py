#!/usr/bin/env python3
"""shard_pipeline.py — BASELINE.json config 4's runner: S independent 4K clips sharded one-clip-per-GPU, each rank running
host NV12 frames -> PyFrameUploader (pinned staging + side copy stream) -> PySurfaceConverter NV12->RGB through the drop-in
Python API.  Clip s goes to rank s mod N (sharding.assign_clips); ranks never exchange data (no collective on the data
path) and meet only at the timing barriers.

Reference pattern: samples/SampleDecodeMultiThread.py:50-115 (one decoder + converter chain per GPU) over CudaResMgr's
per-GPU context/stream (src/PyNvCodec/src/PyNvCodec.cpp:57-111).

The decode stage is a STAND-IN: this image has no libav, so each clip's "decoder" is a generator that hands out frames
already sitting in AllocPinned() buffers (what a software decoder writing into page-locked memory would produce), cycling
over a few distinct frames per clip.  Two rates are reported (SURVEY §8e asks for both):
  end_to_end       host frame -> upload -> convert, PCIe-inclusive (never the headline `value` of bench.py)
  device_resident  convert only, on the surfaces already uploaded

  python tools/shard_pipeline.py [--clips 8] [--frames 64]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         tools/shard_pipeline.py --gpus N [--backend gloo]   (gloo: several ranks may share one GPU, for testing)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
from videoprocessingframework_amd import sharding  # noqa: E402
import PyNvCodec as nvc  # noqa: E402


def _async(up):
    """the pipeline's uploaders return once their copy is queued (additive SetAsync; the default waits for the copy like the reference): every
    consumer here runs on the uploader's own stream, and the stand-in decoder never rewrites a frame the stream has not consumed"""
    up.SetAsync(True, in_place=True)  # (pageable frames that come back are page-locked and read where they lie: the stand-in decoder keeps the promise)
    return up


_parked = []


def _mapped_empty(n):
    """np.empty(n) whose memory is a mapping of its own (what HostPinCache takes: a buffer that owns its pages).  glibc raises its mmap threshold
    whenever a mapped chunk is freed — pinned here at 128 KiB (M_MMAP_THRESHOLD = -3) — and even above it malloc prefers a free heap chunk when it
    has one: such results are parked until the allocator has to map"""
    import ctypes
    ctypes.CDLL(None).mallopt(-3, 128 * 1024)
    for _ in range(64):
        a = np.empty(n, np.uint8)
        p = a.ctypes.data
        if p % 4096 == 16 and (ctypes.c_size_t.from_address(p - 8).value & 2):
            return a
        _parked.append(a)
    return a


class SyntheticClip:
    """Stand-in for demux + software decode of one clip: `distinct` NV12 frames in host memory, seeded per clip — page-locked (what a decoder
    writing into AllocPinned() buffers produces: DMA'd in place) or ordinary pageable numpy arrays (what an unmodified decoder produces: the
    uploader stages them through its own pinned slots, one host copy per frame, and returns as soon as the DMA is queued)."""

    def __init__(self, clip_id, w, h, distinct=4, pinned=True):
        rng = np.random.default_rng(7000 + clip_id)
        self.frames = []
        for _ in range(distinct):
            buf = nvc.AllocPinned(w * h * 3 // 2) if pinned else _mapped_empty(w * h * 3 // 2)
            buf[:] = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
            self.frames.append(buf)

    def decode(self, i):
        return self.frames[i % len(self.frames)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--frames", type=int, default=64, help="frames per clip")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--source", choices=["pinned", "pageable"], default="pinned", help="where the stand-in decoder leaves its frames")
    ap.add_argument("--threads", action="store_true", help="one worker thread per clip (the reference sample's shape: the GIL is released inside "
                    "upload / convert, so the staging copies of different clips run in parallel) instead of one thread round-robin")
    ap.add_argument("--no-pin-cache", action="store_true", help="leave the page-locked caller-buffer cache off (its default): pageable frames are staged through a copy")
    ap.add_argument("--no-numa", action="store_true", help="do not confine the rank to the CPUs of its GPU's NUMA node")
    a = ap.parse_args()
    rank, world, local = sharding.env_rank()
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE is {world}")
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU (no CPU fallback)")
    gpu = local % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    numa = {"bound": False, "why": "--no-numa"} if a.no_numa else sharding.bind_to_gpu_numa(gpu)  # BEFORE pinned buffers and threads exist
    sharding.init(a.backend, dev)
    red_dev = dev if a.backend == "nccl" else None
    w, h, pf = a.width, a.height, nvc.PixelFormat
    cc = nvc.ColorspaceConversionContext(nvc.ColorSpace.BT_709, nvc.ColorRange.MPEG)
    if not a.no_pin_cache and not os.environ.get("VPF_HIP_PIN_CACHE_MB"):
        nvc.PinCacheSetBudgetMB(1024)  # opt in: pageable frames that come back are page-locked where they lie (Tasks.hpp HostPinCache)
    mine = sharding.assign_clips(a.clips, world, rank)
    chains = []
    for c in mine:  # one stream + task chain per clip, like one worker thread per stream in the reference sample
        stream = torch.cuda.Stream(device=dev)
        ctx = nvc.GetContext(gpu)
        chains.append({"clip": c, "src": SyntheticClip(c, w, h, pinned=a.source == "pinned"), "stream": stream,
                       "up": _async(nvc.PyFrameUploader(w, h, pf.NV12, ctx, stream.cuda_stream)),
                       "conv": nvc.PySurfaceConverter(w, h, pf.NV12, pf.RGB, ctx, stream.cuda_stream),
                       "down": nvc.PySurfaceDownloader(w, h, pf.RGB, ctx, stream.cuda_stream)})

    def one(ch, i, end_to_end, last):
        if end_to_end or "nv12" not in ch:
            ch["nv12"] = ch["up"].UploadSingleFrame(ch["src"].decode(i))
        rgb = ch["conv"].Execute(ch["nv12"], cc)
        if rgb.Empty():
            raise SystemExit(f"rank {rank}: conversion failed on clip {ch['clip']}")
        last[ch["clip"]] = (i, rgb)

    def run(end_to_end, frames):
        last = {}
        if a.threads and len(chains) > 1:
            import threading

            def worker(ch):
                for i in range(frames):
                    one(ch, i, end_to_end, last)
            ts = [threading.Thread(target=worker, args=(ch,)) for ch in chains]
            [t.start() for t in ts]
            [t.join() for t in ts]
        else:
            for i in range(frames):
                for ch in chains:  # round-robin over this rank's clips: their uploads and kernels overlap across streams
                    one(ch, i, end_to_end, last)
        torch.cuda.synchronize(dev)
        return last

    run(True, 8)  # warm-up (every distinct frame of every clip twice: a pageable buffer is page-locked the second time it is seen)
    rates = {}
    pin = {}
    for key, e2e in (("end_to_end", True), ("device_resident", False)):
        sharding.barrier(dev)
        pin[key] = nvc.PinCacheStats()
        t0 = time.perf_counter()
        last = run(e2e, a.frames)
        own = time.perf_counter() - t0
        pin[key] = {k: int(v) - int(pin[key][k]) for k, v in nvc.PinCacheStats().items()}
        sharding.barrier(dev)
        px, t = sharding.aggregate(len(mine) * a.frames * w * h, time.perf_counter() - t0, red_dev)
        rates[key] = {"value": round(px / t / 1e9, 3), "unit": "Gpix/s", "frames_per_s": round(px / t / (w * h), 1),
                      "per_rank_s": [round(x, 4) for x in sharding.gather(own, red_dev)]}
    # every clip's last frame: download and compare against the same conversion of the host frame this rank fed, done
    # by a second, independent converter instance on the same GPU (catches cross-clip / cross-stream mix-ups)
    verified = 0
    chk = nvc.PySurfaceConverter(w, h, pf.NV12, pf.RGB, gpu)
    up2, dl2 = nvc.PyFrameUploader(w, h, pf.NV12, gpu), nvc.PySurfaceDownloader(w, h, pf.RGB, gpu)
    for ch in chains:
        i, rgb = last[ch["clip"]]
        got, want = np.zeros(1, np.uint8), np.zeros(1, np.uint8)
        assert ch["down"].DownloadSingleSurface(rgb, got)
        assert dl2.DownloadSingleSurface(chk.Execute(up2.UploadSingleFrame(ch["src"].decode(i)), cc), want)
        if zlib.crc32(got.tobytes()) != zlib.crc32(want.tobytes()):
            raise SystemExit(f"rank {rank}: clip {ch['clip']} frame {i} differs from its own source's conversion")
        verified += 1
    verified_all, _ = sharding.aggregate(verified, 0.0, red_dev)
    clips_per_rank = [sharding.assign_clips(a.clips, world, r) for r in range(world)]
    # What one uploaded frame costs the HOST's memory system (round 5: the 8-GPU end-to-end rate is set by host DRAM, not by the GPUs).  A frame in
    # page-locked memory is read once, by the DMA engine.  A pageable frame is read by the uploader's host copy, written into a pinned slot
    # (non-temporal stores: no read-for-ownership) and read again by the DMA engine: three times its size.  Next to it, what one core of this
    # box copies per second (one pass over 256 MiB, best of three): the scale for "how many ranks can one socket feed".
    frame_bytes = w * h * 3 // 2
    # Round 6: a pageable buffer that comes back is page-locked where it lies (Tasks.hpp HostPinCache) and read once, by the DMA engine, like a
    # pinned one: what this rank's timed end-to-end run really did is counted (uploads DMA'd in place / staged through a copy).
    ups = len(mine) * a.frames
    staged = ups - pin["end_to_end"]["in_place"] if a.source == "pageable" else 0
    host_bytes = frame_bytes * (3 * staged + (ups - staged)) // max(1, ups)
    probe_src, probe_dst = np.ones(1 << 28, np.uint8), np.empty(1 << 28, np.uint8)
    copy_gbps = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        np.copyto(probe_dst, probe_src)
        copy_gbps = max(copy_gbps, (1 << 28) / (time.perf_counter() - t0) / 1e9)
    del probe_src, probe_dst
    if rank == 0:
        print(json.dumps({"runner": "shard_pipeline", "n_gpus": world, "clips": a.clips, "clips_per_rank": clips_per_rank,
                          "frames_total": a.clips * a.frames, "size": f"{w}x{h}", "verified_clips": int(verified_all),
                          "decode": f"stand-in: frames pre-decoded into {'AllocPinned()' if a.source == 'pinned' else 'pageable numpy'} host buffers "
                                    "(no libav in this image; tools/clip_pipeline.py runs a real decoder where libav exists)",
                          "threads": "one per clip" if a.threads else "one, round-robin", "numa": numa,
                          "bytes_per_s_end_to_end": round(rates["end_to_end"]["frames_per_s"] * w * h * 1.5 / 1e9, 2),
                          "host_memory": {"source": a.source, "frame_bytes": frame_bytes, "host_dram_bytes_per_frame": host_bytes,
                                          "uploads_rank0": ups, "staged_rank0": staged, "pin_cache_rank0": pin["end_to_end"],
                                          "host_dram_GBps_at_this_rate": round(rates["end_to_end"]["frames_per_s"] * host_bytes / 1e9, 1),
                                          "one_core_copy_GBps": round(copy_gbps, 1),
                                          "model": "DMA'd in place (pinned source, or a pageable buffer seen before: page-locked on second sight): 1 x frame; staged: 3 x (host copy read + pinned-slot write + DMA read)"},
                          "end_to_end": rates["end_to_end"], "device_resident": rates["device_resident"],
                          "sharding": "clip s -> rank s mod N; no data-path collective"}), flush=True)
    if world > 1:
        sharding.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
