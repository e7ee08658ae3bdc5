#!/usr/bin/env python3
"""clip_pipeline.py — BASELINE.json config 1's runner (and the per-clip leg of config 4): one clip through
    PyFfmpegDecoder (libav demux + software decode on the host -> real NV12)  ->  PyFrameUploader  ->  PySurfaceConverter NV12 -> RGB
through the drop-in Python API.  Reference pattern: tests/test_PySurface.py:55-64 + src/TC/src/FfmpegSwDecoder.cpp:254-360 (the reference's
only software component is the decode; its conversion is NPP on the GPU).

Three rates are reported, with the host core count (`cores` = CPUs this process may use):
  decode_only       frames / s of demux + decode + NV12 repack alone (host)
  end_to_end        decode -> upload -> convert, PCIe-inclusive
  device_resident   convert only, over the surfaces of the clip's first frames kept on the device
Every converted frame is checked: its RGB bytes must equal a second, independent conversion of the same uploaded surface by the generic
any-alignment kernel family (tuning 9) — and `--dump-crc FILE` writes the CRC-32 of every decoded NV12 frame and of its RGB conversion, which
tests/test_clip_pipeline.py compares with the oracle (the oracle is test infrastructure: this tool never imports it).

The decoder exists only where the bindings were built against libav (PyNvCodec.HAVE_LIBAV; this image has none).  `--pynvcodec DIR` points
at another build of the package — the test suite passes tests/_build/pynvcodec_stubav, whose decoder is linked against the stub libav of
tests/libav_stub and "decodes" synthetic clips (`--clip synth:w=..,h=..,n=..`).

  python tools/clip_pipeline.py --clip tests/test.mp4 [--frames N] [--decode-only] [--gpu 0]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_nvc(pkg_dir: str | None):
    sys.path.insert(0, ROOT)  # videoprocessingframework_amd.capi (the tuning hook of the cross-check)
    sys.path.insert(0, pkg_dir if pkg_dir else os.path.join(ROOT, "videoprocessingframework_amd"))
    import PyNvCodec as nvc

    return nvc


def decode_only(nvc, clip: str, max_frames: int):
    """-> (frames, seconds, [crc32 of every NV12 frame], (w, h))"""
    dec = nvc.PyFfmpegDecoder(clip, {}, 0)
    frame = np.zeros(1, np.uint8)
    crcs, n = [], 0
    t0 = time.perf_counter()
    while n < max_frames and dec.DecodeSingleFrame(frame):
        n += 1
    dt = time.perf_counter() - t0
    dec = nvc.PyFfmpegDecoder(clip, {}, 0)  # CRCs outside the timed loop
    for _ in range(n):
        assert dec.DecodeSingleFrame(frame)
        crcs.append(zlib.crc32(frame.tobytes()))
    return n, dt, crcs, (dec.Width(), dec.Height())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clip", required=True)
    ap.add_argument("--frames", type=int, default=1 << 30, help="stop after this many frames")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--decode-only", action="store_true", help="host leg only (runs without a GPU)")
    ap.add_argument("--pynvcodec", default=None, help="directory that holds the PyNvCodec package to use (default: this repo's build)")
    ap.add_argument("--dump-crc", default=None, help="write {'nv12': [...], 'rgb': [...]} CRC-32 lists here (for the test suite's oracle check)")
    a = ap.parse_args()
    nvc = load_nvc(a.pynvcodec)
    if not getattr(nvc, "HAVE_LIBAV", False):
        print(json.dumps({"runner": "clip_pipeline", "clip": a.clip, "error": "this PyNvCodec build has no libav decoder (HAVE_LIBAV is False): "
                          "build where libavformat / libavcodec exist, or pass --pynvcodec with such a build"}), flush=True)
        return 2
    cores = len(os.sched_getaffinity(0))
    n, dt, nv12_crc, (w, h) = decode_only(nvc, a.clip, a.frames)
    out = {"runner": "clip_pipeline", "clip": a.clip, "size": f"{w}x{h}", "frames": n, "cores": cores,
           "decode_only": {"frames_per_s": round(n / dt, 1) if dt > 0 else None, "Gpix_per_s": round(n * w * h / dt / 1e9, 4) if dt > 0 else None}}
    crc = {"nv12": nv12_crc, "rgb": []}
    if not a.decode_only:
        import torch

        from videoprocessingframework_amd import capi

        pf = nvc.PixelFormat
        dec = nvc.PyFfmpegDecoder(a.clip, {}, a.gpu)
        cs, cr = dec.ColorSpace(), dec.ColorRange()
        if cs == nvc.ColorSpace.UNSPEC:
            cs = nvc.ColorSpace.BT_709 if h >= 720 else nvc.ColorSpace.BT_601   # what the reference's samples assume for untagged streams
        if cr == nvc.ColorRange.UDEF:
            cr = nvc.ColorRange.MPEG
        nvc.SetExtendedColorspaces(True)  # BT.601 + MPEG is not reachable from NV12 in the reference's converter (TasksColorCvt.cpp:156-163)
        cc = nvc.ColorspaceConversionContext(cs, cr)
        conv = nvc.PySurfaceConverter(w, h, pf.NV12, pf.RGB, a.gpu)
        chk = nvc.PySurfaceConverter(w, h, pf.NV12, pf.RGB, a.gpu)
        dl = nvc.PySurfaceDownloader(w, h, pf.RGB, a.gpu)
        got, want = np.zeros(1, np.uint8), np.zeros(1, np.uint8)
        # --- end to end, timed: decode -> upload -> convert (the converter's output surface is reused: consume it per frame)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = 0
        while m < n:
            s = dec.DecodeSingleSurface()
            if s.Empty():
                break
            rgb = conv.Execute(s, cc)
            if rgb.Empty():
                raise SystemExit("conversion failed")
            m += 1
        torch.cuda.synchronize()
        e2e = time.perf_counter() - t0
        # --- every frame again, untimed, checked: default kernels vs the generic family on the same uploaded surface
        dec = nvc.PyFfmpegDecoder(a.clip, {}, a.gpu)
        keep, verified = [], 0
        for i in range(m):
            s = dec.DecodeSingleSurface()
            rgb = conv.Execute(s, cc)
            assert dl.DownloadSingleSurface(rgb, got)
            prev = capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 9)
            try:
                assert dl.DownloadSingleSurface(chk.Execute(s, cc), want)
            finally:
                capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, prev)
            c = zlib.crc32(got.tobytes())
            if c != zlib.crc32(want.tobytes()):
                raise SystemExit(f"frame {i}: the default and the generic converter disagree")
            crc["rgb"].append(c)
            verified += 1
            if len(keep) < 32:
                keep.append(s.Clone(a.gpu))  # deep copies: the uploader's slots are recycled
        # --- device resident: convert the kept surfaces round-robin
        torch.cuda.synchronize()
        reps = max(1, 256 // max(1, len(keep)))
        t0 = time.perf_counter()
        for _ in range(reps):
            for s in keep:
                conv.Execute(s, cc)
        torch.cuda.synchronize()
        res = time.perf_counter() - t0
        out.update({"colour": [int(cs), int(cr)], "verified_frames": verified,
                    "end_to_end": {"frames_per_s": round(m / e2e, 1), "Gpix_per_s": round(m * w * h / e2e / 1e9, 4)},
                    "device_resident": {"frames_per_s": round(reps * len(keep) / res, 1), "Gpix_per_s": round(reps * len(keep) * w * h / res / 1e9, 3)}})
    if a.dump_crc:
        json.dump(crc, open(a.dump_crc, "w"))
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
