#!/usr/bin/env python3
"""make_npp_fixtures.py — the parity PIN KIT: dump the REAL reference's pixels for this path.

Run this on a machine with an NVIDIA GPU and the reference's own PyNvCodec (NVIDIA/VideoProcessingFramework built against
CUDA + NPP):

    python tests/golden/make_npp_fixtures.py                       # imports PyNvCodec from the environment
    python tests/golden/make_npp_fixtures.py --module-dir /path/to/dir/containing/PyNvCodec

It needs numpy and PyNvCodec only (no oracle, no repo package) and writes tests/golden/npp/*.npz + manifest.json.  Commit
those files; tests/test_reference_fixtures.py then checks the CPU oracle (EXACT mode, under the assumption switches of
oracle/vpf_oracle.h) and — with `-m gpu` — the HIP path against them within +-1 LSB.  Until that has happened the arithmetic
of this repo is "parity unpinned" (the reference delegates every pixel to closed-source NPP and ships no golden frames).

What is recorded (inputs travel inside the fixture, so nothing has to be regenerated identically elsewhere):
  convert   every format pair the reference's ConvertSurface constructs (src/TC/src/TasksColorCvt.cpp:1313-1360) that can be
            fed through PyFrameUploader, under cc_ctx = None and all nine (ColorSpace, ColorRange) contexts, at 64x32;
            refused contexts are recorded as refused (the reference returns an Empty() surface).  The headline pairs also at
            848x464 (the size of the reference's tests/test.mp4, tests/test_PySurface.py:55-64) for three input distributions
  resize    PySurfaceResizer (NPP Lanczos, src/TC/src/Tasks.cpp:1190) 848x464 -> 224x224 / 424x232 / 1280x720 / 283x155 for
            RGB, RGB_PLANAR, YUV420, NV12; plus a 16x16 impulse image -> 40x40 and 5x5 (reads off the filter's coordinate
            convention and tap weights directly)
  remap     PySurfaceRemaper (nppiRemap_8u_C3R linear, Tasks.cpp:1590-1595) on 848x464 RGB: identity, half-pixel shift, barrel
            distortion r' = r (1 + 0.1 r^2), and a map with out-of-range entries over a pre-filled destination

The same script runs against this repo's drop-in PyNvCodec (same API); fixtures produced that way are marked
"producer": "vpf-hip" in the manifest and are NOT a pin — tests treat them as a rehearsal of the kit only.
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

PAIRS = [  # the reference's ConvertSurface ctor, in its order; P10/P12 -> NV12 omitted (SurfaceP10 cannot be uploaded: 8-bit allocation)
    ("NV12", "YUV420"), ("YUV420", "NV12"), ("NV12", "RGB"), ("NV12", "BGR"), ("RGB", "RGB_PLANAR"), ("RGB_PLANAR", "RGB"),
    ("RGB_PLANAR", "YUV444"), ("Y", "YUV444"), ("YUV420", "RGB"), ("RGB", "YUV420"), ("RGB", "YUV444"), ("BGR", "YCBCR"),
    ("RGB", "BGR"), ("BGR", "RGB"), ("YUV420", "BGR"), ("YUV444", "BGR"), ("YUV444", "RGB"), ("BGR", "YUV444"), ("NV12", "Y"),
    ("RGB", "RGB_32F"), ("RGB", "Y"), ("RGB_32F", "RGB_32F_PLANAR"),
]
HEADLINE = [("NV12", "RGB"), ("NV12", "BGR"), ("YUV420", "RGB"), ("RGB", "YUV420"), ("RGB", "YUV444"), ("YUV444", "RGB")]


def host_size(fmt, w, h):
    """elements of the tight host frame (planes concatenated; CudaUploadFrame / CudaDownloadSurface, Tasks.cpp:643-658,746-763)"""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return {"Y": w * h, "RGB": 3 * w * h, "BGR": 3 * w * h, "RGB_PLANAR": 3 * w * h, "YUV444": 3 * w * h, "NV12": w * h + 2 * cw * ch,
            "YUV420": w * h + 2 * cw * ch, "YCBCR": w * h + 2 * cw * ch, "RGB_32F": 3 * w * h, "RGB_32F_PLANAR": 3 * w * h}[fmt]


def synth(fmt, w, h, seed, dist="A"):
    """A: uniform 0..255; B: legal video (luma 16..235, chroma 16..240); C: ramps (row / column / plane mix-ups show)"""
    rng = np.random.default_rng(seed)
    n = host_size(fmt, w, h)
    if fmt.startswith("RGB_32F"):
        return rng.random(n, dtype=np.float32)
    if dist == "A" or fmt in ("RGB", "BGR", "RGB_PLANAR", "Y") and dist == "B":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if dist == "B":
        a = rng.integers(16, 241, n, dtype=np.uint8)
        a[:w * h] = rng.integers(16, 236, w * h, dtype=np.uint8)
        return a
    i = np.arange(n, dtype=np.int64)
    return ((i % max(w, 1)) + 3 * (i // max(w, 1))).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--module-dir", default=None, help="directory that contains the PyNvCodec package to import")
    ap.add_argument("--out", default=os.path.join(HERE, "npp"))
    ap.add_argument("--gpu", type=int, default=0)
    a = ap.parse_args()
    if a.module_dir:
        sys.path.insert(0, a.module_dir)
    import PyNvCodec as nvc

    PF = nvc.PixelFormat
    os.makedirs(a.out, exist_ok=True)
    producer = "vpf-hip" if hasattr(nvc, "ConverterResolve") else "nvidia-vpf"  # ConverterResolve exists only in this repo's module
    manifest = {"producer": producer, "module": getattr(nvc, "__file__", "?"), "numpy": np.__version__, "cases": []}

    def upload(fmt, w, h, frame):
        return nvc.PyFrameUploader(w, h, getattr(PF, fmt), a.gpu).UploadSingleFrame(frame)

    def download(fmt, w, h, surf):
        out = np.zeros(1, np.float32 if fmt.startswith("RGB_32F") else np.uint8)
        ok = nvc.PySurfaceDownloader(w, h, getattr(PF, fmt), a.gpu).DownloadSingleSurface(surf, out)
        if not ok:
            raise RuntimeError(f"download of {fmt} {w}x{h} failed")
        return out

    def save(name, **kw):
        np.savez_compressed(os.path.join(a.out, name + ".npz"), **kw)
        manifest["cases"].append(name)

    # ---- converters -------------------------------------------------------------------------------------------------
    ctxs = [None] + [(cs, cr) for cs in range(3) for cr in range(3)]
    for sizes, pairs, dists in (((64, 32), PAIRS, "A"), ((848, 464), HEADLINE, "ABC")):
        w, h = sizes
        for sf, df in pairs:
            try:
                conv = nvc.PySurfaceConverter(w, h, getattr(PF, sf), getattr(PF, df), a.gpu)
            except Exception as e:  # noqa: BLE001  (the reference throws invalid_argument for pairs it does not build)
                print(f"skip {sf}->{df}: {e}")
                continue
            for dist in dists:
                src = synth(sf, w, h, 1000 + len(manifest["cases"]), dist)
                surf = upload(sf, w, h, src)
                outs, accepted = {}, []
                for c in (ctxs if (w, h) == (64, 32) else [None, (1, 0), (1, 1), (0, 0), (0, 1)]):
                    cc = None if c is None else nvc.ColorspaceConversionContext(nvc.ColorSpace(c[0]), nvc.ColorRange(c[1]))
                    dst = conv.Execute(surf, cc)
                    key = "none" if c is None else f"{c[0]}{c[1]}"
                    if dst is None or dst.Empty():
                        outs["refused_" + key] = np.zeros(0, np.uint8)
                        continue
                    outs["out_" + key] = download(df, w, h, dst)
                    accepted.append(key)
                save(f"convert_{sf}_{df}_{w}x{h}_{dist}", kind="convert", src_fmt=sf, dst_fmt=df, w=w, h=h, src=src, **outs)
                print(f"convert {sf}->{df} {w}x{h} {dist}: accepted contexts {accepted}")

    # ---- resize (the reference's resizer = NPP Lanczos) ---------------------------------------------------------------
    def resizer(dw, dh, fmt):
        rs = nvc.PySurfaceResizer(dw, dh, getattr(PF, fmt), a.gpu)
        if hasattr(rs, "SetInterpolation"):  # this repo's module defaults to bilinear (north_star); the reference always asks NPP for Lanczos
            rs.SetInterpolation(2)
        return rs

    w, h = 848, 464
    for fmt in ("RGB", "RGB_PLANAR", "YUV420", "NV12"):
        src = synth(fmt, w, h, 2000 + len(manifest["cases"]), "A")
        surf = upload(fmt, w, h, src)
        outs = {}
        for dw, dh in ((224, 224), (424, 232), (1280, 720), (283, 155)):
            dst = resizer(dw, dh, fmt).Execute(surf)
            if dst is not None and not dst.Empty():
                outs[f"out_{dw}x{dh}"] = download(fmt, dw, dh, dst)
        save(f"resize_{fmt}_{w}x{h}", kind="resize", fmt=fmt, w=w, h=h, src=src, **outs)
        print(f"resize {fmt}: {sorted(outs)}")
    imp = np.zeros((16, 16, 3), np.uint8)
    imp[5, 7] = (255, 128, 64)       # one lit pixel: the output IS the filter's footprint
    imp[12, 2] = (32, 255, 200)
    outs = {}
    for dw, dh in ((40, 40), (5, 5), (16, 16), (32, 32), (8, 8)):
        dst = resizer(dw, dh, "RGB").Execute(upload("RGB", 16, 16, imp.reshape(-1)))
        if dst is not None and not dst.Empty():
            outs[f"out_{dw}x{dh}"] = download("RGB", dw, dh, dst)
    save("resize_RGB_impulse_16x16", kind="resize", fmt="RGB", w=16, h=16, src=imp.reshape(-1), **outs)

    # ---- remap ---------------------------------------------------------------------------------------------------------
    src = synth("RGB", w, h, 3000, "A")
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    nx, ny = (xx - (w - 1) / 2) / ((w - 1) / 2), (yy - (h - 1) / 2) / ((h - 1) / 2)
    k = 1 + 0.1 * (nx * nx + ny * ny)
    maps = {"identity": (xx, yy), "shift_half": (xx + 0.5, yy + 0.5),
            "barrel": ((nx * k * ((w - 1) / 2) + (w - 1) / 2).astype(np.float32), (ny * k * ((h - 1) / 2) + (h - 1) / 2).astype(np.float32)),
            "partly_outside": (xx * 1.25 - 40, yy * 1.25 - 30)}
    for name, (mx, my) in maps.items():
        mx, my = np.ascontiguousarray(mx, np.float32), np.ascontiguousarray(my, np.float32)
        dst = nvc.PySurfaceRemaper(mx, my, PF.RGB, a.gpu).Execute(upload("RGB", w, h, src))
        if dst is None or dst.Empty():
            print(f"remap {name}: refused")
            continue
        save(f"remap_RGB_{name}_{w}x{h}", kind="remap", fmt="RGB", w=w, h=h, src=src, xmap=mx, ymap=my, out=download("RGB", w, h, dst))
        print(f"remap {name}: ok")

    json.dump(manifest, open(os.path.join(a.out, "manifest.json"), "w"), indent=1)
    print(f"{len(manifest['cases'])} fixtures from producer '{producer}' in {a.out}")


if __name__ == "__main__":
    main()
