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

  quirks    the reference behaviours this repo knowingly does NOT reproduce are recorded as cases of their own, each with a "quirk" tag
            the loader understands, so that a first run against real NPP output classifies them instead of reporting a generic mismatch:
              R2   PySurfaceResizer on RGB_PLANAR / YUV444: GetSurfacePlane(i) returns the same stacked W x 3H plane for i = 0, 1, 2
                   (src/TC/src/MemoryInterfaces.cpp:1617-1621), so the reference resizes the whole stack as ONE W x 3H image, three times
                   (src/TC/src/Tasks.cpp:1227-1253): rows near the two seams blend pixels of neighbouring planes.  This repo resizes
                   each plane on its own.  The loader checks both readings and says which one the fixture follows.
              C13  RGB -> YUV444 under an MPEG-range context calls the PACKED nppiRGBToYCbCr_8u_C3R on plane 0 of the planar surface
                   (src/TC/src/TasksColorCvt.cpp:758): interleaved Y Cb Cr triples where the Y plane should be.  This repo writes
                   planar YCbCr.  The loader checks whether plane 0's first row starts with packed triples.
              BGR  Surface::Make(BGR) (no-size overload) has no BGR case (src/TC/src/MemoryInterfaces.cpp:596-630): a refused
                   conversion into BGR returns Python None where other formats return an Empty() surface.  Recorded per refused
                   context as refusedkind_<ctx> = 1 (None) / 0 (Empty()).
              R5   PySurfaceResizer on RGB_32F_PLANAR dereferences a null plane for i >= 1 (src/TC/src/Tasks.cpp:1390-1445,
                   MemoryInterfaces.cpp:1815-1818).  Probed in a CHILD process; its exit status is the fixture.
              default context  cc_ctx = None is recorded for every converter (out_none / refused_none): each *_Impl picks its own default.

The same script runs against this repo's drop-in PyNvCodec (same API); fixtures produced that way are marked
"producer": "vpf-hip" in the manifest and are NOT a pin — tests treat them as a rehearsal of the kit only.

    python tests/golden/make_npp_fixtures.py --verify-only [--out DIR]
reads the fixtures back (no GPU, no PyNvCodec: it needs this repository's CPU oracle and tests/test_reference_fixtures.py) and prints,
per fixture and recorded output, which combinations of the oracle's assumption switches A2 / A6 / A8 (oracle/vpf_oracle.h) land within
1 LSB — "the default" when (0, 0, 0) is among them — and how each quirk case classifies.  First contact with NPP output is then a
table to read, not a debugging session.
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


def verify_only(out_dir):
    """Per fixture: the A2 / A6 / A8 (/ A10 for resizes) combinations (oracle EXACT mode) within 1 LSB of every recorded output, the quirk
    classification, and for every recorded down-scale which reading of A10 (Lanczos support when minifying) it follows."""
    root = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import oracle as o
    import test_reference_fixtures as T

    man = json.load(open(os.path.join(out_dir, "manifest.json")))
    print(f"{len(man['cases'])} fixtures, producer '{man['producer']}'" + ("  (self-produced: a rehearsal, not a pin)" if man["producer"] != "nvidia-vpf" else ""))
    default, bad = {"A2": 0, "A6": 0, "A8": 0}, 0
    for name in man["cases"]:
        z = np.load(os.path.join(out_dir, name + ".npz"))
        kind, quirk = str(z["kind"]), (str(z["quirk"]) if "quirk" in z.files else "")
        lines = []
        if kind == "convert":
            sf, df, w, h = str(z["src_fmt"]), str(z["dst_fmt"]), int(z["w"]), int(z["h"])
            for key in (k for k in z.files if k.startswith("out_")):
                res = T.resolve_like_the_reference(sf, df, T._ctx_of(key[4:]))
                if res is None:
                    lines.append(f"ctx {key[4:]}: the reference ACCEPTS it, this repo's dispatch refuses it")
                    continue
                run = lambda: o.convert(getattr(o, sf), getattr(o, df), res[0], res[1], w, h, T.split_planes(o, sf, w, h, z["src"]), o.EXACT)  # noqa: E731
                lines.append(f"ctx {key[4:]}: {T.describe_hits(T.which_assumptions(o, run, z[key]), default)}")
                if quirk == "C13" and key[4:] != "none" and key[5] == "0":   # MPEG range
                    lines.append(f"ctx {key[4:]}: quirk C13 -> {T.classify_c13(o, w, h, z['src'], res, z[key])}")
            for key in (k for k in z.files if k.startswith("refusedkind_")):
                if quirk == "BGR":
                    lines.append(f"ctx {key[12:]}: refused with {'None (quirk BGR: Surface::Make(BGR) is null)' if int(z[key]) else 'an Empty() surface'}")
        elif kind == "resize":
            fmt, w, h = str(z["fmt"]), int(z["w"]), int(z["h"])
            src = T.split_planes(o, fmt, w, h, z["src"])
            for key in (k for k in z.files if k.startswith("out_")):
                dw, dh = (int(v) for v in key[4:].split("x"))
                run = lambda: o.resize(getattr(o, fmt), o.LANCZOS3, w, h, src, dw, dh, o.EXACT)  # noqa: E731
                lines.append(f"-> {dw}x{dh}: {T.describe_hits(T.which_assumptions(o, run, z[key], lanczos=True), default)}")
                if (dw < w or dh < h) and quirk != "R2":
                    lines.append(f"-> {dw}x{dh}: assumption A10 -> {T.classify_a10(o, fmt, w, h, src, dw, dh, z[key])}")
                if quirk == "R2":
                    lines.append(f"-> {dw}x{dh}: quirk R2 -> {T.classify_r2(o, w, h, src, dw, dh, z[key])}")
        elif kind == "remap":
            w, h = int(z["w"]), int(z["h"])
            st, out = o.remap(o.RGB, w, h, T.split_planes(o, "RGB", w, h, z["src"]), z["xmap"], z["ymap"], o.EXACT)
            inside = (z["xmap"] >= 0) & (z["xmap"] <= w - 1) & (z["ymap"] >= 0) & (z["ymap"] <= h - 1)
            mx, frac = T.lsb_report(out[0].reshape(h, w, 3)[inside], z["out"].reshape(h, w, 3)[inside])
            lines.append(f"max |diff| {mx:g} over the mapped pixels" + ("" if mx <= 1 else f"  ({frac:.2%} off by more than 1 LSB)  MISMATCH"))
        elif kind == "quirk":
            lines.append(f"quirk {quirk}: child exit status {int(z['returncode'])} ({'crashed: the reference quirk' if int(z['returncode']) not in (0,) else 'ran'}): {str(z['said']).strip()[-120:]}")
        bad += sum("MISMATCH" in l for l in lines)
        print(name)
        for l in lines:
            print("   ", l)
    print(f"{bad} recorded outputs match no assumption combination" if bad else "every recorded output matches at least one assumption combination")
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--module-dir", default=None, help="directory that contains the PyNvCodec package to import")
    ap.add_argument("--out", default=os.path.join(HERE, "npp"))
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--verify-only", action="store_true", help="read the fixtures in --out back and print which A2 / A6 / A8 combination matches each (needs this repo's oracle, no GPU)")
    a = ap.parse_args()
    if a.verify_only:
        return verify_only(a.out)
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
                        outs["refusedkind_" + key] = np.array(1 if dst is None else 0, np.uint8)  # quirk BGR: None instead of an Empty() surface
                        continue
                    outs["out_" + key] = download(df, w, h, dst)
                    accepted.append(key)
                quirk = "C13" if (sf, df) == ("RGB", "YUV444") else "BGR" if df == "BGR" else ""
                save(f"convert_{sf}_{df}_{w}x{h}_{dist}", kind="convert", src_fmt=sf, dst_fmt=df, w=w, h=h, src=src, quirk=quirk, **outs)
                print(f"convert {sf}->{df} {w}x{h} {dist}: accepted contexts {accepted}")

    # ---- resize (the reference's resizer = NPP Lanczos) ---------------------------------------------------------------
    def resizer(dw, dh, fmt):
        rs = nvc.PySurfaceResizer(dw, dh, getattr(PF, fmt), a.gpu)
        if hasattr(rs, "SetInterpolation"):  # additive in this repo's module (its default is Lanczos too since round 3); the reference always asks NPP for Lanczos
            rs.SetInterpolation(2)
        return rs

    w, h = 848, 464
    for fmt in ("RGB", "RGB_PLANAR", "YUV444", "YUV420", "NV12"):
        src = synth(fmt, w, h, 2000 + len(manifest["cases"]), "A")
        surf = upload(fmt, w, h, src)
        outs = {}
        for dw, dh in ((224, 224), (424, 232), (1280, 720), (283, 155)):
            dst = resizer(dw, dh, fmt).Execute(surf)
            if dst is not None and not dst.Empty():
                outs[f"out_{dw}x{dh}"] = download(fmt, dw, dh, dst)
        save(f"resize_{fmt}_{w}x{h}", kind="resize", fmt=fmt, w=w, h=h, src=src, quirk="R2" if fmt in ("RGB_PLANAR", "YUV444") else "", **outs)
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

    # ---- quirk R5: the float planar resizer, probed in a child process (the reference dereferences a null plane) --------------------
    import subprocess

    probe = ("import sys, numpy as np\n" + (f"sys.path.insert(0, {a.module_dir!r})\n" if a.module_dir else "") +
             "import PyNvCodec as nvc\nPF = nvc.PixelFormat\n"
             f"up = nvc.PyFrameUploader(64, 32, PF.RGB_32F_PLANAR, {a.gpu})\n"
             "s = up.UploadSingleFrame(np.random.default_rng(5).random(3 * 64 * 32, dtype=np.float32))\n"
             f"d = nvc.PySurfaceResizer(32, 16, PF.RGB_32F_PLANAR, {a.gpu}).Execute(s)\n"
             "print('R5 probe:', 'refused' if d is None or d.Empty() else 'resized')\n")
    try:
        r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300)
        rc, said = r.returncode, (r.stdout + r.stderr)[-300:]
    except Exception as e:  # noqa: BLE001
        rc, said = -999, str(e)
    save("quirk_R5_resize_RGB_32F_PLANAR", kind="quirk", quirk="R5", returncode=np.array(rc, np.int32), said=np.array(said))
    print(f"quirk R5 (RGB_32F_PLANAR resize in a child process): exit status {rc}")

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
    sys.exit(main())
