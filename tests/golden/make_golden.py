#!/usr/bin/env python3
"""Generates tests/golden/*.json — known-answer vectors for the surface-conversion path.

PARITY UNPINNED: the reference's arithmetic is closed-source NVIDIA NPP and the reference holds no
golden frames for this path, so these vectors are NOT reference outputs.  They
are computed here by a second, independent restatement — pure Python `fractions.Fraction`, written
directly from the formulas NVIDIA publishes for NPP's colour models ("YUV", "YCbCr") and from the
BT.709 matrices — so that oracle/vpf_oracle.c (C, integer arithmetic) is checked against something
that shares no code with it.  Rounding: round-half-up, then clamp to [0,255].

Run:  python tests/golden/make_golden.py        (no dependency on /root/reference, oracle/ or the GPU)
"""
import json
import os
from fractions import Fraction as F
from math import floor

HERE = os.path.dirname(os.path.abspath(__file__))


def rhu(v: F) -> int:  # round half up + clamp
    return max(0, min(255, floor(v + F(1, 2))))


# (color_space, color_range) -> published decimal coefficients, as strings so Fraction is exact
YUV2RGB = {
    # NPP "YCbCr" model (nppiYCbCr420ToRGB etc.): limited range BT.601
    "601_MPEG": dict(cy="1.164", off=16, rv="1.596", gu="-0.392", gv="-0.813", bu="2.017"),
    # NPP "YUV" model (nppiYUVToRGB / nppiNV12ToRGB): full range
    "601_JPEG": dict(cy="1", off=0, rv="1.140", gu="-0.394", gv="-0.581", bu="2.032"),
    # BT.709 limited range ("709CSC"): 255/219 luma, Kr=.2126 Kb=.0722 scaled by 255/224
    "709_MPEG": dict(cy="1.164384", off=16, rv="1.792741", gu="-0.213249", gv="-0.532909", bu="2.112402"),
    # BT.709 full range ("709HDTV")
    "709_JPEG": dict(cy="1", off=0, rv="1.5748", gu="-0.187324", gv="-0.468124", bu="1.8556"),
}
CS = {"601": 0, "709": 1}
CR = {"MPEG": 0, "JPEG": 1}


def yuv2rgb(key, y, u, v):
    m = YUV2RGB[key]
    yy = F(m["cy"]) * (y - m["off"])
    uu, vv = u - 128, v - 128
    return [rhu(yy + F(m["rv"]) * vv), rhu(yy + F(m["gu"]) * uu + F(m["gv"]) * vv), rhu(yy + F(m["bu"]) * uu)]


def rgb2yuv(rng, r, g, b):
    if rng == "JPEG":  # NPP "YUV": U, V computed from the UNROUNDED luma
        y = F("0.299") * r + F("0.587") * g + F("0.114") * b
        return [rhu(y), rhu(F("0.492") * (b - y) + 128), rhu(F("0.877") * (r - y) + 128)]
    y = F("0.257") * r + F("0.504") * g + F("0.098") * b + 16
    cb = F("-0.148") * r + F("-0.291") * g + F("0.439") * b + 128
    cr = F("0.439") * r + F("-0.368") * g + F("-0.071") * b + 128
    return [rhu(y), rhu(cb), rhu(cr)]


def lcg(seed):
    s = seed
    while True:
        s = (s * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        yield (s >> 33) & 0xFF


def main():
    # --- per-pixel KATs: corners, primaries, legal-range extremes, grey ramp, pseudo-random ---
    triples = []
    for y in (0, 16, 128, 235, 255):
        for u in (0, 16, 128, 240, 255):
            for v in (0, 16, 128, 240, 255):
                triples.append((y, u, v))
    triples += [(k, 128, 128) for k in range(0, 256, 15)]
    g = lcg(2024)
    triples += [(next(g), next(g), next(g)) for _ in range(400)]
    px = {}
    for key in YUV2RGB:
        px[key] = [[y, u, v] + yuv2rgb(key, y, u, v) for (y, u, v) in triples]
    rgbs = [(r, gg, b) for r in (0, 1, 127, 128, 254, 255) for gg in (0, 128, 255) for b in (0, 127, 255)]
    g = lcg(7)
    rgbs += [(next(g), next(g), next(g)) for _ in range(400)]
    inv = {rng: [[r, gg, b] + rgb2yuv(rng, r, gg, b) for (r, gg, b) in rgbs] for rng in ("MPEG", "JPEG")}
    with open(os.path.join(HERE, "pixel_kat.json"), "w") as f:
        json.dump({"yuv2rgb": px, "rgb2yuv": inv, "cs": CS, "cr": CR}, f, separators=(",", ":"))

    # --- one small NV12 frame (10 x 6, odd-ish width to exercise chroma replication) per matrix ---
    w, h = 10, 6
    g = lcg(99)
    ypl = [[next(g) for _ in range(w)] for _ in range(h)]
    uvpl = [[next(g) for _ in range(w)] for _ in range(h // 2)]
    frames = {}
    for key in YUV2RGB:
        rgb = []
        for yy in range(h):
            row = []
            for xx in range(w):
                u, v = uvpl[yy // 2][2 * (xx // 2)], uvpl[yy // 2][2 * (xx // 2) + 1]
                row += yuv2rgb(key, ypl[yy][xx], u, v)
            rgb.append(row)
        frames[key] = rgb
    with open(os.path.join(HERE, "nv12_frame_kat.json"), "w") as f:
        json.dump({"w": w, "h": h, "y": ypl, "uv": uvpl, "rgb": frames}, f, separators=(",", ":"))

    # --- bilinear resize KAT: 5x4 single channel -> 3x3 and -> 8x7, exact rationals ---
    sw, sh = 5, 4
    g = lcg(5)
    src = [[next(g) for _ in range(sw)] for _ in range(sh)]

    def resize(dw, dh):
        out = []
        for dy in range(dh):
            sy = min(max((F(2 * dy + 1, 2)) * F(sh, dh) - F(1, 2), 0), sh - 1)
            y0 = floor(sy); y1 = min(y0 + 1, sh - 1); fy = sy - y0
            row = []
            for dx in range(dw):
                sx = min(max((F(2 * dx + 1, 2)) * F(sw, dw) - F(1, 2), 0), sw - 1)
                x0 = floor(sx); x1 = min(x0 + 1, sw - 1); fx = sx - x0
                top = src[y0][x0] + fx * (src[y0][x1] - src[y0][x0])
                bot = src[y1][x0] + fx * (src[y1][x1] - src[y1][x0])
                row.append(rhu(top + fy * (bot - top)))
            out.append(row)
        return out

    with open(os.path.join(HERE, "resize_kat.json"), "w") as f:
        json.dump({"sw": sw, "sh": sh, "src": src, "to_3x3": resize(3, 3), "to_8x7": resize(8, 7),
                   "to_5x4": resize(5, 4)}, f, separators=(",", ":"))
    print("wrote pixel_kat.json nv12_frame_kat.json resize_kat.json")


if __name__ == "__main__":
    main()
