"""One dispatch per frame, split into KERNEL time and the GAP between kernels: run under rocprofv3 --kernel-trace, then `--report <dir>`.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sf -o s -- python tools/single_frame_trace.py
  python tools/single_frame_trace.py --report gpurun_out/sf
Per-frame resizes (Lanczos / bilinear, RGB / NV12 / Y) over rings past the Infinity Cache, each shape separated by a marker dispatch (a
tiny NV12->RGB conversion, 64 x 16) so the report can cut the trace into sections.  VPF_BENCH_MFMA forces a Lanczos launch shape."""
import csv
import glob
import os
import statistics as st
import sys

CASES = (("RGB", 2, 1920, 1080, 1280, 720), ("RGB", 1, 1920, 1080, 1280, 720), ("NV12", 2, 1920, 1080, 1280, 720), ("NV12", 1, 1920, 1080, 1280, 720),
         ("Y", 2, 1920, 1080, 1280, 720), ("Y", 1, 1920, 1080, 1280, 720), ("RGB", 2, 1920, 1080, 416, 416), ("RGB", 2, 1280, 720, 1920, 1080),
         ("RGB", 2, 3840, 2160, 1920, 1080), ("RGB", 1, 3840, 2160, 1920, 1080))


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    sections, cur = [], []
    for r in rows:
        if "k_nv12_rgb" in r["Kernel_Name"] or "k_yuv2rgb" in r["Kernel_Name"]:
            if cur:
                sections.append(cur)
            cur = []
        elif "vpf::" in r["Kernel_Name"] and "k_lzm_build" not in r["Kernel_Name"]:  # (torch's fill / random kernels of the next case's allocations are not ours)
            cur.append(r)
    if cur:
        sections.append(cur)
    sections = [s for s in sections if len(s) >= 32]
    for case, s in zip(CASES, sections[-len(CASES):]):
        s = s[len(s) // 4:]  # the warm part
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s]
        gap = [int(s[i + 1]["Start_Timestamp"]) - int(s[i]["End_Timestamp"]) for i in range(len(s) - 1)]
        per = [int(s[i + 1]["Start_Timestamp"]) - int(s[i]["Start_Timestamp"]) for i in range(len(s) - 1)]
        names = sorted({r["Kernel_Name"].split("(")[0][-60:] for r in s})
        wg = {(r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("LDS_Block_Size", "?")) for r in s}
        print(f"[single-frame] {case[0]:5s} {case[2]}x{case[3]}->{case[4]}x{case[5]} interp {case[1]}: kernel {st.median(dur) / 1e3:6.2f} us | gap {st.median(gap) / 1e3:5.2f} us | "
              f"start to start {st.median(per) / 1e3:6.2f} us  ({len(s)} dispatches; {', '.join(names)}; grid x/y/z, lds {sorted(wg)[:3]})")


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from videoprocessingframework_amd import capi
    import resize_batch_bench as rb  # noqa: F401  (surf)
    ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
    capi.set_tuning(capi.TUNE_RESIZE_MFMA, int(os.environ.get("VPF_BENCH_MFMA", "0"), 0))
    dev = torch.device("cuda", 0)
    my = torch.randint(0, 256, (24, 256), dtype=torch.uint8, device=dev)
    mo = torch.zeros((16, 256), dtype=torch.uint8, device=dev)

    def marker():
        capi.convert(ex, capi.NV12, capi.RGB, capi.BT_709, capi.MPEG, 64, 16, [(my.data_ptr(), 256), (my.data_ptr() + 16 * 256, 256)], [(mo.data_ptr(), 256)])

    fmts = {"RGB": capi.RGB, "NV12": capi.NV12, "Y": capi.Y}
    for name, interp, sw, sh, dw, dh in CASES:
        fmt = fmts[name]
        ring = max(32, min(128, int(600e6 // (sw * sh * 3 + dw * dh * 3)) // 32 * 32))
        S = [rb.surf(fmt, sw, sh, True) for _ in range(ring)]
        D = [rb.surf(fmt, dw, dh, False) for _ in range(ring)]
        planes = [(capi.planes(s[1]), capi.planes(d[1])) for s, d in zip(S, D)]
        torch.cuda.synchronize()
        marker()
        for _ in range(4):
            for s, d in planes:
                capi.resize(ex, fmt, interp, sw, sh, s, dw, dh, d)
        torch.cuda.synchronize()
        del S, D, planes
    marker()
    torch.cuda.synchronize()


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()
