#!/usr/bin/env python3
"""Static ISA statistics of one translation unit's gfx950 kernels (no GPU needed): instruction mix per kernel, VGPR / SGPR / LDS use,
occupancy-relevant numbers, and per-loop-body counts.  python tools/isa_stats.py csrc/k_resize.hip [name-substring ...]
python tools/isa_stats.py --spills csrc/k_lanczos_mfma.hip: one line per kernel (VGPRs, spilled VGPRs / SGPRs, scratch bytes) and exit status 1 when any
kernel of the translation unit spills a VGPR or uses scratch (tests/test_isa_no_spills.py: VERDICT r4 item 5)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm_of(src):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm",
                           "-amdgpu-kernarg-preload-count=16", f"-I{ROOT}/include", f"-I{ROOT}/videoprocessingframework_amd/csrc", "-S", "--cuda-device-only",
                           src, "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def klass(i):
    if i.startswith("v_"):
        return "valu"
    if i.startswith("s_"):
        return "salu"
    if i.startswith("ds_"):
        return "lds"
    if i.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def spills(src):
    """[(kernel, vgprs, spilled vgprs, spilled sgprs, scratch bytes)] from the code object's metadata"""
    s = asm_of(src)
    out = []
    for blk in s.split("  - .agpr_count:")[1:]:
        g = lambda key: int(re.search(key + r":\s+(\d+)", blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out.append((subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name, g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"),
                    g(r"\.sgpr_spill_count"), g(r"\.private_segment_fixed_size")))
    return out


def main():
    if sys.argv[1] == "--spills":
        src = sys.argv[2]
        if not os.path.isabs(src):
            src = os.path.join(ROOT, "videoprocessingframework_amd", src) if not os.path.exists(src) else src
        bad = 0
        for name, vg, vs, ss, scr in spills(src):
            print(f"{name[:110]:110s} vgpr {vg:3d} spilled {vs} (sgpr {ss}) scratch {scr} B")
            bad += vs > 0 or scr > 0
        sys.exit(1 if bad else 0)
    src = sys.argv[1]
    if not os.path.isabs(src):
        src = os.path.join(ROOT, "videoprocessingframework_amd", src) if not os.path.exists(src) else src
    want = sys.argv[2:]
    s = asm_of(src)
    meta = {m.group(1): m.group(2) for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, flags=re.S)}
    lines = s.split("\n")
    idx = {l.split(":")[0]: n for n, l in enumerate(lines) if ":" in l and not l.startswith((".", " ", "\t", ";"))}
    for name, md in meta.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        if want and not any(w in dem for w in want):
            continue
        body, n = [], idx[name] + 1
        while n < len(lines) and not lines[n].startswith(".Lfunc_end"):
            body.append(lines[n])
            n += 1
        ins = [(l.strip().split()[0], l) for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))]
        c = collections.Counter(klass(i) for i, _ in ins)
        special = collections.Counter()
        for i, _ in ins:
            for key in ("dot2", "dot4", "fma", "cvt", "perm", "alignb", "div_", "rcp", "mul_lo", "mad_u64", "ds_read", "ds_write", "ds_load", "ds_store", "global_load", "global_store", "s_waitcnt", "s_barrier"):
                if key in i:
                    special[key] += 1
        vg = re.search(r"next_free_vgpr (\d+)", md).group(1)
        sg = re.search(r"next_free_sgpr (\d+)", md).group(1)
        lds = re.search(r"group_segment_fixed_size (\d+)", md).group(1)
        scr = re.search(r"private_segment_fixed_size (\d+)", md).group(1)
        print(f"{dem[:140]}\n    vgpr {vg} sgpr {sg} static-lds {lds} scratch {scr} | static instr: {dict(c)}\n    {dict(special)}")


if __name__ == "__main__":
    main()
