"""The product library holds the kernels some policy path can select, nothing else.

libvpfhip.so's kernel symbols (`nm -C`: the host-side launch stubs, one per instantiation) are compared with the sources as the PRODUCT build
sees them — every `#ifdef VPF_LAB_FORMS` block cut out: each kernel family in the binary must be named by a launch in those sources, and none of
the forms that moved to the lab build (tools/lab/libvpfhip_forms.so: the persistent band launch, the two-role Lanczos form, the fused kernel's
per-wave strips) may be in it."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "videoprocessingframework_amd", "libvpfhip.so")
CSRC = os.path.join(ROOT, "videoprocessingframework_amd", "csrc")
LAB_ONLY = ("k_planes_mp_persist", "LanczosPairTask", "LzPair", "k_convert_strip<")


def product_view(text):
    """the text the product build compiles: `#ifdef VPF_LAB_FORMS ... [#else ...] #endif` reduced to its #else part, `#ifndef` to its body"""
    out, stack = [], []  # stack entries: [kind, keep_now] for blocks opened on VPF_LAB_FORMS, None for any other conditional
    for line in text.split("\n"):
        t = line.strip()
        if re.match(r"#\s*ifdef\s+VPF_LAB_FORMS\b", t):
            stack.append(["lab", False]); continue
        if re.match(r"#\s*ifndef\s+VPF_LAB_FORMS\b", t):
            stack.append(["lab", True]); continue
        if re.match(r"#\s*if", t):
            stack.append(None)
        elif re.match(r"#\s*else\b", t) and stack and stack[-1] is not None:
            stack[-1][1] = not stack[-1][1]; continue
        elif re.match(r"#\s*endif\b", t):
            top = stack.pop()
            if top is not None:
                continue
        if all(e is None or e[1] for e in stack):
            out.append(line)
    assert not stack
    return "\n".join(out)


def stubs():
    if not os.path.exists(LIB):
        pytest.skip("libvpfhip.so not built")
    r = subprocess.run(["nm", "-C", LIB], capture_output=True, text=True, check=True).stdout
    return [l.split("__device_stub__", 1)[1] for l in r.split("\n") if "__device_stub__" in l]


def test_no_lab_only_form_is_in_the_product_library():
    names = stubs()
    assert len(names) > 300
    for n in names:
        for lab in LAB_ONLY:
            assert lab not in n, n


def test_every_kernel_family_in_the_library_is_launched_by_the_product_sources():
    src = ""
    for d, _, fs in os.walk(CSRC):
        for f in fs:
            if f.endswith((".hip", ".h")) and f != "vpf_persist.h":  # (host side of the persistent launch: included by lab builds only)
                src += product_view(open(os.path.join(d, f)).read()) + "\n"
    code = "\n".join(l.split("//")[0] for l in src.split("\n"))
    for lab in ("k_planes_mp_persist", "LanczosPairTask", "launch_planes_mp_persist", "k_convert_strip,", "vpf_persist.h"):
        assert lab not in code, lab  # (comments may mention them; code the product compiles may not)
    launches = set(re.findall(r"\b(k_[a-z0-9_]+)\b", " ".join(re.findall(r"(?:hipLaunchKernelGGL|VPF_LAUNCH(?:_BA)?|VPF_LAUNCH_[A-Z0-9_]+)\s*\((.*)", code))))
    families = sorted({re.match(r"(k_[a-z0-9_]+)", n).group(1) for n in stubs()})
    assert len(families) > 40
    missing = [f for f in families if f not in launches and not re.search(r"\b" + f + r"\b\s*<", code.replace("void " + f, ""))]
    assert not missing, missing


def test_product_view_of_conditionals():
    t = "a\n#ifdef VPF_LAB_FORMS\nb\n#else\nc\n#endif\nd\n#ifndef VPF_LAB_FORMS\ne\n#endif\n#ifdef OTHER\nf\n#endif\n"
    assert product_view(t).split() == ["a", "c", "d", "e", "#ifdef", "OTHER", "f", "#endif"]
