"""The Python API surface of the in-scope classes, checked against the reference's own type stub
(src/PyNvCodec/__init__.pyi -> tests/golden/reference_api_surface.json via tests/golden/make_api_surface.py): every class,
every method, every overload's parameter names must exist in this repo's compiled PyNvCodec with the same keyword names,
so reference user code calling with keywords keeps working.  Additive members (ExecuteBatch, Wrap, ...) are allowed."""
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "videoprocessingframework_amd"))
nvc = pytest.importorskip("PyNvCodec")
API = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_api_surface.json")))


def overloads(fn):
    """parameter-name lists of every overload in a pybind11 docstring"""
    doc = fn.__doc__ or ""
    out = []
    for m in re.finditer(r"^\s*(?:\d+\.\s*)?\w+\((.*?)\)\s*(?:->.*)?$", doc, re.M):
        names = []
        depth, cur = 0, ""
        for ch in m.group(1) + ",":  # split on top-level commas
            if ch in "[(":
                depth += 1
            elif ch in "])":
                depth -= 1
            if ch == "," and depth == 0:
                if cur.strip():
                    names.append(cur.strip().split(":")[0].strip().lstrip("*"))
                cur = ""
            else:
                cur += ch
        out.append([n for n in names if n not in ("self", "cls", "arg0")])
    return out


@pytest.mark.parametrize("cls", sorted(API["classes"]))
def test_class_surface(cls):
    mine = getattr(nvc, cls, None)
    assert mine is not None, f"class {cls} missing"
    for name, ref_overloads in API["classes"][cls].items():
        fn = getattr(mine, name, None)
        assert fn is not None, f"{cls}.{name} missing"
        have = overloads(fn)
        for ov in ref_overloads:
            if ov["opaque"]:  # the stub only says (*args, **kwargs): existence is all it pins
                continue
            # same leading keyword names; extra trailing parameters must be optional additions of ours
            assert any(h[:len(ov["params"])] == ov["params"] for h in have), f"{cls}.{name}{tuple(ov['params'])} not among {have}"


def test_enums_and_functions():
    for enum, members in API["enums"].items():
        e = getattr(nvc, enum)
        for m in members:
            assert hasattr(e, m), f"{enum}.{m} missing"
    for fn in API["functions"]:
        assert callable(getattr(nvc, fn))
