"""Extracts the Python API surface of the in-scope classes from the reference's own type stub
(/root/reference/src/PyNvCodec/__init__.pyi: class -> method -> list of overloads, each a list of parameter names) into
tests/golden/reference_api_surface.json.  Names only; run in the build container (the GPU box has no /root/reference)."""
import ast
import json
import os

SRC = "/root/reference/src/PyNvCodec/__init__.pyi"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_api_surface.json")
SCOPE = ["ColorspaceConversionContext", "CudaBuffer", "Surface", "SurfacePlane", "PySurfaceConverter", "PySurfaceResizer",
         "PySurfaceRemaper", "PyFrameUploader", "PySurfaceDownloader", "PyBufferUploader", "PyCudaBufferDownloader"]
ENUMS = ["PixelFormat", "ColorSpace", "ColorRange"]

tree = ast.parse(open(SRC).read())
api, enums, functions = {}, {}, {}
for node in tree.body:
    if isinstance(node, ast.ClassDef) and node.name in SCOPE:
        methods = {}
        for f in node.body:
            if isinstance(f, ast.FunctionDef) and not f.name.startswith("__") or (isinstance(f, ast.FunctionDef) and f.name == "__init__"):
                params = [a.arg for a in f.args.args if a.arg not in ("self", "cls")]
                star = bool(f.args.vararg or f.args.kwarg)
                methods.setdefault(f.name, []).append({"params": params, "opaque": star})
        api[node.name] = methods
    elif isinstance(node, ast.ClassDef) and node.name in ENUMS:
        enums[node.name] = sorted(t.target.id for t in node.body if isinstance(t, ast.AnnAssign) and isinstance(t.target, ast.Name)
                                  and t.target.id.isupper() or (isinstance(t, ast.AnnAssign) and isinstance(t.target, ast.Name) and t.target.id[:1].isupper() and t.target.id not in ("name", "value")))
    elif isinstance(node, ast.FunctionDef) and node.name in ("GetNumGpus",):
        functions[node.name] = [a.arg for a in node.args.args]
json.dump({"source": "src/PyNvCodec/__init__.pyi", "classes": api, "enums": enums, "functions": functions}, open(OUT, "w"), indent=1, sort_keys=True)
print("wrote", OUT, {k: len(v) for k, v in api.items()}, enums)
