"""fused NV12 -> RGB at the shapes the per-wave strips (k_convert_strip, lab build, variant 47) used to take by policy: product (second pass of the
workgroup strips) against the lab build under variant 47 and variant 0"""
import os, sys, importlib.util
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from videoprocessingframework_amd import capi
spec = importlib.util.spec_from_file_location("videoprocessingframework_amd._capi_forms", os.path.join(ROOT, "videoprocessingframework_amd", "capi.py"))
forms = importlib.util.module_from_spec(spec); sys.modules[spec.name] = forms; spec.loader.exec_module(forms)
forms.LIB_PATH = os.path.join(ROOT, "tools", "lab", "libvpfhip_forms.so"); forms.lib()
argv, sys.argv = sys.argv, sys.argv[:1]
from resize_batch_bench import surf
sys.argv = argv
def run(lib, variant, sw, sh, dw, dh, n, ring=64):
    S = [surf(capi.NV12, sw, sh, True) for _ in range(ring)]
    D = [surf(capi.RGB, dw, dh, False) for _ in range(ring)]
    ex = lib.make_exec(torch.cuda.current_stream().cuda_stream)
    lib.set_tuning(lib.TUNE_NV12_RGB_VARIANT, variant)
    batches = [lib.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + n]]) for i in range(0, ring, n)]
    def go():
        for b in batches: lib.convert_resize_batch(ex, lib.NV12, lib.RGB, 1, 0, sw, sh, dw, dh, b)
    for _ in range(20): go()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): go()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 10 / ring)
    lib.set_tuning(lib.TUNE_NV12_RGB_VARIANT, 0)
    return best
for (sw, sh, dw, dh) in [(1920, 1080, 1163, 654), (1920, 1080, 1066, 750), (1280, 720, 775, 436), (3840, 2160, 2133, 1500), (1920, 1080, 1280, 720)]:
    for n in (1, 32):
        a = run(capi, 0, sw, sh, dw, dh, n); b = run(forms, 47, sw, sh, dw, dh, n); c = run(forms, 0, sw, sh, dw, dh, n)
        print(f"[fused_ab] NV12 {sw}x{sh} -> RGB {dw}x{dh}, {n:2d} frames per dispatch: product {a:6.2f} us/frame | lab build, per-wave strips (47) {b:6.2f} | lab build, policy {c:6.2f}", flush=True)
