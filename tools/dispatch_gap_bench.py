"""What a stream pays between two batched dispatches: 32-frame vpf_resize_batch dispatches back to back on ONE stream against the same
dispatches alternating over TWO streams (independent frames), us per frame; rocprofv3's kernel time is the floor of both.
python tools/dispatch_gap_bench.py [interp]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from videoprocessingframework_amd import capi
interp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sys.argv = sys.argv[:1]
from resize_batch_bench import surf
streams = [torch.cuda.Stream() for _ in range(4)]
for fmt, name, (sw, sh, dw, dh) in ((capi.RGB, "RGB", (1920, 1080, 1280, 720)), (capi.NV12, "NV12", (1920, 1080, 1280, 720)), (capi.RGB, "RGB", (3840, 2160, 1920, 1080))):
    ring = 128 if sw < 3000 else 64
    S = [surf(fmt, sw, sh, True) for _ in range(ring)]
    D = [surf(fmt, dw, dh, False) for _ in range(ring)]
    batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + 32]]) for i in range(0, ring, 32)]
    out = []
    for ns in (1, 2, 4):
        exs = [capi.make_exec(st.cuda_stream) for st in streams[:ns]]
        def run(reps):
            for r in range(reps):
                for i, b in enumerate(batches):
                    capi.resize_batch(exs[i % ns], fmt, interp, sw, sh, dw, dh, b)
        run(2); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); run(10); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / (10 * ring) * 1e6)
        out.append(f"{ns} stream(s) {sorted(ts)[2]:.2f}")
    print(f"[dispatch-gap] {name} {sw}x{sh}->{dw}x{dh} interp {interp}, 32 frames per dispatch, us/frame: " + " | ".join(out), flush=True)
    del S, D, batches
    torch.cuda.empty_cache()
