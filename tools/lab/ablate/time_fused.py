"""time_fused.py LIB [LIB ...]: us/frame of the batched fused NV12 -> bilinear -> RGB (32 frames per dispatch, rings past the Infinity Cache) with each kernel library,
one subprocess per library and pass, interleaved, minimum of the passes; VPF_LAB_FUSED_NB is passed through (lab builds of the march form)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CASES = ((1920, 1080, 1280, 720), (1920, 1080, 3840, 2160), (1280, 720, 1920, 1080), (3840, 2160, 2560, 1440), (3840, 2160, 1600, 900))


def child(libp):
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from videoprocessingframework_amd import capi
    capi.LIB_PATH = os.path.abspath(libp)
    sys.argv = sys.argv[:1]
    from resize_batch_bench import surf, timed
    ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
    out = []
    for sw, sh, dw, dh in CASES:
        ring = max(32, min(128, int(600e6 // (sw * sh * 3 // 2 + dw * dh * 3)) // 32 * 32))
        S = [surf(capi.NV12, sw, sh, True) for _ in range(ring)]
        D = [surf(capi.RGB, dw, dh, False) for _ in range(ring)]
        batches = [capi.make_batch([(s[1], d[1]) for s, d in list(zip(S, D))[i:i + 32]]) for i in range(0, ring, 32)]
        out.append(timed(lambda: [capi.convert_resize_batch(ex, capi.NV12, capi.RGB, 1, 0, sw, sh, dw, dh, b) for b in batches], 5) / ring)
        del S, D, batches
        torch.cuda.empty_cache()
    print("RESULT " + " ".join(f"{t:.3f}" for t in out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2]); sys.exit(0)
    libs = sys.argv[1:]
    best = {l: None for l in libs}
    for _ in range(int(os.environ.get("AB_PASSES", "3"))):
        for l in libs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True)
            line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")]
            if not line:
                print(f"[fused-ab] {l}: FAILED\n{r.stderr[-600:]}"); continue
            v = [float(x) for x in line[0].split()[1:]]
            best[l] = v if best[l] is None else [min(a, b) for a, b in zip(best[l], v)]
    print("[fused-ab] " + " " * 24 + " | ".join(f"{c[0]}x{c[1]}->{c[2]}x{c[3]}" for c in CASES))
    for l in libs:
        if best[l]:
            print(f"[fused-ab] {os.path.basename(l):24s}" + " | ".join(f"{t:20.2f}" for t in best[l]), flush=True)
