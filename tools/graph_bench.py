"""hipGraph capture of the per-frame API: N single-frame vpf_convert launches captured once (torch.cuda.CUDAGraph on the
capturing stream = hipStreamBeginCapture) and replayed, vs the same launches issued one by one, vs one batched dispatch.
The C ABI never synchronises or allocates, so it is capturable as is."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi

dev = torch.device("cuda", 0)
for (w, h, dfmt, name) in ((1920, 1080, capi.RGB_PLANAR, "1080p NV12->RGB_PLANAR"), (3840, 2160, capi.RGB, "4K NV12->RGB")):
    N = 32
    pitch = (w + 255) // 256 * 256
    dpitch = pitch if dfmt == capi.RGB_PLANAR else (3 * w + 255) // 256 * 256
    drows = 3 * h if dfmt == capi.RGB_PLANAR else h
    src = [torch.randint(0, 256, (h * 3 // 2, pitch), dtype=torch.uint8, device=dev) for _ in range(N)]
    dst = [torch.zeros((drows, dpitch), dtype=torch.uint8, device=dev) for _ in range(N)]
    sdesc = [[(s.data_ptr(), pitch), (s.data_ptr() + h * pitch, pitch)] for s in src]
    ddesc = [[(d.data_ptr() + i * h * dpitch, dpitch) for i in range(3)] if dfmt == capi.RGB_PLANAR else [(d.data_ptr(), dpitch)] for d in dst]
    st = torch.cuda.Stream()
    ex = capi.make_exec(st.cuda_stream)

    def per_frame():
        for a, b in zip(sdesc, ddesc):
            capi.convert(ex, capi.NV12, dfmt, capi.BT_709, capi.MPEG, w, h, a, b)

    batch = capi.make_batch(list(zip(sdesc, ddesc)))

    def batched():
        capi.convert_batch(ex, capi.NV12, dfmt, capi.BT_709, capi.MPEG, w, h, batch)

    with torch.cuda.stream(st):
        per_frame(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            per_frame()
        ref = [d.clone() for d in dst]
        for d in dst:
            d.zero_()
        g.replay(); st.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, dst)), "graph replay differs"

        # the frames are independent: fork the captured work over 4 side streams so the graph holds 4 parallel chains and
        # the ~2 us drain/fill gap between dependent kernels overlaps with a neighbour chain's kernel
        side = [torch.cuda.Stream() for _ in range(4)]
        exs = [capi.make_exec(q.cuda_stream) for q in side]
        g4 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g4, stream=st):
            for q in side:
                q.wait_stream(st)
            for i, (a_, b_) in enumerate(zip(sdesc, ddesc)):
                capi.convert(exs[i % 4], capi.NV12, dfmt, capi.BT_709, capi.MPEG, w, h, a_, b_)
            for q in side:
                st.wait_stream(q)
        for d in dst:
            d.zero_()
        g4.replay(); st.synchronize()
        assert all(torch.equal(a_, b_) for a_, b_ in zip(ref, dst)), "forked graph replay differs"

        def timed(fn, reps=20):
            fn(); st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st); st.synchronize()
            return e0.elapsed_time(e1) * 1e3 / (reps * N)

        a, b, c, d4 = timed(per_frame), timed(g.replay), timed(batched), timed(g4.replay)
    px = w * h
    print(f"[graph] {name}: per-frame launches {a:6.2f} us/frame ({px / a / 1e3:6.0f} Gpix/s) | hipGraph replay of {N} launches {b:6.2f} us/frame "
          f"({px / b / 1e3:6.0f} Gpix/s) | hipGraph with 4 parallel chains {d4:6.2f} us/frame ({px / d4 / 1e3:6.0f} Gpix/s) | one batched dispatch {c:6.2f} us/frame ({px / c / 1e3:6.0f} Gpix/s)", flush=True)
