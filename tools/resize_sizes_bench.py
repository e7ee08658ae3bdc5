import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoprocessingframework_amd import capi
dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
if len(sys.argv) > 1:
    capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, int(sys.argv[1]))
for (sw, sh, dw, dh) in ((3840,2160,1920,1080),(3840,2160,960,540),(3840,2160,2560,1440),(1920,1080,960,540),(1920,1080,416,416),(3840,2160,640,360)):
    sp, dp = (3*sw+255)//256*256, (3*dw+255)//256*256
    N=8
    src=[torch.randint(0,256,(sh,sp),dtype=torch.uint8,device=dev) for _ in range(N)]
    dst=[torch.zeros((dh,dp),dtype=torch.uint8,device=dev) for _ in range(N)]
    def step():
        for s,d in zip(src,dst): capi.resize(ex, capi.RGB, capi.INTERP_LINEAR, sw, sh, [(s.data_ptr(),sp)], dw, dh, [(d.data_ptr(),dp)])
    step(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/(10*N)
    rd = 3*sw*sh if sw/dw<2 else 3*sw*2*dh
    print(f"[rs] RGB {sw}x{sh}->{dw}x{dh} bilinear: {us:6.2f} us/frame  ~{(rd+3*dw*dh)/us/1e3:6.0f} GB/s")
