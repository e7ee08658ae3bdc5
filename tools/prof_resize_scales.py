import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from videoprocessingframework_amd import capi
dev = torch.device("cuda", 0)
ex = capi.make_exec(torch.cuda.current_stream().cuda_stream)
capi.set_tuning(capi.TUNE_NV12_RGB_VARIANT, int(sys.argv[1]))
for (sw, sh, dw, dh) in ((1920,1080,1280,720),(3840,2160,2560,1440),(3840,2160,3000,1688),(1280,720,1920,1080),(1920,1080,3840,2160),(3840,2160,2160,1216)):
    sp, dp = (3*sw+255)//256*256, (3*dw+255)//256*256
    src=[torch.randint(0,256,(sh,sp),dtype=torch.uint8,device=dev) for _ in range(4)]
    dst=[torch.zeros((dh,dp),dtype=torch.uint8,device=dev) for _ in range(4)]
    for _ in range(3):
        for s,d in zip(src,dst): capi.resize(ex, capi.RGB, capi.INTERP_LINEAR, sw, sh, [(s.data_ptr(),sp)], dw, dh, [(d.data_ptr(),dp)])
    torch.cuda.synchronize()
