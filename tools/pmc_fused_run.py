import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench"]
import bench
dev = torch.device("cuda", 0)
wl = bench.Workload("fused_4k_720p", dev, 32, 0, "batch")
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
wl2 = bench.Workload("resize_4k_720p", dev, 8, 0, "single")
for _ in range(3):
    wl2.step()
torch.cuda.synchronize()
