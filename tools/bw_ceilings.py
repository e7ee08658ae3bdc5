"""Reference points for the HBM ceilings on this box, measured with library kernels (torch fill_/copy_, i.e. rocclr / ATen
kernels) on buffers far beyond the Infinity Cache: write-only, read+write copy, read-only (sum)."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev)
b = torch.empty(n, dtype=torch.uint8, device=dev)
a32, b32 = a.view(torch.int32), b.view(torch.int32)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


print(f"[ceil] torch fill_ (write 1 GiB)      : {n / t(lambda: a32.fill_(7)) / 1e9:7.0f} GB/s")
print(f"[ceil] torch copy_ (read+write 2 GiB) : {2 * n / t(lambda: b32.copy_(a32)) / 1e9:7.0f} GB/s")
print(f"[ceil] torch sum   (read 1 GiB)       : {n / t(lambda: a32.sum()) / 1e9:7.0f} GB/s")
print(f"[ceil] hipMemsetAsync via zero_       : {n / t(lambda: a.zero_()) / 1e9:7.0f} GB/s")
