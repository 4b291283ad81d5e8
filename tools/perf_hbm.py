"""Achievable HBM write / read / copy bandwidth with framework streaming kernels (context for the decoder's hidden-layer stores)."""
import torch
dev = torch.device("cuda")
x = torch.empty(2 ** 27, device=dev)      # 512 MiB
y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
gb = x.numel() * 4 / 1e9
print(f"fill 512 MiB   {gb / t(lambda: x.fill_(1.0)):8.0f} GB/s (write)")
print(f"sum  512 MiB   {gb / t(lambda: x.sum()):8.0f} GB/s (read)")
print(f"copy 512 MiB   {2 * gb / t(lambda: y.copy_(x)):8.0f} GB/s (read + write)")
