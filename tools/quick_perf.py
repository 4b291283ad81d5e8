"""Scratch perf probe for individual kernels (not the contract bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from collision_handling_in_instantngp_amd import ops
from collision_handling_in_instantngp_amd import models as orc   # level_resolutions only (the CPU oracle is for tests)

def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

P, L, F, T, K = 2**20, 16, 2, 2**19, 4
dev = "cuda"
n_ls = torch.tensor(orc.level_resolutions(16, 512, L), dtype=torch.int32, device=dev)
# strawberry-like coords: rows in [0,1], cols in [0,0.667], shuffled
g = torch.Generator(device=dev).manual_seed(0)
xy = torch.rand((P, 2), device=dev, generator=g); xy[:, 1] *= 338 / 507
tables = (torch.rand((L, T, F), device=dev, generator=g) - 0.5) * 2e-4
genc = torch.randn((P, L * F), device=dev, generator=g)
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr
enc = torch.empty((P, L * F), device=dev)
dt = torch.zeros_like(tables)
def fwd_hash(): call("gngf_encode_fwd", ptr(xy), ptr(tables), 0, ptr(None), ptr(None), ptr(n_ls), ptr(enc), P, L, F, T, 0, 0, 0, 0, 0, L, ptr(None), stream_ptr())
def bwd_hash(): call("gngf_encode_bwd", ptr(xy), ptr(tables), 0, ptr(None), ptr(None), ptr(n_ls), ptr(genc), ptr(dt), ptr(None), P, L, F, T, 0, 0, 0, 0, 0, L, ptr(None), stream_ptr())
t = timeit(fwd_hash); print(f"direct hash fwd  {t:.3f} ms  {P/t/1e3:.1f} Mpx/s")
t = timeit(bwd_hash); print(f"direct hash bwd  {t:.3f} ms  {P/t/1e3:.1f} Mpx/s")
vs = 514; NV = vs * vs
vidx = torch.randint(0, T, (NV, K), device=dev, dtype=torch.int32, generator=g)
vw = torch.rand((NV, K), device=dev, generator=g)
dvw = torch.zeros_like(vw)
def fwd_vt(): call("gngf_encode_fwd", ptr(xy), ptr(tables), 0, ptr(vidx), ptr(vw), ptr(n_ls), ptr(enc), P, L, F, T, K, 1, vs, NV, 0, L, ptr(None), stream_ptr())
def bwd_vt(): call("gngf_encode_bwd", ptr(xy), ptr(tables), 0, ptr(vidx), ptr(vw), ptr(n_ls), ptr(genc), ptr(dt), ptr(dvw), P, L, F, T, K, 1, vs, NV, 0, L, ptr(None), stream_ptr())
t = timeit(fwd_vt); print(f"direct VT fwd    {t:.3f} ms  {P/t/1e3:.1f} Mpx/s")
t = timeit(bwd_vt); print(f"direct VT bwd    {t:.3f} ms  {P/t/1e3:.1f} Mpx/s")
t = timeit(lambda: dt.zero_()); print(f"zero 64MiB       {t:.3f} ms")
a = torch.empty(2**28, device=dev); b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a)); print(f"copy 1GiB->1GiB  {t:.3f} ms  {2*a.numel()*4/t/1e9:.2f} TB/s")
