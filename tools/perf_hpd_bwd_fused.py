"""Backward of the HPD's last layer on one 4096-row chunk at T = 2^19: the three separate entry points (softmax backward in place,
dW, dh) against gngf_hpd_bwd_dot + gngf_hpd_bwd_fused (dz formed in the GEMM loaders), two and three planes; kernel times by HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops, _lib
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr, query
dev = torch.device("cuda")
U, T, H, L, K = 4096, 2 ** 19, 128, 16, 4
g = torch.Generator(device=dev).manual_seed(1)
h = torch.relu(torch.randn((U, H), device=dev, generator=g) * 30)
W = (torch.rand((T, H), device=dev, generator=g) * 2 - 1) / H ** 0.5
z = torch.randn((U, T), device=dev, generator=g) * 4
mw = torch.rand((U, L), device=dev, generator=g) / (4 * U)
G = torch.randn((L, T), device=dev, generator=g) * 3
m = z.max(dim=1, keepdim=True).values
s = torch.exp(z - m).sum(dim=1, keepdim=True)
rowstat = torch.cat([m, s], dim=1).contiguous()
tp, ti = torch.topk(z, K, dim=1)
pk = (torch.exp(tp - m) / s).contiguous()
ti32 = ti.to(torch.int32).contiguous()
dq = torch.randn((U, K), device=dev, generator=g) * 1e-3
def timeit(fn, reps=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
dot = torch.empty((U,), device=dev)
dW, db, dH = torch.zeros((T, H), device=dev), torch.zeros((T,), device=dev), torch.zeros((U, H), device=dev)
scratch = torch.empty((U * (1 + K),), device=dev)
dz = torch.empty_like(z)
def f_dot():
    call("gngf_hpd_bwd_dot", ptr(z), ptr(rowstat), ptr(dq), ptr(pk), ptr(mw), ptr(G), L, ptr(dot), U, T, K, stream_ptr())
preps = {2: ops._HpdBwdPlanes(h, mw, W, G, L, 2), 3: ops._HpdBwdPlanes(h, mw, W, G, L, 3)}
def f_fused(planes):
    preps[planes].fused(z, rowstat, dot, dq, pk, ti32, h, W, dW, db, dH, 0, U, T, K)
def f_lowrank():
    call("gngf_softmax_bwd_lowrank", ptr(dz), ptr(rowstat), ptr(dq), ptr(ti32), ptr(mw), ptr(G), L, ptr(db), ptr(scratch), ptr(pk),
         U, T, K, stream_ptr())
def f_gemms(mode):
    prev = query("gngf_set_gemm_split_bf16", mode)
    ops.linear_bwd_weight(dz, None, h, dW, None, ops.ACT_NONE)
    ops.gemm_acc(dz, W, dH, U, H, T, ta=False, tb=False)
    query("gngf_set_gemm_split_bf16", prev)
for rep in range(2):
    dz.copy_(z)
    t_dot = timeit(f_dot)
    t_prep = timeit(lambda: ops._HpdBwdPlanes(h, mw, W, G, L, 2), reps=3)
    t_f2, t_f3 = timeit(lambda: f_fused(2)), timeit(lambda: f_fused(3))
    t_lr = timeit(f_lowrank, reps=3)           # (in place: after the first call it runs on its own output — same traffic)
    t_g2, t_g1 = timeit(lambda: f_gemms(2)), timeit(lambda: f_gemms(1))
    print(f"prepare (once per backward pass; here for one chunk's rows) {t_prep:.2f} ms | dot {t_dot:.2f} ms | fused dW+dh: two planes {t_f2:.2f}, three {t_f3:.2f} | separate: softmax backward (dot+apply) {t_lr:.2f}, "
          f"dW+dh two planes {t_g2:.2f}, three {t_g1:.2f}  ->  per chunk {t_dot + t_f2:.2f} / {t_dot + t_f3:.2f} vs {t_lr + t_g2:.2f} / {t_lr + t_g1:.2f} ms")
# accuracy at full size against float64 on a sample
dW.zero_(); db.zero_(); dH.zero_(); f_dot(); f_fused(2); torch.cuda.synchronize()
res2 = (dW.clone(), db.clone(), dH.clone())
dW.zero_(); db.zero_(); dH.zero_(); f_fused(3); torch.cuda.synchronize()
res3 = (dW.clone(), db.clone(), dH.clone())
cols = torch.arange(0, T, 2048, device=dev)
p = torch.exp(z.double() - m.double()) / s.double()
gfull = (mw.double() @ G.double()).scatter_add(1, ti, dq.double())
dotr = (p * gfull).sum(dim=1, keepdim=True)
dzr = p * (gfull - dotr)
del p, gfull
dWr, dbr, dHr = dzr[:, cols].T @ h.double(), dzr.sum(dim=0), dzr @ W.double()
for name, (a, b, c) in (("two planes", res2), ("three planes", res3)):
    print(name, "|err| / max:  dW", float((a[cols].double() - dWr).abs().max() / dWr.abs().max()),
          " db", float((b.double() - dbr).abs().max() / dbr.abs().max()), " dH", float((c.double() - dHr).abs().max() / dHr.abs().max()))
