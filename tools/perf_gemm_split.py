"""split-bf16 GEMM vs exact fp32-MFMA GEMM: error against float64 and time, on the three shapes of the HPD's last layer"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops, _lib
dev = torch.device("cuda")
n, T, H = 2048, 2 ** 19, 128
g = torch.Generator(device=dev).manual_seed(1)
h = torch.randn((n, H), device=dev, generator=g) * 50
W = (torch.rand((T, H), device=dev, generator=g) * 2 - 1) / H ** 0.5
b = (torch.rand((T,), device=dev, generator=g) * 2 - 1) / H ** 0.5
dz = torch.randn((n, T), device=dev, generator=g) * 1e-3
def timeit(fn, reps=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {}
MODES = (0, 17, 1, 2)      # exact fp32 MFMA | per-wave three-way split | three bf16 planes in LDS | two planes (accumulating GEMMs)
for split in MODES + MODES:
    _lib.query("gngf_set_gemm_split_bf16", split)
    z = ops.linear_fwd(h, W, b, ops.ACT_NONE)
    t1 = timeit(lambda: ops.linear_fwd(h, W, b, ops.ACT_NONE))
    dW = torch.zeros_like(W)
    ops.linear_bwd_weight(dz, None, h, dW, None, ops.ACT_NONE)
    def f2():
        ops.linear_bwd_weight(dz, None, h, dW, None, ops.ACT_NONE)
    t2 = timeit(f2)
    dW.zero_(); ops.linear_bwd_weight(dz, None, h, dW, None, ops.ACT_NONE)
    dh = torch.zeros((n, H), device=dev)
    ops.gemm_acc(dz, W, dh, n, H, T, ta=False, tb=False)
    def f3():
        ops.gemm_acc(dz, W, dh, n, H, T, ta=False, tb=False)
    t3 = timeit(f3)
    dh.zero_(); ops.gemm_acc(dz, W, dh, n, H, T, ta=False, tb=False)
    res[split] = (z, dW, dh)
    fl = 2.0 * n * T * H
    print(f"split={split}: logits {t1:.3f} ms ({fl/t1/1e9:.0f} TF)  dW {t2:.3f} ms ({fl/t2/1e9:.0f} TF)  dh {t3:.3f} ms ({fl/t3/1e9:.0f} TF)")
_lib.query("gngf_set_gemm_split_bf16", 0)
# float64 references on a sample of rows / columns
rows = torch.arange(0, n, 64, device=dev)
zr = (h[rows].double() @ W.double().T + b.double())
cols = torch.arange(0, T, 4096, device=dev)
dWr = dz[:, cols].double().T @ h.double()
dhr = dz[rows].double() @ W.double()
scale_z = (h[rows].double().abs() @ W.double().abs().T)          # sum |a_k b_k|
for split in MODES:
    z, dW, dh = res[split]
    ez = (z[rows].double() - zr).abs()
    sW = dz[:, cols].double().abs().T @ h.double().abs()
    sH = dz[rows].double().abs() @ W.double().abs()
    print(f"split={split}: logits max abs err {ez.max():.3e} (max |z| {zr.abs().max():.1f}), err / sum|ab| {float((ez / scale_z).max()):.3e};"
          f"  dW max err / max {float((dW[cols].double() - dWr).abs().max() / dWr.abs().max()):.3e}, err / sum|ab| {float(((dW[cols].double() - dWr).abs() / sW).max()):.3e};"
          f"  dh max err / max {float((dh[rows].double() - dhr).abs().max() / dhr.abs().max()):.3e}, err / sum|ab| {float(((dh[rows].double() - dhr).abs() / sH).max()):.3e}")
z1, z17 = res[1][0], res[17][0]
print("three planes vs per-wave split, logits bit-identical:", bool(torch.equal(z1, z17)),
      " dW:", float((res[1][1] - res[17][1]).abs().max()), " dh:", float((res[1][2] - res[17][2]).abs().max()))
