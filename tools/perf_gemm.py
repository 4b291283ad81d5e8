"""Times the three large HPD GEMMs of learning mode (n=2048 rows, T=2^19, H=128) with HIP events.
   logits = h W^T (gngf_linear_fwd), dW = dz^T h (gngf_linear_bwd_weight), dh = dz W (gngf_gemm_acc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops

n, T, H = 2048, 1 << 19, 128
dev = "cuda:0"
h = torch.randn(n, H, device=dev)
W = torch.randn(T, H, device=dev) * 0.05
b = torch.zeros(T, device=dev)
z = torch.empty(n, T, device=dev)
dW = torch.zeros(T, H, device=dev)
g = torch.zeros(n, H, device=dev)


def timeit(name, fn, flop, reps=6):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    print(f"{name:8s} median {ts[reps // 2]:.3f} ms  min {ts[0]:.3f} ms   {flop / ts[reps // 2] / 1e9:.1f} TFLOP/s")


flop = 2.0 * n * T * H
timeit("logits", lambda: ops.call("gngf_linear_fwd", ops.ptr(h), ops.ptr(W), ops.ptr(b), ops.ptr(z), n, T, H, ops.ACT_NONE, ops.stream_ptr()), flop)
timeit("dW", lambda: ops.linear_bwd_weight(z, None, h, dW, None, ops.ACT_NONE), flop)
timeit("dh", lambda: ops.gemm_acc(z, W, g, n, H, T, ta=False, tb=False), flop)
ref = (z[:64, :4096].double() @ W[:4096].double())
g.zero_(); ops.gemm_acc(z[:, :4096].contiguous(), W[:4096], g, n, H, 4096, ta=False, tb=False)
print("dh check", float((g[:64].double() - ref).abs().max() / ref.abs().max()))
for Hx in (32, 64, 256):
    hx = torch.randn(n, Hx, device=dev); Wx = torch.randn(T, Hx, device=dev)
    timeit(f"logitsK{Hx}", lambda: ops.call("gngf_linear_fwd", ops.ptr(hx), ops.ptr(Wx), ops.ptr(b), ops.ptr(z), n, T, Hx, ops.ACT_NONE, ops.stream_ptr()), 2.0 * n * T * Hx)
