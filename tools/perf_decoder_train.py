"""gngf_decoder_train (forward + MSE gradient + backward in one launch) next to decoder_fwd + decoder_bwd; with a -DGNGF_STAMPS
library (GNGF_LIB_PATH=...stamps.so) also the cycle shares of its phases."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import _lib
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr, query
dev = torch.device("cuda")
P, in_dim, out_dim = 2**20, 32, 3
torch.manual_seed(0)
enc = torch.randn((P, in_dim), device=dev) * 0.5
Ws = [torch.randn((64, in_dim), device=dev) / 6, torch.randn(64, device=dev) * 0.1, torch.randn((64, 64), device=dev) / 8, torch.randn(64, device=dev) * 0.1,
      torch.randn((out_dim, 64), device=dev) / 8, torch.randn(out_dim, device=dev) * 0.1]
target = torch.rand((P, out_dim), device=dev)
one = torch.ones((), device=dev)
slabs = torch.empty((query("gngf_decoder_bwd_slabs", P) * query("gngf_decoder_slab_floats", in_dim, out_dim),), device=dev)
hidden = torch.empty((query("gngf_decoder_hidden_floats", P),), device=dev)
rgb = torch.empty((P, out_dim), device=dev); denc = torch.empty_like(enc)
def two():
    call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
    call("gngf_decoder_bwd", ptr(enc), ptr(rgb), ptr(None), ptr(target), ptr(one), ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(Ws[3]), ptr(Ws[4]), ptr(denc), *[ptr(None)] * 6, ptr(slabs), ptr(None), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
def fused():
    call("gngf_decoder_train", ptr(enc), ptr(target), ptr(one), *[ptr(w) for w in Ws], ptr(rgb), ptr(denc), *[ptr(None)] * 6, ptr(slabs), ptr(None), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
zbuf = torch.empty((16 * 2 ** 19 * 2 + 4 * 720000,), device=dev)       # the step's [table gradient | fixed-point vertex grid]: 75 MiB
zsmall = torch.empty((4 * 720000,), device=dev)                       # the fixed-point vertex grid alone: 11 MiB
def fused_clear(z):
    def fn():
        call("gngf_decoder_train", ptr(enc), ptr(target), ptr(one), *[ptr(w) for w in Ws], ptr(rgb), ptr(denc), *[ptr(None)] * 6, ptr(slabs), ptr(None), ptr(z), z.numel(), P, in_dim, out_dim, 0, stream_ptr())
    return fn
for name, fn in (("decoder_fwd + decoder_bwd", two), ("decoder_train", fused), ("decoder_train + 75 MiB clear", fused_clear(zbuf)),
                 ("decoder_train + 11 MiB clear", fused_clear(zsmall)), ("decoder_train", fused)):
    for _ in range(60): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us")
if os.environ.get("GNGF_LIB_PATH", "").endswith("stamps.so"):
    buf = (ctypes.c_uint64 * 16)()
    _lib.load().gngf_debug_read_stamps.argtypes = [ctypes.c_void_p]
    _lib.load().gngf_debug_read_stamps(buf)
    names = ["load/copy", "forward (rest: act2, L3, sigmoid, d rgb)", "(dbg) fwd: bias, x split, L1, act1", "dh2+dact", "img+dW1", "dh1+dact", "img+dW0", "dX+store", "(dbg) fwd: split of chunk 0", "(dbg) fwd: L2 + h1 image"]
    tot = sum(buf[:10])
    for n, v in zip(names, buf[:10]):
        print(f"  {n:28s} {v/32:9.0f} cycles/tile  {100*v/max(tot,1):5.1f}%")
    print("  total per tile", tot / 32)
