import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import _lib
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr, query
dev = torch.device("cuda")
P, in_dim, out_dim = 2**20, 32, 3
Ws = [torch.randn((64, in_dim), device=dev) / 6, torch.randn(64, device=dev) * 0.1, torch.randn((64, 64), device=dev) / 8, torch.randn(64, device=dev) * 0.1,
      torch.randn((out_dim, 64), device=dev) / 8, torch.randn(out_dim, device=dev) * 0.1]
enc = torch.randn((P, in_dim), device=dev) * 0.5
rgb = torch.rand((P, out_dim), device=dev); drgb = torch.randn((P, out_dim), device=dev) * 1e-6; denc = torch.empty_like(enc)
slabs = torch.empty((query("gngf_decoder_bwd_slabs", P) * query("gngf_decoder_slab_floats", in_dim, out_dim),), device=dev)
query("gngf_set_decoder_split_bf16", 1)
fn = lambda: call("gngf_decoder_bwd", ptr(enc), ptr(rgb), ptr(drgb), ptr(None), ptr(None), ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(Ws[3]), ptr(Ws[4]), ptr(denc), *[ptr(None)] * 6, ptr(slabs), ptr(None), ptr(None), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
for _ in range(60): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"kernel {us:.1f} us")
buf = (ctypes.c_uint64 * 16)()
_lib.load().gngf_debug_read_stamps.argtypes = [ctypes.c_void_p]
print("rc", _lib.load().gngf_debug_read_stamps(buf))
names = ["loop top / addresses", "L1 (24)", "L2 + h1 image (48)", "dz3, h2 img, dh2, dW2", "dh1 + dz2 image (48)", "dW1 + db1 (60)", "d enc + dz1, x images (24)", "dW0 + db0 (36)", "out + copies"]
ntile = P // 128 // 256
tot = sum(buf[:len(names)])
for n, v in zip(names, buf[:len(names)]):
    print(f"  {n:28s} {v/ntile:9.0f} ticks/tile  {100*v/max(tot,1):5.1f}%")
print("  total per tile", tot / ntile, f" -> {tot / us / 1e3:.2f} GHz if the loop were the whole kernel")
