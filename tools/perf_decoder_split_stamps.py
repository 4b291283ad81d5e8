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
rgb = torch.empty((P, out_dim), device=dev)
query("gngf_set_decoder_split_bf16", 1)
for _ in range(100): call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(None), P, in_dim, out_dim, 0, stream_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(None), P, in_dim, out_dim, 0, stream_ptr())
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"kernel {us:.1f} us")
buf = (ctypes.c_uint64 * 16)()
_lib.load().gngf_debug_read_fwd_stamps.argtypes = [ctypes.c_void_p]
print("rc", _lib.load().gngf_debug_read_fwd_stamps(buf))
names = sys.argv[1:] or ["loop top", "L1", "L2", "act2 + L3", "sigmoid + store", "x copy (wait for next x)"]
ntile = P // 128 // 512
tot = sum(buf[:len(names)])
for n, v in zip(names, buf[:len(names)]):
    print(f"  {n:28s} {v/ntile:9.0f} ticks/tile  {100*v/max(tot,1):5.1f}%")
print("  total per tile", tot / ntile, f" loop ticks {tot}  -> if the loop were the whole kernel: {tot / us / 1e3:.2f} GHz")
