"""fixed cost of the decoder kernels: time against the number of 128-pixel tiles per workgroup"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr, query
dev = torch.device("cuda")
in_dim, out_dim = 32, 3
Ws = [torch.randn((64, in_dim), device=dev) / 6, torch.randn(64, device=dev) * 0.1, torch.randn((64, 64), device=dev) / 8, torch.randn(64, device=dev) * 0.1,
      torch.randn((out_dim, 64), device=dev) / 8, torch.randn(out_dim, device=dev) * 0.1]
for split in (0, 1):
    query("gngf_set_decoder_split_bf16", split)
    for P in (2**16, 2**17, 2**18, 2**19, 2**20, 2**21):
        enc = torch.randn((P, in_dim), device=dev) * 0.5
        rgb = torch.empty((P, out_dim), device=dev)
        fn = lambda: call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(None), P, in_dim, out_dim, 0, stream_ptr())
        for _ in range(200): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"split={split} P=2^{P.bit_length()-1}  {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us")
query("gngf_set_decoder_split_bf16", 0)
