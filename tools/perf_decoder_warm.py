"""Does the decoder kernel time depend on how long the GPU has been busy (clock ramp)?  Timings of successive batches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import _lib
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda")
P, in_dim, out_dim = 2 ** 20, 32, 3
Ws = [torch.randn((64, in_dim), device=dev) / 6, torch.zeros(64, device=dev), torch.randn((64, 64), device=dev) / 8, torch.zeros(64, device=dev),
      torch.randn((out_dim, 64), device=dev) / 8, torch.zeros(out_dim, device=dev)]
enc = torch.randn((P, in_dim), device=dev); rgb = torch.empty((P, out_dim), device=dev); drgb = torch.randn((P, out_dim), device=dev)
denc = torch.empty_like(enc); grads = [torch.empty_like(w) for w in Ws]
slabs = torch.empty((_lib.query("gngf_decoder_bwd_slabs", P) * _lib.query("gngf_decoder_slab_floats", in_dim, out_dim),), device=dev)
hidden = torch.empty((_lib.query("gngf_decoder_hidden_floats", P),), device=dev) if os.environ.get("GNGF_RECOMPUTE", "0") != "1" else None
def fwd(): call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
def bwd(): call("gngf_decoder_bwd", ptr(enc), ptr(rgb), ptr(drgb), ptr(None), ptr(None), ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(Ws[3]), ptr(Ws[4]), ptr(denc), *[ptr(g) for g in grads], ptr(slabs), ptr(None), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
def batch(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fwd(); bwd(); torch.cuda.synchronize()
for rnd in range(6):
    print(f"round {rnd}: fwd {batch(fwd, 10):7.1f} us   bwd {batch(bwd, 10):7.1f} us")
print("after 0.5 s idle:"); time.sleep(0.5)
print(f"          fwd {batch(fwd, 10):7.1f} us   bwd {batch(bwd, 10):7.1f} us")
x = torch.empty(2 ** 28, device=dev)
for rnd in range(3):
    for _ in range(20): x.add_(1.0)      # HBM-bound filler
    print(f"after HBM-bound filler: fwd {batch(fwd, 10):7.1f} us   bwd {batch(bwd, 10):7.1f} us")
enc2 = torch.randn((P, in_dim), device=dev) * 1e-3
enc.copy_(enc2)
print(f"small-magnitude enc:    fwd {batch(fwd, 10):7.1f} us   bwd {batch(bwd, 10):7.1f} us")
enc.zero_(); drgb.zero_()
print(f"all-zero enc/drgb:      fwd {batch(fwd, 10):7.1f} us   bwd {batch(bwd, 10):7.1f} us")
