"""Fixed (prologue + epilogue + launch) cost of the decoder kernels: time vs pixels per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import _lib
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda")
in_dim, out_dim = 32, 3
Ws = [torch.randn((64, in_dim), device=dev) / 6, torch.zeros(64, device=dev), torch.randn((64, 64), device=dev) / 8, torch.zeros(64, device=dev),
      torch.randn((out_dim, 64), device=dev) / 8, torch.zeros(out_dim, device=dev)]
for P in (2 ** 15, 2 ** 16, 2 ** 17, 2 ** 18, 2 ** 19, 2 ** 20, 2 ** 21):
    enc = torch.randn((P, in_dim), device=dev); rgb = torch.empty((P, out_dim), device=dev); drgb = torch.randn((P, out_dim), device=dev)
    denc = torch.empty_like(enc); grads = [torch.empty_like(w) for w in Ws]
    slabs = torch.empty((_lib.query("gngf_decoder_bwd_slabs", P) * _lib.query("gngf_decoder_slab_floats", in_dim, out_dim),), device=dev)
    hidden = torch.empty((_lib.query("gngf_decoder_hidden_floats", P),), device=dev) if os.environ.get("GNGF_RECOMPUTE", "0") != "1" else None
    def fwd(): call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
    def bwd(): call("gngf_decoder_bwd", ptr(enc), ptr(rgb), ptr(drgb), ptr(None), ptr(None), ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(Ws[3]), ptr(Ws[4]), ptr(denc), *[ptr(g) for g in grads], ptr(slabs), ptr(None), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
    out = []
    for fn in (fwd, bwd):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"P=2^{P.bit_length()-1:2d}  tiles/WG {P/128/256:5.1f}  fwd {out[0]:7.1f} us  bwd {out[1]:7.1f} us")
