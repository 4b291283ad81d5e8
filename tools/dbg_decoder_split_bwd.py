import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr, query
dev = torch.device("cuda")
P, in_dim, out_dim = int(os.environ.get("P", 2**20)), 32, 3
torch.manual_seed(0)
enc = torch.randn((P, in_dim), device=dev) * 0.5
Ws = [torch.randn((64, in_dim), device=dev) / 6, torch.randn(64, device=dev) * 0.1, torch.randn((64, 64), device=dev) / 8, torch.randn(64, device=dev) * 0.1,
      torch.randn((out_dim, 64), device=dev) / 8, torch.randn(out_dim, device=dev) * 0.1]
drgb = torch.randn((P, out_dim), device=dev) * 1e-6
slabs = torch.empty((query("gngf_decoder_bwd_slabs", P) * query("gngf_decoder_slab_floats", in_dim, out_dim),), device=dev)
def run(split):
    query("gngf_set_decoder_split_bf16", split)
    rgb = torch.empty((P, out_dim), device=dev); denc = torch.full_like(enc, float("nan")); grads = [torch.empty_like(w) for w in Ws]
    call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(None), P, in_dim, out_dim, 0, stream_ptr())
    call("gngf_decoder_bwd", ptr(enc), ptr(rgb), ptr(drgb), ptr(None), ptr(None), ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(Ws[3]), ptr(Ws[4]), ptr(denc), *[ptr(g) for g in grads], ptr(slabs), ptr(None), ptr(None), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
    torch.cuda.synchronize()
    query("gngf_set_decoder_split_bf16", 0)
    return denc, grads
a, ga = run(0)
b, gb = run(1)
d = (a - b).abs().max(1).values / a.abs().max()
bad = (d > 1e-5) | torch.isnan(d)
print("bad pixels", int(bad.sum()), "of", P)
idx = bad.nonzero().flatten()
if idx.numel():
    tiles = torch.unique(idx // 128)
    print("bad 128-pixel tiles:", tiles[:20].tolist(), "... count", tiles.numel(), " mod 256:", torch.unique(tiles % 256)[:20].tolist())
    print("first bad pixels:", idx[:20].tolist())
    p = int(idx[0]); print("fp32", a[p, :8].tolist()); print("split", b[p, :8].tolist())
for k, nm in enumerate(("dW0", "db0", "dW1", "db1", "dW2", "db2")):
    print(nm, float((ga[k] - gb[k]).abs().max() / ga[k].abs().max()))
