import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import _lib
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda")
torch.manual_seed(0)
for in_dim, P in ((8, 1000), (16, 1000), (32, 1000), (32, 70000), (64, 1000), (24, 3000), (48, 5000), (64, 40001), (2, 300), (32, 1)):
    out_dim = 3
    Ws = [torch.randn((64, in_dim), device=dev) / 4, torch.randn(64, device=dev) / 4, torch.randn((64, 64), device=dev) / 8, torch.randn(64, device=dev) / 4,
          torch.randn((out_dim, 64), device=dev) / 8, torch.randn(out_dim, device=dev) / 4]
    enc = torch.randn((P, in_dim), device=dev); rgb = torch.empty((P, out_dim), device=dev); drgb = torch.randn((P, out_dim), device=dev)
    denc = torch.empty_like(enc); grads = [torch.empty_like(w) for w in Ws]
    slabs = torch.empty((_lib.query("gngf_decoder_bwd_slabs", P) * _lib.query("gngf_decoder_slab_floats", in_dim, out_dim),), device=dev)
    hidden = torch.empty((_lib.query("gngf_decoder_hidden_floats", P),), device=dev) if os.environ.get("GNGF_RECOMPUTE", "0") != "1" else None
    call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
    call("gngf_decoder_bwd", ptr(enc), ptr(rgb), ptr(drgb), ptr(None), ptr(None), ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(Ws[3]), ptr(Ws[4]), ptr(denc), *[ptr(g) for g in grads], ptr(slabs), ptr(None), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
    x = enc.double().requires_grad_(); ps = [w.double().requires_grad_() for w in Ws]
    h1 = torch.relu(x @ ps[0].T + ps[1]); h2 = torch.relu(h1 @ ps[2].T + ps[3]); y = torch.sigmoid(h2 @ ps[4].T + ps[5])
    y.backward(drgb.double())
    def err(a, b): return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
    print(f"in_dim {in_dim:2d} P {P:6d}  y {err(rgb, y.detach()):.1e} dx {err(denc, x.grad):.1e} " + " ".join(f"{n} {err(g, p.grad):.1e}" for n, g, p in zip(("dW0", "db0", "dW1", "db1", "dW2", "db2"), grads, ps)))
    if in_dim == 8:
        d = (grads[0].double() - ps[0].grad).abs()
        print("  dW0 err by row block:", [float(d[r:r+8].max()) for r in range(0, 64, 8)], " by col:", [float(d[:, c].max()) for c in range(in_dim)])
    if in_dim == 32 and P == 1000:
        d = (denc.double() - x.grad).abs()
        print("  dx err by column:", [f"{float(d[:, c].max()):.1e}" for c in range(in_dim)])
        print("  dx err by row block of 32:", [f"{float(d[r:r+32].max()):.1e}" for r in range(0, 256, 32)])
