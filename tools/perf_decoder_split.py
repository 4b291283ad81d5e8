"""Split-bf16 decoder kernels next to the fp32-MFMA ones: time, and error of both against a float64 evaluation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import _lib
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr, query
dev = torch.device("cuda")
P, in_dim, out_dim = 2**20, int(os.environ.get("IN_DIM", 32)), 3
torch.manual_seed(0)
enc = torch.randn((P, in_dim), device=dev) * 0.5
Ws = [torch.randn((64, in_dim), device=dev) / 6, torch.randn(64, device=dev) * 0.1, torch.randn((64, 64), device=dev) / 8, torch.randn(64, device=dev) * 0.1,
      torch.randn((out_dim, 64), device=dev) / 8, torch.randn(out_dim, device=dev) * 0.1]
drgb = torch.randn((P, out_dim), device=dev) * 1e-6
slabs = torch.empty((query("gngf_decoder_bwd_slabs", P) * query("gngf_decoder_slab_floats", in_dim, out_dim),), device=dev)
hidden = torch.empty((query("gngf_decoder_hidden_floats", P),), device=dev)
def run(split, with_bwd):
    query("gngf_set_decoder_split_bf16", 1 if split else 0)
    rgb = torch.empty((P, out_dim), device=dev); denc = torch.empty_like(enc); grads = [torch.empty_like(w) for w in Ws]
    hid = None if split else hidden
    def fwd(): call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(hid), P, in_dim, out_dim, 0, stream_ptr())
    def bwd(): call("gngf_decoder_bwd", ptr(enc), ptr(rgb), ptr(drgb), ptr(None), ptr(None), ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(Ws[3]), ptr(Ws[4]), ptr(denc), *[ptr(g) for g in grads], ptr(slabs), ptr(None), ptr(hid), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
    t = {}
    for name, fn in (("fwd", fwd),) + ((("bwd", bwd),) if with_bwd else ()):
        for _ in range(30): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        t[name] = e0.elapsed_time(e1) / 20 * 1e3
    query("gngf_set_decoder_split_bf16", 0)
    return t, rgb, denc, grads
# float64 evaluation on a slice (autograd)
n = 2**16
x64 = enc[:n].double().requires_grad_(True)
W64 = [w.double().requires_grad_(True) for w in Ws]
h = torch.relu(x64 @ W64[0].T + W64[1]); h = torch.relu(h @ W64[2].T + W64[3]); y64 = torch.sigmoid(h @ W64[4].T + W64[5])
with_bwd = os.environ.get("BWD", "1") == "1"
res = {}
for split in (False, True):
    t, rgb, denc, grads = run(split, with_bwd)
    err = float((rgb[:n].double() - y64).abs().max())
    line = f"{'split-bf16' if split else 'fp32 MFMA '}  fwd {t['fwd']:7.1f} us  max |rgb - f64| {err:.2e}"
    if with_bwd:
        (gx,) = torch.autograd.grad(y64, x64, drgb[:n].double(), retain_graph=True)
        line += f"   bwd {t['bwd']:7.1f} us  max |denc - f64| / max|denc| {float((denc[:n].double() - gx).abs().max() / gx.abs().max()):.2e}"
    res[split] = (rgb, denc, grads)
    print(line)
print("fwd fp32 vs split max |diff|", float((res[False][0] - res[True][0]).abs().max()))
if with_bwd:
    for k, nm in enumerate(("dW0", "db0", "dW1", "db1", "dW2", "db2")):
        a, b = res[False][2][k], res[True][2][k]
        print(f"  {nm}: max |fp32 - split| / max |fp32| = {float((a - b).abs().max() / a.abs().max()):.2e}")
    a, b = res[False][1], res[True][1]
    print(f"  denc: max |fp32 - split| / max = {float((a - b).abs().max() / a.abs().max()):.2e}")
