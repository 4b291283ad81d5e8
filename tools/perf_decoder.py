import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops, _lib
from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda")
_lib.query("gngf_set_decoder_bwd_hybrid", int(os.environ.get("GNGF_HYBRID", "1")))
P, in_dim, out_dim = 2**20, 32, 3
enc = torch.randn((P, in_dim), device=dev)
Ws = [torch.randn((64, in_dim), device=dev) / 6, torch.zeros(64, device=dev), torch.randn((64, 64), device=dev) / 8, torch.zeros(64, device=dev),
      torch.randn((out_dim, 64), device=dev) / 8, torch.zeros(out_dim, device=dev)]
rgb = torch.empty((P, out_dim), device=dev)
drgb = torch.randn((P, out_dim), device=dev)
denc = torch.empty_like(enc)
grads = [torch.empty_like(w) for w in Ws]
slabs = torch.empty((_lib.query("gngf_decoder_bwd_slabs", P) * _lib.query("gngf_decoder_slab_floats", in_dim, out_dim),), device=dev)
hidden = torch.empty((_lib.query("gngf_decoder_hidden_floats", P),), device=dev) if os.environ.get("GNGF_RECOMPUTE", "0") != "1" else None
def fwd(): call("gngf_decoder_fwd", ptr(enc), *[ptr(w) for w in Ws], ptr(rgb), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
def bwd(): call("gngf_decoder_bwd", ptr(enc), ptr(rgb), ptr(drgb), ptr(None), ptr(None), ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(Ws[3]), ptr(Ws[4]), ptr(denc), *[ptr(g) for g in grads], ptr(slabs), ptr(None), ptr(hidden), ptr(None), 0, P, in_dim, out_dim, 0, stream_ptr())
for name, fn in (("decoder_fwd", fwd), ("decoder_bwd", bwd)):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = {"decoder_fwd": 2 * (32 * 64 + 64 * 64 + 64 * 3), "decoder_bwd": 0}[name]
    print(f"{name:14s} {us:8.1f} us  {P/us:8.1f} Mpx/s")
if os.environ.get("GNGF_LIB_PATH", "").endswith("stamps.so"):
    import ctypes
    buf = (ctypes.c_uint64 * 16)()
    _lib.load().gngf_debug_read_stamps.argtypes = [ctypes.c_void_p]
    torch.cuda.synchronize()
    print("rc", _lib.load().gngf_debug_read_stamps(buf))
    names = ["load/copy", "recompute L1+L2", "(hyb) dW2 mf4 block", "dh2+dact", "img+dW1", "dh1+dact (hyb: mask only)", "img+dW0", "dX+store", "(hyb) d1 chunks 0-1", "(hyb) d1 chunks 2-3"]
    tot = sum(buf[:10])
    for n, v in zip(names, buf[:10]):
        print(f"  {n:18s} {v/32:9.0f} cycles/tile  {100*v/max(tot,1):5.1f}%")
    print("  total per tile", tot / 32)
    _lib.load().gngf_debug_read_fwd_stamps.argtypes = [ctypes.c_void_p]
    print("fwd rc", _lib.load().gngf_debug_read_fwd_stamps(buf))
    names = ["loop top", "consume/store/prefetch", "L1 (+act tile 0)", "L2 k<32 (+act tile 1)", "L2 k>=32 (+out tile 0)", "out tile 1 + sigmoid"]
    tot = sum(buf[:6])
    for n, v in zip(names, buf[:6]):
        print(f"  {n:26s} {v/32:9.0f} cycles/tile  {100*v/max(tot,1):5.1f}%")
    print("  total per tile", tot / 32)

    import numpy as np
    bt = (ctypes.c_uint64 * 1024)()
    _lib.load().gngf_debug_read_blocktime.argtypes = [ctypes.c_void_p]
    _lib.load().gngf_debug_read_blocktime(bt)
    a = np.array(bt[:], dtype=np.float64).reshape(2, 256, 2)
    for k, nm in enumerate(("fwd", "bwd")):
        t0, t1 = a[k, :, 0], a[k, :, 1]
        dur = t1 - t0
        print(f"{nm}: workgroup durations (s_memtime ticks) min {dur.min():.0f} mean {dur.mean():.0f} max {dur.max():.0f};  first start -> last end {t1.max() - t0.min():.0f};"
              f"  start spread {t0.max() - t0.min():.0f}")
