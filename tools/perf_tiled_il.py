"""A/B of the pixel-stage LDS layouts (gngf_set_tiled_interleaved 0 / 1) on the headline shape: per-kernel HIP-event times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops, _lib
from collision_handling_in_instantngp_amd import models as mdl
import bench
dev = torch.device("cuda")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
c = bench.SHAPES[cfg]
xy, target, _b = bench.make_batch(cfg, 2**20, 0, dev)
n_host = [int(v) for v in mdl.level_resolutions(c["n_min"], c["n_max"], c["L"])]
n_ls = torch.tensor(n_host, dtype=torch.int32, device=dev)
tables = (torch.rand((c["L"], c["T"], c["F"]), device=dev) - 0.5) * 2e-4
genc = torch.randn((2**20, c["L"] * c["F"]), device=dev)
for variant, bias in ((0, 0), (1, 0), (0, 0), (1, 0)):
    ops.TILED_TILE_SHIFT_BIAS = bias
    _lib.query("gngf_set_tiled_interleaved", variant)
    ks = ops.encode_kernels(xy, n_ls, n_host, tables, None, None, 0, genc)
    out = []
    for name in [k for k in ("encode_fwd:tiled", "encode_bwd:tiled", "encode_bwd:tiled+dG64", "prepare", "vertex_bwd", "encode_fwd:direct", "encode_bwd:direct") if k in ks]:
        fn = ks[name]
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1)/20*1e3:7.1f} us")
    print(f"interleaved={variant} shift_bias={bias}  " + "   ".join(out), flush=True)
    if variant:
        import ctypes
        torch.cuda.synchronize()
        st = (ctypes.c_uint64 * 8)()
        _lib.call("gngf_debug_il_stamps", st)
        v = list(st)
        if v[6]:
            names = ["tail wait", "setup", "clear+bound", "main(thread0)", "wait others", "store pass"]
            print("   bwd wg0 per item [cycles]: " + ", ".join(f"{n_} {v[k] / v[6]:.0f}" for k, n_ in enumerate(names)) + f"; items {v[6]}, px/item {v[7] / v[6]:.0f}")
