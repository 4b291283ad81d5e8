"""Work-item size (ops.TILED_CHUNK) against the pixel-stage kernels of the headline shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops, _lib
from collision_handling_in_instantngp_amd import models as mdl
import bench
dev = torch.device("cuda")
c = bench.SHAPES["cfg2"]
xy, target, _b = bench.make_batch("cfg2", 2**20, 0, dev)
n_host = [int(v) for v in mdl.level_resolutions(c["n_min"], c["n_max"], c["L"])]
n_ls = torch.tensor(n_host, dtype=torch.int32, device=dev)
tables = (torch.rand((c["L"], c["T"], c["F"]), device=dev) - 0.5) * 2e-4
genc = torch.randn((2**20, c["L"] * c["F"]), device=dev)
for chunk in (1024, 1536, 2048, 3072, 4096, 1024, 2048, 4096):
    ops.TILED_CHUNK = chunk
    ks = ops.encode_kernels(xy, n_ls, n_host, tables, None, None, 0, genc)
    out = []
    for name in ("encode_fwd:tiled", "encode_bwd:tiled+dG64", "prepare"):
        fn = ks[name]
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1)/20*1e3:7.1f} us")
    print(f"chunk={chunk}  " + "   ".join(out), flush=True)
