"""What would the pixel stage gain if enc / d enc rows were laid out in tile (binned) order?  Runs the tiled kernels on the
same pixels given in random order and given pre-sorted by tile (rows then stream instead of scattering 128-byte pieces)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops
from collision_handling_in_instantngp_amd import models as orc   # level_resolutions only (the CPU oracle is for tests)
import bench
dev = torch.device("cuda")
xy, target, _ = bench.strawberry_batch(2**20, 0, dev)
n_host = [int(v) for v in orc.level_resolutions(16, 512, 16)]
n_ls = torch.tensor(n_host, dtype=torch.int32, device=dev)
tables = (torch.rand((16, 2**19, 2), device=dev) - 0.5) * 2e-4
genc = torch.randn((2**20, 32), device=dev)
plan = ops.EncodePlan(2**20, n_host, 2)
ws = ops.TiledWorkspace(plan, xy)
xy_sorted = ws.sorted[:, :2].contiguous()
# third ordering: row-major order of the finest cells (binning keeps it inside each tile up to a block's local shuffle):
# adjacent lanes then touch adjacent LDS words — what would a within-tile sort buy the LDS-bound pixel stage?
key = (xy[:, 1] * 512).floor().long() * 512 + (xy[:, 0] * 512).floor().long()
xy_cell = xy[torch.argsort(key, stable=True)].contiguous()
for name, pts in (("random order", xy), ("tile order", xy_sorted), ("cell order", xy_cell)):
    ks = ops.encode_kernels(pts, n_ls, n_host, tables, None, None, 0, genc)
    for k in ("encode_fwd:tiled", "encode_bwd:tiled"):
        fn = ks[k]
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:14s} {k:18s} {e0.elapsed_time(e1)/10*1e3:8.1f} us")
