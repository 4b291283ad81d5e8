"""Binning time vs number of binning workgroups (NB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops
from collision_handling_in_instantngp_amd import models as orc   # level_resolutions only (the CPU oracle is for tests)
import bench
dev = torch.device("cuda")
xy, target, _ = bench.strawberry_batch(2**20, 0, dev)
n_host = [int(v) for v in orc.level_resolutions(16, 512, 16)]
plan = ops.EncodePlan(2**20, n_host, 2)
for NB in (64, 128, 256, 384, 512):
    plan.NB = NB
    for _ in range(3): ws = ops.TiledWorkspace(plan, xy)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ws = ops.TiledWorkspace(plan, xy)
    e1.record(); torch.cuda.synchronize()
    print(f"NB {NB:4d}  {e0.elapsed_time(e1)/10*1e3:8.1f} us per TiledWorkspace (allocations + bin_pixels)")
