"""vertex_bwd (GNGF, slot-ordered) timing on the bench's frozen-HPD table, plus run-length statistics of the slot order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from collision_handling_in_instantngp_amd import ops
dev = torch.device("cuda")
xy, target, _ = bench.strawberry_batch(2**20, 0, dev)
net, models = bench.build_model("gngf_frozen", dev)
kt = bench.kernel_times(net, models, "gngf_frozen", xy, n=20)
print({k: round(v * 1e6, 1) for k, v in kt.items()}, "us")
with torch.no_grad():
    _tv, ti, w, vstride, NV, order = net._frozen_vertex_table(0)
flat = ti.reshape(-1)[order.long()]
runs = int((flat[1:] != flat[:-1]).sum().item()) + 1
print("entries", flat.numel(), "distinct slots", int(torch.unique(flat).numel()), "runs in visiting order", runs)
