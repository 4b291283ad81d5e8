"""Whole-step timing vs ops.TILED_CHUNK (pixels per work item of the pixel stage)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from collision_handling_in_instantngp_amd import ops
dev = torch.device("cuda")
xy, target, _ = bench.strawberry_batch(2**20, 0, dev)
for chunk in (1024, 2048, 4096, 8192, None, 2048, 4096):
    ops.TILED_CHUNK = chunk
    net, models = bench.build_model("gngf_frozen", dev)
    step = bench.graphed(bench.make_step(net, models, "gngf_frozen", xy, target, 1))
    for _ in range(100): step()                          # steady clock
    dt = bench.timed(step, 200, 5, 1)
    print(f"chunk {chunk!s:>5s}  {dt / 200 * 1e3:.4f} ms/step  {2**20 * 200 / dt / 1e6:.1f} Mpixel/s")
    del net, step
    torch.cuda.empty_cache()
