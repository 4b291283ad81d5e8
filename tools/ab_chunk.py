"""Whole-step timing vs ops.TILED_CHUNK (pixels per work item of the pixel stage)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from collision_handling_in_instantngp_amd import ops, train
dev = torch.device("cuda")
xy, target, bounds = bench.make_batch("cfg2", 2**20, 0, dev)
for chunk in (1024, 2048, 4096, 8192, None, 2048, 4096):
    ops.TILED_CHUNK = chunk
    net, models = bench.build_model("gngf_frozen", dev, bounds)
    gs = train.GraphedStep(net, train.Loss(delta=1, gamma=-2, epsilon=1), None, 1, 1, 1e-3, unroll=4)
    gs.run_many([(xy, target)] * 4)
    step = gs.replay_only
    for _ in range(30): step()                          # steady clock
    dt = bench.timed(step, 50, 2, 1)
    print(f"chunk {chunk!s:>5s}  {dt / 200 * 1e3:.4f} ms/step  {2**20 * 200 / dt / 1e6:.1f} Mpixel/s")
    del net, step, gs
    torch.cuda.empty_cache()
ops.TILED_CHUNK = None
