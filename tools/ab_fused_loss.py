"""in-step A/B: kernel times inside eagerly launched steps, and the replayed step, with / without the fused pixel loss"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from collision_handling_in_instantngp_amd import train, models as M
dev = torch.device("cuda")
xy, target, bounds = bench.make_batch("cfg2", 2 ** 20, 0, dev)
for fused in (True, False, True, False):
    net, models = bench.build_model("gngf_frozen", dev, bounds)
    if not fused:
        net.fused_mse = lambda t: __import__("contextlib").nullcontext()
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    gs = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3)
    gs(xy, target)
    for _ in range(60): gs.replay_only()
    dt = bench.timed(gs.replay_only, 50, 3, 1)
    kt, _ = bench.kernel_times_in_step(bench.eager_step_fn(net, "gngf_frozen", xy, target, 1))
    print("fused" if fused else "plain", f"replayed step {dt / 50 * 1e3:.4f} ms;", {k: round(v * 1e6, 1) for k, v in kt.items()})
    del net, gs
