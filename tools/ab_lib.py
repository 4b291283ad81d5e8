"""Whole-step timing of the library selected with GNGF_LIB_PATH (run twice with different builds for an A/B)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda")
xy, target, _ = bench.strawberry_batch(2**20, 0, dev)
for rep in range(3):
    net, models = bench.build_model("gngf_frozen", dev)
    step = bench.graphed(bench.make_step(net, models, "gngf_frozen", xy, target, 1))
    dt = bench.timed(step, 40, 5, 1)
    print(f"{os.environ.get('GNGF_LIB_PATH', 'default'):40s} {dt / 40 * 1e3:.4f} ms/step")
    del net, step
    torch.cuda.empty_cache()
