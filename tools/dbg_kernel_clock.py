"""decoder kernels launched back to back (dense, steady clocks) vs inside eagerly launched steps (gaps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda")
xy, target, _ = bench.strawberry_batch(2 ** 20, 0, dev)
net, models = bench.build_model("gngf_frozen", dev)
step = bench.make_step(net, models, "gngf_frozen", xy, target, 1)
replay = bench.graphed(step)
for _ in range(100): replay()
kt = bench.kernel_times(net, models, "gngf_frozen", xy, n=50)
print("standalone dense:", {k: round(v * 1e6, 1) for k, v in kt.items()})
for _ in range(100): replay()
ks = bench.kernel_times_in_step(bench.make_step(net, models, "gngf_frozen", xy, target, 1), n=20, warm=3)
print("in eager steps  :", {k: round(v * 1e6, 1) for k, v in ks.items()})
