"""cfg5-shaped single-GPU probe in fp32: synthetic 8192^2 coordinates, L=16, F=4, T=2^24, N 16->8192, hash, 2^20 px."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import models, ops
dev = torch.device("cuda")
models.should_use_hash_function = True
net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=2**24, num_levels=16, n_min=16, n_max=8192, MLP_hidden_layers_widths=[64, 64],
                                      HPD_hidden_layers_widths=[32, 64, 128], HPD_out_features=2**24, feature_dim=4, topk_k=4,
                                      table_dtype=(torch.float16 if "fp16" in sys.argv else torch.float32)).to(dev)
net.return_indices = False
P = 2**20
g = torch.Generator(device=dev).manual_seed(65535)
xy = (torch.randint(0, 8192, (P, 2), device=dev, generator=g).float() / 8191).contiguous()
tgt = torch.rand((P, 3), device=dev, generator=g)
plan = ops.EncodePlan(P, net._n_ls_host, 4)
print("n_ls", net._n_ls_host, "staged", plan.Ls, "tile_shift", plan.tile_shift, "lds", plan.lds_bytes)
params = [p for p in net.parameters() if p.requires_grad]
def step():
    for p in params: p.grad = None
    rgb, *_ = net(xy, 1.0)
    torch.nn.functional.mse_loss(rgb, tgt).backward()
for _ in range(2): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"cfg5-shape ({net.encoding._base.dtype} tables) hash fwd+bwd: {ms:.3f} ms/step  {P/ms/1e3:.1f} Mpixel/s")
# direct-vs-tiled forward agreement at this size
e1_ = ops.encode_apply(xy, net._n_ls_flat(dev), net._n_ls_host, net.encoding.packed_tables(), None, None, 0, path="direct")
e2_ = ops.encode_apply(xy, net._n_ls_flat(dev), net._n_ls_host, net.encoding.packed_tables(), None, None, 0, path="tiled")
print("forms agree bit-for-bit:", bool(torch.equal(e1_, e2_)))
