"""cfg4-shaped single-GPU probe: synthetic 4096^2 coordinates, L=16, F=2, T=2^22, N 16->4096, hash indexing, 2^20 px."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import models, ops
dev = torch.device("cuda")
ops.TILED_CELLS_PER_PIXEL = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
models.should_use_hash_function = True
net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=2**22, num_levels=16, n_min=16, n_max=4096, MLP_hidden_layers_widths=[64, 64],
                                      HPD_hidden_layers_widths=[32, 64, 128], HPD_out_features=2**22, feature_dim=2, topk_k=4).to(dev)
net.return_indices = False
P = 2**20
g = torch.Generator(device=dev).manual_seed(65535)
xy = (torch.randint(0, 4096, (P, 2), device=dev, generator=g).float() / 4095).contiguous()
tgt = torch.rand((P, 3), device=dev, generator=g)
plan = ops.EncodePlan(P, net._n_ls_host, 2)
print("n_ls", net._n_ls_host, "staged levels", plan.Ls, "tile_shift", plan.tile_shift, "lds", plan.lds_bytes)
params = [p for p in net.parameters() if p.requires_grad]
def step():
    for p in params: p.grad = None
    rgb, *_ = net(xy, 1.0)
    torch.nn.functional.mse_loss(rgb, tgt).backward()
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"cfg4-shape hash fwd+bwd: {ms:.3f} ms/step  {P/ms/1e3:.1f} Mpixel/s")
