"""debug: which ingredient of test_hipgraph_replay_of_a_step_equals_eager crashes capture_end"""
import sys, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, ".")
from collision_handling_in_instantngp_amd import models, train
variant = sys.argv[1]
DEV = "cuda"
models.should_use_hash_function = True
torch.manual_seed(1)
big = "big" in variant
net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=2 ** (19 if big else 15), num_levels=(16 if big else 8), n_min=16, n_max=(512 if big else 128),
                                      MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                      HPD_out_features=2 ** 15, feature_dim=2, topk_k=4)
net.return_indices = "idx" in variant
P = 57404 if "p57" in variant else 60000
xy = torch.rand((P, 2), device=DEV); tgt = torch.rand((P, 3), device=DEV)
loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
opt = train.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6) if "opt" in variant else None
def eager():
    rgb, probs, _i, _c = net(xy, 1.0)
    if "fwdonly" in variant:
        return rgb
    mse, kls, coll = loss_fn(rgb, tgt, None, probs, torch.tensor([], device=DEV), torch.tensor([], device=DEV))
    loss = train.assemble_loss(mse, kls, coll, 1, 1, 1e-3)
    if "gradarg" in variant:
        loss.backward(gradient=torch.ones((), device=DEV))
    else:
        loss.backward()
    return rgb
if "eager" in variant:
    if "stream" in variant:
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            keep = eager()
        torch.cuda.current_stream().wait_stream(s_)
    else:
        keep = eager()
    if "clean" in variant:
        del keep
        net.zero_grad(set_to_none=True)
        import gc
        torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    if "sync" in variant:
        torch.cuda.synchronize()
gs = train.GraphedStep(net, loss_fn, opt, 1, 1, 1e-3)
r = gs(xy, tgt)
r = gs(xy, tgt)
torch.cuda.synchronize()
print(variant, "ok", float(r.mse))
