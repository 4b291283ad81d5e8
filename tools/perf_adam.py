import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from collision_handling_in_instantngp_amd import train
dev = torch.device("cuda")
for mode in ("hash", "gngf_frozen"):
    net, models = bench.build_model(mode, dev)
    xy, tgt, _ = bench.strawberry_batch(2**20, 0, dev)
    step = bench.make_step(net, models, mode, xy, tgt, 1)
    step()
    for kw in ({}, {"fused": True}, {"foreach": True}, "gngf_adam_step"):
        groups = [{"params": net.encoding.parameters(), "lr": 1e-4, "weight_decay": 0}, {"params": net.mlp.parameters(), "lr": 1e-3, "weight_decay": 1e-6}]
        opt = train.FusedAdam(groups, betas=(0.9, 0.99), eps=1e-15) if isinstance(kw, str) else torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-15, **kw)
        for _ in range(3): opt.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): opt.step()
        e1.record(); torch.cuda.synchronize()
        print(mode, kw, f"{e0.elapsed_time(e1)/10*1e3:.1f} us per Adam step")
    models.should_use_hash_function = False
