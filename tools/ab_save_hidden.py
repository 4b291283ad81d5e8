"""A/B of ops.DECODER_SAVE_HIDDEN on the whole training step (graph replay), same process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from collision_handling_in_instantngp_amd import ops
dev = torch.device("cuda")
xy, target, _ = bench.strawberry_batch(2**20, 0, dev)
for mode in ("gngf_frozen", "hash"):
    for save in (True, False, True, False):
        ops.DECODER_SAVE_HIDDEN = save
        net, models = bench.build_model(mode, dev)
        step = bench.graphed(bench.make_step(net, models, mode, xy, target, 1))
        for _ in range(100): step()                      # steady clock
        dt = bench.timed(step, 200, 5, 1)
        print(f"{mode:12s} save_hidden={save!s:5s}  {dt / 200 * 1e3:.4f} ms/step  {2**20 * 200 / dt / 1e6:.1f} Mpixel/s")
        models.should_use_hash_function = False
        del net, step
        torch.cuda.empty_cache()
