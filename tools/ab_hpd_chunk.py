"""Learning-mode step time vs the HPD row-chunk size (models.HPD_CHUNK_BYTES: rows = bytes / (4 T))."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from collision_handling_in_instantngp_amd import models as M
dev = torch.device("cuda")
xy, target, _ = bench.strawberry_batch(2 ** 20, 0, dev)
for gib in (4, 8, 16, 32):
    M.HPD_CHUNK_BYTES = gib << 30
    net, models = bench.build_model("gngf_learning", dev)
    step = bench.make_step(net, models, "gngf_learning", xy, target, 1)
    dt = bench.timed(step, 2, 1, 1)
    print(f"chunk {gib:2d} GiB ({(gib << 30) // (4 * bench.T)} rows): {dt / 2 * 1e3:8.1f} ms/step, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del net, step
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
