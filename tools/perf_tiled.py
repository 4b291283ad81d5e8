import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops
from collision_handling_in_instantngp_amd import models as orc   # level_resolutions only (the CPU oracle is for tests)
import bench
dev = torch.device("cuda")
xy, target, _b = bench.make_batch("cfg2", 2**20, 0, dev)
n_host = [int(v) for v in orc.level_resolutions(16, 512, 16)]
n_ls = torch.tensor(n_host, dtype=torch.int32, device=dev)
tables = (torch.rand((16, 2**19, 2), device=dev) - 0.5) * 2e-4
genc = torch.randn((2**20, 32), device=dev)
for chunk in (int(c) for c in (sys.argv[1:] or ["1024"])):
    ops.TILED_CHUNK = chunk
    ks = ops.encode_kernels(xy, n_ls, n_host, tables, None, None, 0, genc)
    for name, fn in ks.items():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"chunk {chunk:5d} {name:20s} {e0.elapsed_time(e1)/10*1e3:9.1f} us")
