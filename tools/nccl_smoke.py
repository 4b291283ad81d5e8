"""Single-process exercise of the RCCL code path of bench.py's N > 1 step (world-size-1 'nccl' group on one GPU):
graph-replayed forward + backward with the deferred vertex stage, then the eager exchange.  Checks that capture with an
initialised RCCL communicator works and that collectives interleave with replays; numbers are not meaningful."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
import bench
from collision_handling_in_instantngp_amd import parallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
xy, target, _ = bench.strawberry_batch(2 ** 20, 0, dev)
world = 2          # pretend: the exchange code runs, the group really has one rank
parallel.enable_vertex_grid_exchange(world)
parallel.defer_vertex_stage(True)
net, models = bench.build_model("gngf_frozen", dev)
replay = bench.graphed(bench.make_step(net, models, "gngf_frozen", xy, target, world, exchange=False))
t = torch.ones(8, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    replay()
    parallel.allreduce_gradients(net, world, keep_tables_flag=True)
torch.cuda.synchronize()
print(f"graph replay + RCCL exchange: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step at 2^18 px; table grad norm "
      f"{float(net.encoding._grad_base.norm()):.3e}")
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
from collision_handling_in_instantngp_amd import ops
dG = ops.DP_DEFERRED[6]
print(f"replay only            {timeit(replay):.3f} ms")
print(f"all_reduce(dG 5.7 MB)  {timeit(lambda: dist.all_reduce(dG)):.3f} ms")
print(f"dG.mul_                {timeit(lambda: dG.mul_(0.5)):.3f} ms")
print(f"deferred vertex stage  {timeit(ops.run_deferred_vertex_stage):.3f} ms")
flat = parallel._flat_alias([p.grad for p in net.mlp.parameters()])
print(f"all_reduce(flat {flat.numel()} floats) {timeit(lambda: dist.all_reduce(flat)):.3f} ms")
print(f"allreduce_gradients    {timeit(lambda: parallel.allreduce_gradients(net, world, keep_tables_flag=True)):.3f} ms")
def both():
    replay(); parallel.allreduce_gradients(net, world, keep_tables_flag=True)
print(f"replay + exchange      {timeit(both):.3f} ms")
parallel.defer_vertex_stage(False)
net2, models = bench.build_model("gngf_frozen", dev)
eager = bench.make_step(net2, models, "gngf_frozen", xy, target, world, exchange=True)
for _ in range(5): eager()
print(f"eager step + in-backward exchange (old path) {timeit(eager):.3f} ms")
eager1 = bench.make_step(net2, models, "gngf_frozen", xy, target, 1, exchange=False)
parallel.enable_vertex_grid_exchange(1)
for _ in range(5): eager1()
print(f"eager step, no exchange {timeit(eager1):.3f} ms")
dist.destroy_process_group()
