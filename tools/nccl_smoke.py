"""Single-process exercise of the RCCL code path of bench.py's N > 1 step (world-size-1 'nccl' group on one GPU):
graph-replayed forward + backward with the deferred vertex stage, then the eager exchange.  Checks that capture with an
initialised RCCL communicator works and that collectives interleave with replays; numbers are not meaningful."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
import bench
from collision_handling_in_instantngp_amd import ops, parallel, train
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
xy, target, bounds = bench.make_batch("cfg2", 2 ** 20, 0, dev)
world = 2          # pretend: the exchange code runs, the group really has one rank
net, models = bench.build_model("gngf_frozen", dev, bounds)
parallel.enable_vertex_grid_exchange(net, world)
parallel.defer_vertex_stage(net, True)
loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
gs = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3)
gs(xy, target)
replay = gs.replay_only
t = torch.ones(8, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def both():
    replay(); parallel.allreduce_gradients(net, world, keep_tables_flag=True)
for _ in range(20): both()
print(f"graph replay + RCCL exchange: {timeit(both):.3f} ms/step at 2^20 px; table grad norm {float(net.encoding._grad_base.norm()):.3e}")
dG = net.dp.deferred[6]
print(f"replay only            {timeit(replay):.3f} ms")
print(f"all_reduce(dG {dG.numel() * 4 / 1e6:.1f} MB)  {timeit(lambda: dist.all_reduce(dG)):.3f} ms")
print(f"deferred vertex stage  {timeit(lambda: ops.run_deferred_vertex_stage(net.dp)):.3f} ms")
flat = parallel._flat_alias([p.grad for p in net.mlp.parameters()])
print(f"all_reduce(flat {flat.numel()} floats) {timeit(lambda: dist.all_reduce(flat)):.3f} ms")
print(f"allreduce_gradients    {timeit(lambda: parallel.allreduce_gradients(net, world, keep_tables_flag=True)):.3f} ms")
parallel.defer_vertex_stage(net, False)
parallel.enable_vertex_grid_exchange(net, 1)
dist.destroy_process_group()
