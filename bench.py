#!/usr/bin/env python3
"""Headline benchmark: Mpixel/s forward+backward of the hash-grid encoder path (coords -> encoder -> decoder ->
MSE gradient -> decoder/table gradients) at L=16, F=2, T=2^19, batch 2^20 pixels per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(WORLD_SIZE set), or bare (`python bench.py --gpus N`): the parent then starts the N ranks itself — before it has touched
the GPU — and exits with their status.  A WORLD_SIZE that disagrees with --gpus is refused.

One JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed on the
launch stream), `roofline_encoder`, `roofline_step` and `cpu_baseline` (CPU oracle timed on the host cores, bounded
sample, rank 0, N=1 only).

Modes (--mode picks the headline; the others are reported in the same line under "modes"):
  gngf_frozen    cfg2, GNGF indexing with a frozen HashProbDistribution (the reference's -hwp mode, models.py:364-371):
                 the per-vertex top-K table is a function of frozen weights and is rebuilt only when they change.
  gngf_learning  cfg2, GNGF indexing with a trainable HPD: every step re-evaluates the HPD on every distinct vertex
                 (MFMA-bound: U x 128 x T contraction) and back-propagates into it.
  hash           BASELINE.json configs[2]: macaw.jpg's own pixel list (508x339, tests/golden/macaw_rgb.npz) at the cfg2 table shape
                 with plain spatial-hash indexing (should_use_hash_function=True).
  cfg4_hash      BASELINE.json configs[3] per-GPU shape: synthetic 4096^2 image, T = 2^22, N 16->4095, 2^20 px per GPU.
  cfg5_hash_fp16 BASELINE.json configs[4] per-GPU shape: synthetic 8192^2 image, F = 4, T = 2^24, fp16 tables, N 16->8191.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P_PER_GPU = 2 ** 20
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)

SHAPES = {     # L, F, T, K, n_min, n_max, fp16 tables, image
    "cfg2": dict(L=16, F=2, T=2 ** 19, K=4, n_min=16, n_max=512, half=False, image="strawberry"),
    "cfg3": dict(L=16, F=2, T=2 ** 19, K=4, n_min=16, n_max=512, half=False, image="macaw"),
    "cfg4": dict(L=16, F=2, T=2 ** 22, K=4, n_min=16, n_max=4096, half=False, image=4096),
    "cfg5": dict(L=16, F=4, T=2 ** 24, K=4, n_min=16, n_max=8192, half=True, image=8192),
}
MODES = {"gngf_frozen": "cfg2", "gngf_learning": "cfg2", "hash": "cfg3", "cfg4_hash": "cfg4", "cfg5_hash_fp16": "cfg5"}
XGMI_LINK_GBS = 153.0          # per link and direction (7 links per GPU)


def is_hash(mode):
    return mode == "hash" or mode.startswith("cfg")


def survey_bytes(mode):
    """algorithmic bytes per pixel of the per-instance formulation (~0 reuse assumed)."""
    c = SHAPES[MODES[mode]]
    L, F, K = c["L"], c["F"], c["K"]
    sf = 2 if c["half"] else 4
    Ke = 1 if is_hash(mode) else K
    J = 0 if is_hash(mode) else 8 * K
    b_fwd = 8 + L * 4 * Ke * F * sf + L * 4 * J + L * F * 4
    b_bwd = 8 + L * F * 4 + L * 4 * J + 2 * L * 4 * Ke * F * sf + (L * 4 * K * F * sf if not is_hash(mode) else 0)
    return b_fwd, b_bwd


def compulsory_bytes(mode):
    """Bytes per pixel the IMPLEMENTATION must move in the pixel stage of the tiled form: one 16-byte binned
    pixel record in, one (L*F*4)-byte row of enc out (forward) or of d enc in (backward).  The per-vertex table traffic is
    ~1 % of it at cfg2 (0.72 M vertex rows against 2^20 pixels x 16 levels) and lives in L2 / Infinity Cache."""
    c = SHAPES[MODES[mode]]
    row = c["L"] * c["F"] * 4
    return 16 + row, 16 + row


# ------------------------------------------------------------------------------------------------ launching N ranks
def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes through torch.distributed.run and exit
    with their status.  Nothing here has touched the GPU (no CUDA call before this point, and none in this process at all)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="gngf_frozen", choices=list(MODES))
    ap.add_argument("--pixels", type=int, default=P_PER_GPU, help="pixels per GPU per step")
    ap.add_argument("--no-extra-modes", action="store_true")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B measurements: set a module-level switch of collision_handling_in_instantngp_amd.ops (e.g. DECODER_CLEARS_DG64=0)")
    ap.add_argument("--no-full-outputs", dest="full_outputs", action="store_false",
                    help="skip the extra timing of the step with the reference's index tensor materialised")
    ap.add_argument("--no-unroll", dest="unroll", action="store_false", help="one step per replayed graph")
    ap.add_argument("--unrolls", type=lambda v: tuple(int(x) for x in v.split(",")), default=(10, 8, 5, 4, 6, 7, 3, 2),
                    help="steps per replayed graph: the first of these that divides --steps (inside a graph, step j's pixel-stage "
                         "launches carry the binning of step j + 1's batch; the first step of a replay bins itself)")
    ap.add_argument("--cpu-sample", type=int, default=2 ** 20)
    ap.add_argument("--ramp-steps", type=int, default=60, help="untimed steps before the W warm-up steps (clock ramp)")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: 'gloo' runs the N>1 code path with ranks sharing one GPU")
    return ap.parse_args(argv)


def resolve_world(a, environ=None):
    """(action, world): 'spawn' when N ranks must be started from here, 'run' otherwise.  Refuses (SystemExit 2) a launcher
    world size that disagrees with --gpus: the line's n_gpus must be what actually ran."""
    environ = os.environ if environ is None else environ
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in environ:
        return ("spawn", a.gpus) if a.gpus > 1 else ("run", 1)
    world = int(environ["WORLD_SIZE"])
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a mislabelled run",
              file=sys.stderr)
        raise SystemExit(2)
    return "run", world


# ------------------------------------------------------------------------------------------------ workload
def make_batch(cfg_name, P, rank, dev):
    """cfg2 / cfg3: the strawberry / macaw pixel list (508 x 339; coords = (row,col)/507, main.py:50-51), shuffled and repeated
    to P pixels (the images have only 172 212 pixels).  cfg4 / cfg5: P pixels drawn from a synthetic S x S uniform-random RGB image
    (coords (row, col) / (S - 1)), seed 65535 + rank."""
    import numpy as np
    import torch
    c = SHAPES[cfg_name]
    g = torch.Generator().manual_seed(65535 + rank)
    if c["image"] in ("strawberry", "macaw"):
        img = np.load(os.path.join(ROOT, "tests", "golden", c["image"] + "_rgb.npz"))["img"]
        h, w = img.shape[:2]
        sel = torch.cat([torch.randperm(h * w, generator=g) for _ in range(-(-P // (h * w)))])[:P]
        rows, cols = sel // w, sel % w
        xy = torch.stack([rows, cols], 1).float() / (max(w, h) - 1)
        rgb = torch.from_numpy(img.reshape(-1, 3))[sel].float() / 255
        bounds = (1.0, (min(h, w) - 1) / (max(h, w) - 1))
    else:
        S = int(c["image"])
        rc = torch.randint(0, S, (P, 2), generator=g)
        xy = rc.float() / (S - 1)
        rgb = torch.randint(0, 256, (P, 3), generator=g).float() / 255
        bounds = (1.0, 1.0)
    return xy.to(dev).contiguous(), rgb.to(dev).contiguous(), bounds


def build_model(mode, dev, bounds):
    import torch
    from collision_handling_in_instantngp_amd import models
    from collision_handling_in_instantngp_amd import ops
    c = SHAPES[MODES[mode]]
    models.should_use_hash_function = is_hash(mode)
    # fp16 tables: the gradient is handed over as the fp32 buffer it is accumulated in (what train.FusedAdam consumes), not
    # cast to an fp16 `.grad` first (6 GiB of traffic per step at the cfg5 shape)
    ops.FP16_TABLE_GRAD_FP32 = bool(c["half"])
    torch.manual_seed(65535)
    net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=c["T"], num_levels=c["L"], n_min=c["n_min"], n_max=c["n_max"],
                                          MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                          HPD_out_features=c["T"], feature_dim=c["F"], topk_k=c["K"],
                                          table_dtype=(torch.float16 if c["half"] else torch.float32)).to(dev)
    net.return_indices = False          # the (P,L,4,K) int64 diagnostics tensor (2 GiB/step) is not part of fwd+bwd
    net.dense_probs = False             # the dense (P,L,4,T) tensor is 128 MiB *per pixel* at this shape
    if mode == "gngf_frozen":
        for p in net.HPD.parameters():
            p.requires_grad = False
        net.compute_pbar = False        # the KL/JS term has no trainable input when the HPD is frozen
    if mode == "gngf_learning":
        net.coord_bounds = bounds
    return net, models


def eager_step_fn(net, mode, xy, target, world, exchange=True):
    """One training step launched eagerly: forward, loss, backward (+ the gradient exchange when world > 1 and `exchange`)."""
    import torch
    from collision_handling_in_instantngp_amd import train, parallel
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    params = [p for p in net.parameters() if p.requires_grad]
    empty = torch.tensor([], device=xy.device)
    one = torch.ones((), device=xy.device)          # the seed of backward(): loss.backward() would fill a fresh one per step
    T = net._hash_table_size
    net.dp.persist_ok = True      # this loop lets go of every gradient at the top of each step (ops.PERSISTENT_TABLE_GRAD is opt-in)

    def step():
        for p in params:
            p.grad = None
            if getattr(p, "grad_fp32", None) is not None:      # fp16 tables: the fp32 hand-over buffer is the gradient (ops._grad_out)
                p.grad_fp32 = None
        with net.fused_mse(target, gloss=1.0):     # as train.py's loops do: the pixel loss rides in the decoder kernels
            rgb, probs, _idx, _c = net(xy, 1.0)
        mse, kls, coll = loss_fn(rgb, target, None if probs is None else T, probs, empty, empty)
        loss = train.assemble_loss(mse, kls, coll, 1, 1, 1e-3)    # frozen HPD / hash: the MSE term alone (no p-bar, no gradient in the rest)
        loss.backward(gradient=one)
        if world > 1 and exchange:
            parallel.allreduce_gradients(net, world)
    return step


def timed(step, steps, warmup, world):
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)                   # every rank's own clock over the same barrier-bracketed region
        LAST_PER_RANK[:] = [float(e.item()) for e in every]
        dt = max(LAST_PER_RANK)                      # the contract's value: the slowest rank
    return dt


LAST_PER_RANK = []     # seconds each rank measured in the last timed() call (N > 1)


def measure_hbm_copy_gbs(dev):
    """Streaming-copy bandwidth of THIS box, timed in the same run ("the builder must also report a
    measured streaming-copy bandwidth on the box and the fraction against both"): 512 MiB device-to-device copy, read +
    write bytes over the HIP-event time of 10 copies."""
    import torch
    x = torch.empty(2 ** 27, device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    gbs = 2 * x.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del x, y
    torch.cuda.empty_cache()
    return gbs


def exchange_model(net, world, P=P_PER_GPU):
    """Bytes one rank hands to the gradient exchange per step and the time a ring all-reduce of them takes on the xGMI links
    (2 (n-1)/n x bytes per rank, spread over the 7 links of 153 GB/s each when n = 8, over n-1 links otherwise) — a MODEL, not
    a measurement.  The bytes follow from the model's geometry alone (ops.EncodePlan: which levels are staged — their
    vertex-grid gradient travels — and which run in the direct form — their slice of the fp32 table gradient travels; every
    other trainable parameter's gradient), so the N = 1 line states the prediction for n = 2, 4, 8 that a later multi-GPU run is
    read against (`exchange_model`), and the N > 1 line repeats it for its own n (`exchange`)."""
    from collision_handling_in_instantngp_amd import ops
    enc = net.encoding
    L, T, F = enc.packed_tables().shape
    plan = ops.EncodePlan(P, net._n_ls_host, F)
    staged = plan.vtot * F * 4 if plan.Ls > 0 else 0
    direct = (L - plan.Ls) * T * F * 4                 # the table gradient is accumulated (and exchanged) in fp32 whatever the storage
    dense = sum(p.numel() * 4 for n_, p in net.named_parameters() if p.requires_grad and "_hash_tables" not in n_)
    total = staged + direct + dense
    links = min(7, max(1, world - 1))
    t = 2.0 * (world - 1) / world * total / (links * XGMI_LINK_GBS * 1e9) if world > 1 else 0.0
    return {"n": world, "staged_levels": plan.Ls, "vertex_grid_bytes": staged, "direct_level_table_bytes": direct, "decoder_and_hpd_bytes": dense,
            "exchange_bytes_per_step": total, "modelled_ring_allreduce_ms": t * 1e3, "links_used": links, "link_GBs": XGMI_LINK_GBS,
            "note": "model: ring all-reduce, 2(n-1)/n x bytes per rank over the point-to-point xGMI links; latency not included; the "
                    "exchange follows the step's graph on a communication stream and is NOT overlapped with the next step's compute"}


ENTRY_NAMES = {"gngf_bin_pixels": "bin_pixels", "gngf_bin_pixels2": "bin_pixels(count+scatter)", "gngf_encode_tiled_fwd_fused": "encode_fwd:tiled", "gngf_encode_tiled_prepare": "prepare(bin+vertex_fwd+clears)", "gngf_vertex_grid_fwd": "vertex_fwd", "gngf_encode_tiled_fwd": "encode_fwd:tiled",
               "gngf_encode_tiled_bwd": "encode_bwd:tiled", "gngf_vertex_grid_bwd_sorted": "vertex_bwd", "gngf_vertex_grid_bwd": "vertex_bwd",
               "gngf_decoder_fwd": "decoder_fwd", "gngf_decoder_bwd": "decoder_bwd", "gngf_decoder_train": "decoder_train", "gngf_decoder_reduce": "decoder_reduce", "gngf_mse_fwd": "mse_fwd", "gngf_mse_bwd": "mse_bwd",
               "gngf_encode_fwd": "encode_fwd:direct", "gngf_encode_bwd": "encode_bwd:direct",
               "gngf_encode_bwd_bucketed": "encode_bwd:direct(bucketed)"}


def kernel_times_in_step(eager_step, n=20, warm=3):
    """Average launch duration of every C-ABI entry point INSIDE the training step: HIP events recorded on the stream
    each kernel is launched on (torch's current stream at the call; the helper stream for the vertex stage), bracketing
    every call of n eagerly launched steps.  Returns ({name: seconds per step}, {name: calls per step})."""
    import torch
    from collision_handling_in_instantngp_amd import _lib
    for _ in range(warm):
        eager_step()
    torch.cuda.synchronize()
    _lib.PROFILE = {}
    try:
        for _ in range(n):
            eager_step()
        torch.cuda.synchronize()
        prof = _lib.PROFILE
    finally:
        _lib.PROFILE = None
    out, calls = {}, {}
    for name, pairs in prof.items():
        key = ENTRY_NAMES.get(name, name)
        ms = [a.elapsed_time(b) for a, b in pairs]
        out[key] = out.get(key, 0.0) + sum(ms) / n * 1e-3
        calls[key] = calls.get(key, 0) + len(pairs) / n
    return out, calls


def cpu_baseline(mode, sample_pixels):
    """The CPU oracle timed on this host's cores (kind "port"): forward + backward of the same path (encoder with the
    index table given, decoder, MSE gradient) on a bounded sample of the same workload.  Uses the C/OpenMP
    restatement (oracle/gngf_oracle_c.c) when built, else the numpy one.  Checker code is only timed here."""
    import numpy as np
    from oracle import gngf_oracle as orc, c_oracle
    c = SHAPES["cfg2"]
    L, F, T, K_TOP, N_MIN, N_MAX = c["L"], c["F"], c["T"], c["K"], c["n_min"], c["n_max"]
    rng = np.random.default_rng(0)
    img = np.load(os.path.join(ROOT, "tests", "golden", "strawberry_rgb.npz"))["img"]
    h, w = img.shape[:2]
    sel = np.concatenate([rng.permutation(h * w) for _ in range(-(-sample_pixels // (h * w)))])[:sample_pixels]
    x = np.ascontiguousarray((np.stack([sel // w, sel % w], 1) / np.float32(max(w, h) - 1)).astype(np.float32))
    y = np.ascontiguousarray((img.reshape(-1, 3)[sel] / 255).astype(np.float32))
    n_ls = orc.level_resolutions(N_MIN, N_MAX, L)
    tables = ((rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-4).astype(np.float32)
    dims = [L * F, 64, 64, 3]
    dw = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    db = [np.zeros(dims[i + 1], np.float32) for i in range(3)]
    vstride = N_MAX + 2
    vidx = vw = None
    if not is_hash(mode):
        vidx = rng.integers(0, T, (vstride * vstride, K_TOP)).astype(np.int32)
        vw = rng.random((vstride * vstride, K_TOP), dtype=np.float32)

    if c_oracle.available():
        def step():
            enc = c_oracle.encode_fwd(x, tables, n_ls, vidx, vw, vstride)
            rgb, h1, h2 = c_oracle.decoder_fwd(enc, dw, db)
            grgb = ((2.0 / rgb.size) * (rgb - y)).astype(np.float32)
            genc, _ = c_oracle.decoder_bwd(enc, h1, h2, rgb, grgb, dw)
            c_oracle.encode_bwd(x, tables, n_ls, genc, vidx, vw, vstride)
        cores, impl = c_oracle.num_threads(), "C/OpenMP oracle"
    else:
        def step():
            _, grid = orc.scale_to_grid(x, n_ls)
            if is_hash(mode):
                idx, probs = orc.spatial_hash(grid.astype(np.int32), T), None
            else:
                gi = grid.astype(np.int64)
                vid = gi[:, 1] * vstride + gi[:, 0]
                idx, probs = vidx[vid].astype(np.int64), vw[vid]
            enc = orc.bilinear_forward(x, n_ls, orc.encoding_forward(tables, idx, probs, None))
            rgb = orc.decoder_forward(enc, dw, db)
            genc, _, _ = orc.decoder_backward(enc, dw, db, (2.0 / rgb.size) * (rgb - y))
            orc.encoding_backward(tables, idx, probs, None, orc.bilinear_backward(x, n_ls, genc, F))
        cores, impl = 1, "numpy oracle (single-threaded)"
    step()                                            # warm-up (page faults, thread pool)
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 16:
            break
    return {"value": sample_pixels * n / dt / 1e6, "unit": "Mpixel/s", "cores": cores, "kind": "port",
            "sample": f"{n} x {sample_pixels} strawberry pixels, fwd+bwd (encoder + decoder + MSE grad), "
                      f"{'hash' if is_hash(mode) else 'vertex-table'} indexing with the index table given, {impl}, {dt:.1f} s"}


def learning_flops(stats, Kdim=128):
    """FLOP of the last HPD layer per learning step: logits (2*U*128*T), dW and dh (the same each), plus the logits GEMM again
    for every chunk whose logits were not kept from the forward."""
    U, T = stats["rows_total"], stats["T"]
    gemm = 2.0 * U * Kdim * T
    recomputed = stats["chunks"] - stats["chunks_kept"]
    return 3.0 * gemm + gemm * recomputed / max(1, stats["chunks"]), gemm


def main():
    a = parse_args()
    action, world = resolve_world(a)
    if action == "spawn":
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    collective_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.cuda.set_device(0)
            dist.init_process_group(a.backend)
        probe = torch.ones(1, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(probe)                       # how many ranks actually take part in a collective
        collective_ranks = int(probe.item())
        if collective_ranks != world or dist.get_world_size() != world:
            raise SystemExit(f"bench.py: {collective_ranks} ranks answered the all-reduce, WORLD_SIZE says {world}")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    from collision_handling_in_instantngp_amd import ops, parallel, train
    for kv in a.set:
        name, _, val = kv.partition("=")
        if not hasattr(ops, name):
            raise SystemExit(f"bench.py --set: ops has no switch {name!r}")
        cur = getattr(ops, name)
        setattr(ops, name, type(cur)(int(val)) if isinstance(cur, (bool, int)) else (float(val) if isinstance(cur, float) else val))
    P = a.pixels

    results = {}
    broken = False
    in_graph_ms = None
    hbm_copy = measure_hbm_copy_gbs(dev)          # this box's streaming-copy bandwidth, same run (fractions against it below)
    # the other modes are single-GPU side measurements: at N > 1 only the mode asked for runs (each mode would have to be collective-
    # matched across the ranks, a failure in any of them would cost the scaling line, and the driver computes scaling from `value`)
    extra = [] if (a.no_extra_modes or world > 1) else [m for m in ("gngf_learning", "hash", "gngf_frozen", "cfg4_hash", "cfg5_hash_fp16")
                                                         if m != a.mode]
    kt, kcalls = {}, {}
    batches = {}
    for mode in [a.mode] + extra:
        # an extra mode that fails on this box (out of memory at the 4 GiB shape, say) must not cost the headline line: at
        # N = 1 it is recorded as an error and the run goes on; the headline mode, and every mode at N > 1 (collectives
        # must stay matched across ranks), fail loudly
        try:
            head = mode == a.mode
            cfg_name = MODES[mode]
            if cfg_name not in batches:
                batches.clear()
                batches[cfg_name] = make_batch(cfg_name, P, rank, dev)
            xy, target, bounds = batches[cfg_name]
            learning = mode == "gngf_learning"
            # (the other single-kernel-chain modes: as many steps as the headline — 10 steps of 0.4 ms are a 4 ms window, ±1 % of noise)
            steps, warmup = (a.steps, a.warmup) if head else ((2, 1) if learning else (max(5, a.steps), max(2, a.warmup)))
            if learning and head:
                steps, warmup = min(a.steps, 5), min(a.warmup, 1)
            net, models = build_model(mode, dev, bounds)
            if world > 1:
                parallel.enable_vertex_grid_exchange(net, world)
            loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
            step = eager_step_fn(net, mode, xy, target, world)
            launch = "eager"
            unroll = 1
            if a.graph and not learning:
                # world > 1: the vertex stage of the encoder backward is deferred behind the dG exchange, so forward + backward
                # hold no collective and replay from one hipGraph; the exchange (RCCL) and the vertex stage follow eagerly.
                try:
                    if world > 1:
                        parallel.defer_vertex_stage(net, True)
                    # several steps per replayed graph (a replay costs ~9 us of launch latency whatever it holds): the first of
                    # 4, 5, 6, 7, 8, 3, 2 that divides K, so that exactly K steps are timed
                    unroll = 1
                    if world == 1 and a.unroll:
                        unroll = next((u for u in a.unrolls if steps % u == 0), 1)
                    # cross_replay: the last step of a replay bins the first batch of the next one (three calls capture the cold
                    # graph and the two steady ones; the batches stay in the static buffers)
                    gs = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3, unroll=unroll, cross_replay=True)
                    for _ in range(3):
                        if unroll > 1:
                            gs.run_many([(xy, target)] * unroll, next_first=xy)
                        else:
                            gs(xy, target, next_first=xy)
                    if world > 1:
                        # the exchange (RCCL) and the deferred vertex stage go to the model's communication stream; the next
                        # replay waits for them on the device (it overwrites the exchanged buffers from its first kernel on)
                        def step(gs=gs, net=net):
                            parallel.wait_for_gradients(net)
                            gs.replay_steady()
                            parallel.allreduce_gradients(net, world, keep_tables_flag=True, overlap=True)
                        launch = "hipGraph + exchange on a communication stream"
                    else:
                        step = gs.replay_steady
                        launch = "hipGraph" if unroll == 1 else f"hipGraph ({unroll} steps per replay)"
                except Exception as e:  # pragma: no cover
                    print(f"[bench] graph capture failed ({e!r}); running eagerly", file=sys.stderr)
                    if world > 1:
                        parallel.defer_vertex_stage(net, False)
                    step = eager_step_fn(net, mode, xy, target, world)
            if not learning:
                # Clock ramp: the chip reaches its steady matrix-core clock only after ~15 ms of sustained work (the same kernel
                # is ~10 % slower before; tools/perf_decoder_warm.py), and W warm-up steps of 0.6 ms do not get there.  Untimed
                # steps first (every rank runs the same number, so collectives stay matched), then the W + K of the contract.
                for _ in range(a.ramp_steps // (unroll if launch.startswith("hipGraph (") else 1)):
                    step()
            per = unroll if (a.graph and not learning and launch.startswith("hipGraph (")) else 1
            dt = timed(step, steps // per, -(-warmup // per), world)       # exactly `steps` steps: steps / per replays of `per` steps each
            res = results[mode] = {"mpix_s": P * world * steps / dt / 1e6, "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
                                   "launch": launch, "shape": {k: v for k, v in SHAPES[cfg_name].items()}}
            sc_ = getattr(net.dp, "step_config", None)
            res["step_config"] = sc_.signature() if sc_ is not None else None      # the kernel chain this mode's steps took (ops.StepConfig)
            res["step_chain"] = sc_.chain() if sc_ is not None else None
            if world > 1:
                res["ms_per_step_per_rank"] = [t_ / steps * 1e3 for t_ in LAST_PER_RANK]
            if not learning:
                # the contract's window is `steps` steps (8 ms at the default K): five more windows of the same length, each bracketed
                # like the first, show how much one window moves (every rank runs them: the barriers stay matched)
                res["ms_per_step_windows"] = [timed(step, steps // per, 0, world) / steps * 1e3 for _ in range(5)]
            # the target's own metric: per-instance algorithmic bytes x pixels/s against the HBM peak.  For
            # GNGF indexing that figure assumes one table gather per (pixel, corner, k); the per-vertex de-duplicated algorithm does
            # not move those bytes, so its ratio exceeds 1 — it is printed as what it is, next to the hash modes' real fractions.
            sb_f, sb_b = survey_bytes(mode)
            sv = (sb_f + sb_b) * P * steps / dt / 1e9
            res["roofline_survey"] = {"bytes_per_pixel": sb_f + sb_b, "achieved_GBs": sv, "frac_of_8TBs": sv / HBM_PEAK_GBS,
                                      "frac_of_measured_copy": sv / hbm_copy}
            if world > 1:
                res["exchange"] = exchange_model(net, world, P)
            else:
                # the prediction a multi-GPU run of this mode is to be held against (weak scaling: the same 2^20 px per GPU)
                res["exchange_model"] = {str(n): exchange_model(net, n, P) for n in (2, 4, 8)}
                for em in res["exchange_model"].values():
                    em["predicted_ms_per_step"] = res["ms_per_step"] + em["modelled_ring_allreduce_ms"]
                    em["predicted_scaling_efficiency"] = res["ms_per_step"] / em["predicted_ms_per_step"]
            if SHAPES[cfg_name]["half"]:
                res["table_gradient"] = "fp32 accumulation buffer handed over as param.grad_fp32 (ops.FP16_TABLE_GRAD_FP32), no fp16 .grad copy"
            if learning:
                st = dict(net.hpd_stats)
                fl, gemm = learning_flops(st)
                res["hpd"] = st
                res["roofline"] = {"bound": "mfma", "kernel": "last HPD layer: logits / dW / dh GEMMs (128 x T per distinct vertex), fp32 MFMA",
                                   "achieved": fl / (dt / steps) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": fl / (dt / steps) / 1e12 / MFMA_F32_PEAK_TFLOPS, "flop_per_step": fl, "traffic": None,
                                   "note": "whole-step time against the fp32-EQUIVALENT FLOP of the three GEMMs (+ recomputed logits chunks), priced at "
                                           "the exact-fp32 MFMA peak as in rounds 3-4; the GEMMs run on the bf16 pipe (split operands): see bf16_pipe"}
                # what the matrix pipe actually executes: every fp32 product as 6 bf16 products (three-way split: the logits, forward and
                # recomputed) or 3 (two planes: dW and dh when ops.TUNING.hpd_bwd_two_planes), against the dense bf16 peak
                t_ = ops.TUNING
                nprod_bwd = 3.0 if (t_.hpd_bwd_two_planes and t_.hpd_gemm_kernel == 1) else 6.0
                issued = (fl - 2.0 * gemm) * 6.0 + 2.0 * gemm * nprod_bwd
                res["roofline"]["bf16_pipe"] = {"issued_tflop_per_step": issued / 1e12, "achieved": issued / (dt / steps) / 1e12, "peak": 2500.0,
                                                "unit": "TFLOP/s", "frac": issued / (dt / steps) / 1e12 / 2500.0,
                                                "products_per_fp32_product": {"logits": 6, "dW_dh": nprod_bwd}}
            if head and launch.startswith("hipGraph"):
                # decoder_bwd's duration INSIDE the replayed graph, from the device clock the kernel stamps (events cannot be
                # recorded in a replayed hipGraph here): one sample per burst of replays.  Every rank runs it: at world > 1 a step
                # carries the gradient exchange, and collectives must stay matched across ranks.
                try:
                    import ctypes
                    from collision_handling_in_instantngp_amd import _lib
                    spans = []
                    for _ in range(10):
                        for _ in range(8):
                            step()
                        torch.cuda.synchronize()
                        ns = ctypes.c_double(0.0)
                        _lib.call("gngf_decoder_bwd_last_span_ns", ctypes.byref(ns))
                        spans.append(ns.value)
                    spans.sort()
                    in_graph_ms = spans[len(spans) // 2] * 1e-6
                except Exception as e:  # pragma: no cover
                    in_graph_ms = None
                    print(f"[bench] in-graph span unavailable ({e!r})", file=sys.stderr)
            if head and world == 1 and a.graph and not learning:
                # the same step with the optimizer in the graph (get_optimizer's Adam as one launch): reported, not the metric
                try:
                    opt = train.get_optimizer(net, 1e-2, 1e-3, 1e-3, 0.0, 0.0, 1e-6)
                    gso = train.GraphedStep(net, loss_fn, opt, 1, 1, 1e-3)
                    gso(xy, target)
                    dto = timed(gso.replay_only, steps, warmup, world)
                    res["with_adam_ms_per_step"] = dto / steps * 1e3
                    del opt, gso
                except Exception as e:  # pragma: no cover
                    res["with_adam_ms_per_step"] = repr(e)
            if world == 1 and a.graph and not learning and a.full_outputs:
                # the same step returning the reference's index tensor as well ((P,L,4[,K]) int64, models.py:475-484): the metric's
                # step turns it off (net.return_indices = False, build_model) — its cost is reported here, never in `value`
                try:
                    net.return_indices = True
                    gsf = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3, cross_replay=True)      # one step per replay, binning on the step before
                    for _ in range(3):
                        r = gsf(xy, target, next_first=xy)
                    assert r.idx is not None and r.idx.shape[0] == P
                    nfull = max(4, steps // 4)
                    dtf = timed(gsf.replay_steady, nfull, 2, world)
                    res["with_full_outputs_ms_per_step"] = dtf / nfull * 1e3
                    res["full_outputs_note"] = f"return_indices=True: indices {tuple(r.idx.shape)} int64 materialised every step"
                    del gsf, r
                except Exception as e:  # pragma: no cover
                    res["with_full_outputs_ms_per_step"] = repr(e)
                finally:
                    net.return_indices = False
            if (head and not learning) or (learning and world == 1):
                # per-kernel launch times inside eagerly launched steps (the `roofline` objects).  At N > 1 EVERY rank runs them — the
                # eager step holds the gradient exchange, and collectives must stay matched — and rank 0 reports its own.
                try:
                    if world > 1:
                        parallel.wait_for_gradients(net)
                        parallel.defer_vertex_stage(net, False)
                    k_t, k_c = kernel_times_in_step(eager_step_fn(net, mode, xy, target, world), n=(2 if learning else 20),
                                                    warm=(0 if learning else 3))
                    if head:
                        kt, kcalls = k_t, k_c
                    if learning:
                        res["entry_ms"] = {k: v * 1e3 for k, v in sorted(k_t.items(), key=lambda kv: -kv[1])[:12]}
                        gemm_t = sum(v for k, v in k_t.items() if k in ("gngf_linear_fwd", "gngf_linear_fwd_rowstats", "gngf_gemm_acc", "gngf_linear_bwd_weight", "gngf_hpd_bwd_fused"))
                        if gemm_t > 0 and "roofline" in res:
                            fl, _ = learning_flops(res["hpd"])
                            res["roofline"]["gemm_entries_ms"] = gemm_t * 1e3
                            res["roofline"]["gemm_entries_fp32_equiv_over_fp32_peak"] = fl / gemm_t / 1e12 / MFMA_F32_PEAK_TFLOPS   # (may pass 1: bf16 pipe)
                except Exception as e:  # pragma: no cover
                    if head:
                        kt = {"error": repr(e)}
            models.should_use_hash_function = False
            if world > 1:
                parallel.defer_vertex_stage(net, False)
            del net, step
            torch.cuda.empty_cache()
        except Exception as e:
            if mode == a.mode or world > 1:
                raise
            print(f"[bench] mode {mode} failed: {e!r}", file=sys.stderr)
            results[mode] = {"error": repr(e)}
            models_mod = sys.modules.get("collision_handling_in_instantngp_amd.models")
            if models_mod is not None:
                models_mod.should_use_hash_function = False
            torch.cuda.empty_cache()

    if rank == 0:
        head = results[a.mode]
        c = SHAPES[MODES[a.mode]]
        L, F = c["L"], c["F"]
        s_fwd, s_bwd = survey_bytes(a.mode)
        c_fwd, c_bwd = compulsory_bytes(a.mode)
        times = {k: v for k, v in kt.items() if isinstance(v, float)}
        dec_flops = 2 * (L * F * 64 + 64 * 64 + 64 * 3)          # per pixel, forward; backward (dX + dW) = 2x
        # HBM bytes per launch from the round's PMC passes (profiles/traffic.json, written by tools/make_traffic.py together with
        # the signature of the step's kernel chain it was measured on).  A file measured on ANOTHER chain is not reported.
        traffic, traffic_source = {}, None
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tr_path):
            raw = json.load(open(tr_path))
            meta = raw.pop("_meta", {})
            if meta.get("chain") == ops.STEP_CHAIN_SIGNATURE:
                traffic = raw
                traffic_source = {"file": "profiles/traffic.json", **meta, "matches_running_build": True}
            else:
                traffic_source = {"file": "profiles/traffic.json", **meta, "matches_running_build": False,
                                  "running_chain": ops.STEP_CHAIN_SIGNATURE,
                                  "note": "measured on another kernel chain than the one this build runs: traffic not reported"}
                print(f"[bench] profiles/traffic.json was measured on chain {meta.get('chain')!r}, this build runs "
                      f"{ops.STEP_CHAIN_SIGNATURE!r}: roofline.traffic dropped", file=sys.stderr)

        def pmc(name):
            return (traffic.get(name) or {}).get("hbm_bytes_per_launch")

        def roof_of(name):
            t = times[name]
            base = name.split(":")[0]
            if base in ("encode_fwd", "encode_bwd"):
                fwd = base == "encode_fwd"
                per_px = c_fwd if fwd else c_bwd
                ach = per_px * P / t / 1e9
                return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "frac_of_measured_copy": (ach / hbm_copy) if hbm_copy else None,
                        "traffic": pmc(name), "avg_launch_ms": t * 1e3,
                        "compulsory_bytes_per_pixel": per_px, "pixels_per_launch": P,
                        "algorithmic_survey_bytes": (s_fwd if fwd else s_bwd) * P,
                        "note": "achieved/frac use the bytes this implementation must move per launch (binned pixel record + one "
                                "enc / d-enc row per pixel); SURVEY §8(d)'s per-instance figure (zero reuse, one table gather per "
                                "pixel-corner) is kept as algorithmic_survey_bytes: the per-vertex de-duplication removes that traffic"}
            if base in ("decoder_fwd", "decoder_bwd", "decoder_train"):
                fl = dec_flops * {"decoder_fwd": 1, "decoder_bwd": 2, "decoder_train": 3}[base]
                ach = fl * P / t / 1e12
                notes = {"decoder_fwd": "exact fp32 on v_mfma_f32_32x32x2_f32",
                         "decoder_bwd": "fp32-accurate arithmetic priced against the dense fp32 MFMA peak; at 32 input features the backward "
                                        "kernel runs two of its six products on the bf16 pipe with an exact three-way split (DESIGN.md section 3)",
                         "decoder_train": "forward + MSE gradient + backward of the decoder in ONE launch (gngf_decoder_train: the hidden layers stay "
                                          "in registers); fp32-accurate arithmetic priced against the dense fp32 MFMA peak — the forward layers, dh1 and "
                                          "d enc run on the bf16 pipe with an exact three-way split, the weight gradients on the fp32 pipe "
                                          "(DESIGN.md section 3)"}
                return {"bound": "mfma", "kernel": name, "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": pmc(name),
                        # matrix pipes busy (SQ_VALU_MFMA_BUSY_CYCLES / 4 SQ_BUSY_CU_CYCLES, the round's SQ pass through
                        # profiles/traffic.json): `frac` prices fp32-ACCURATE arithmetic, part of which runs on the 16x faster bf16
                        # pipe — it is not "the matrix pipes are frac busy"
                        "mfma_busy": (traffic.get(name) or {}).get("mfma_busy"),
                        "avg_launch_ms": t * 1e3, "algorithmic_flops_per_pixel": fl, "pixels_per_launch": P, "note": notes[base]}
            return None

        roof = roof_enc = None
        ranked = sorted((k for k in times if roof_of(k) is not None), key=times.get, reverse=True)
        if ranked:
            roof = roof_of(ranked[0])                             # dominant kernel of the step
            enc_k = [k for k in ranked if k.startswith("encode_")]
            if enc_k:
                roof_enc = roof_of(enc_k[0])                      # dominant ENCODER kernel (the HBM-bound part of the path)
        step_s = head["ms_per_step"] * 1e-3
        step_flop = 3 * dec_flops * P
        step_bytes = sum(v for v in (pmc(k) for k in traffic) if v)
        roof_step = {"flop_per_step": step_flop, "tflops": step_flop / step_s / 1e12 * world / max(world, 1),
                     "mfma_frac": step_flop / step_s / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     "hbm_bytes_per_step_pmc": step_bytes or None,
                     "hbm_frac": (step_bytes / step_s / 1e9 / HBM_PEAK_GBS) if step_bytes else None,
                     "survey_bytes_per_step": (s_fwd + s_bwd) * P,
                     "note": "per GPU; FLOP = decoder fwd + bwd (the encoder has no MFMA work); bytes = sum of the per-kernel PMC traffic "
                             "in profiles/traffic.json (measured on the round's profiled run of this same command)"}
        line = {
            "metric": "Mpixels/sec fwd+bwd at L=16,F=2,T=2^19", "value": head["mpix_s"], "unit": "Mpixel/s",
            "n_gpus": world, "steps": head["steps"], "warmup": head["warmup"], "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": (f"{c['image']} image's own pixel list (339x508, tests/golden/{c['image']}_rgb.npz) shuffled and repeated to the batch size; "
                     "random-init weights" if MODES[a.mode] in ("cfg2", "cfg3") else "synthetic (uniform-random coordinates and RGB; random-init weights)"),
            "config": {"workload": f"{MODES[a.mode]}: " + (f"{c['image']} image (339x508) pixel list shuffled+repeated to 2^20 px/GPU, L=16 F=2 T=2^19 K=4 N 16->512"
                                                          if MODES[a.mode] in ("cfg2", "cfg3") else f"synthetic {c['image']}^2 image, L={L} F={F} T={c['T']} N {c['n_min']}->{c['n_max']}")
                                   + f", {a.mode} indexing, random-init weights, MSE loss, fwd+bwd (no optimizer)",
                       "mode": a.mode, "pixels_per_gpu": P, "parallelism": f"dp{world}",
                       "untimed_ramp_steps_before_warmup": a.ramp_steps, "launch": head["launch"], "step_config": head.get("step_config")},
            "collective_ranks": collective_ranks, "rccl_ranks": (collective_ranks if (world > 1 and a.backend == "nccl") else None),
            "backend": (a.backend if world > 1 else None),
            "modes": results, "kernel_ms": {k: (v * 1e3 if isinstance(v, float) else v) for k, v in kt.items()},
            "roofline": roof, "roofline_encoder": roof_enc, "roofline_step": roof_step, "traffic_source": traffic_source,
            "ms_per_step_windows": head.get("ms_per_step_windows"), "ms_per_step_per_rank": head.get("ms_per_step_per_rank"),
            "hbm_copy_measured_GBs": hbm_copy, "hbm_peak_GBs": HBM_PEAK_GBS,
            "roofline_survey": {m: r_["roofline_survey"]["frac_of_8TBs"] for m, r_ in results.items() if "roofline_survey" in r_},
            "roofline_survey_note": "SURVEY 8(d) algorithmic bytes/pixel x pixels/s / 8 TB/s per mode (the target's '>= 60 % of the HBM-read "
                                    "roofline' is this number for the hash modes); per-mode detail incl. the fraction of the measured copy "
                                    "bandwidth under modes.<mode>.roofline_survey.  GNGF modes exceed 1 by construction (see DESIGN.md section 5)",
        }
        if roof is not None and roof.get("kernel") in ("decoder_bwd", "decoder_train") and in_graph_ms:
            # same kernel, timed by its own device-clock stamps inside the replayed graph (agrees with rocprofv3's average)
            roof["in_graph_launch_ms"] = in_graph_ms
            roof["in_graph_frac"] = roof["algorithmic_flops_per_pixel"] * P / (in_graph_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.mode, a.cpu_sample)
        assert line["n_gpus"] == a.gpus
        print(json.dumps(line))
        # a fraction of a peak above 1 is a broken measurement, not a result: show it (nothing is clamped) and fail the run
        fr = {"roofline.frac": (roof or {}).get("frac"), "roofline.in_graph_frac": (roof or {}).get("in_graph_frac"),
              "roofline_encoder.frac": (roof_enc or {}).get("frac"), "roofline_step.mfma_frac": roof_step.get("mfma_frac"),
              "roofline_step.hbm_frac": roof_step.get("hbm_frac")}
        for m, r_ in results.items():
            fr[f"modes.{m}.roofline.frac"] = (r_.get("roofline") or {}).get("frac")
            fr[f"modes.{m}.roofline.bf16_pipe.frac"] = ((r_.get("roofline") or {}).get("bf16_pipe") or {}).get("frac")
            if is_hash(m) and "roofline_survey" in r_:
                fr[f"modes.{m}.roofline_survey.frac_of_8TBs"] = r_["roofline_survey"]["frac_of_8TBs"]
        bad = {k: v for k, v in fr.items() if isinstance(v, float) and v > 1.05}
        if bad:
            print(f"bench.py: fraction(s) of a hardware peak above 1.05 — the measurement is broken: {bad}", file=sys.stderr)
            broken = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if broken:
        sys.exit(4)


if __name__ == "__main__":
    main()
