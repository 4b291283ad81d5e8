#!/usr/bin/env python3
"""Headline benchmark: Mpixel/s forward+backward of the hash-grid encoder path (coords -> encoder -> decoder ->
MSE gradient -> decoder/table gradients) at L=16, F=2, T=2^19, batch 2^20 pixels per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W            # N>1: launched by torch.distributed.run, one rank per GPU

One JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed on the
launch stream) and `cpu_baseline` (CPU oracle timed on the host cores, bounded sample, rank 0, N=1 only).

Modes (all at the same shape; --mode picks the headline, the others are reported in the same line under "modes"):
  gngf_frozen    GNGF indexing with a frozen HashProbDistribution (the reference's -hwp mode, models.py:364-371):
                 the per-vertex top-K table is a function of frozen weights and is rebuilt only when they change.
  gngf_learning  GNGF indexing with a trainable HPD: every step re-evaluates the HPD on every distinct vertex
                 (MFMA-bound: U x 128 x T contraction) and back-propagates into it.
  hash           plain spatial-hash indexing (should_use_hash_function=True), BASELINE.json configs[2] shape.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L, F, T, K_TOP = 16, 2, 2 ** 19, 4
N_MIN, N_MAX = 16, 512
P_PER_GPU = 2 ** 20
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)


def algorithmic_bytes(mode):
    """algorithmic bytes per pixel (fp32, ~0 reuse assumed)."""
    sf = 4
    Ke = 1 if mode == "hash" else K_TOP
    J = 0 if mode == "hash" else 8 * K_TOP
    b_fwd = 8 + L * 4 * Ke * F * sf + L * 4 * J + L * F * 4
    b_bwd = 8 + L * F * 4 + L * 4 * J + 2 * L * 4 * Ke * F * sf + (L * 4 * K_TOP * F * sf if mode != "hash" else 0)
    return b_fwd, b_bwd


def strawberry_batch(P, rank, dev):
    """cfg2 input: the 339x508 strawberry pixel list (coords = (row,col)/507, main.py:50-51), shuffled and repeated
    to P pixels (the image has only 172 212 pixels)."""
    img = np.load(os.path.join(ROOT, "tests", "golden", "strawberry_rgb.npz"))["img"]
    h, w = img.shape[:2]
    g = torch.Generator().manual_seed(65535 + rank)
    sel = torch.cat([torch.randperm(h * w, generator=g) for _ in range(-(-P // (h * w)))])[:P]
    rows, cols = sel // w, sel % w
    xy = torch.stack([rows, cols], 1).float() / (max(w, h) - 1)
    rgb = torch.from_numpy(img.reshape(-1, 3))[sel].float() / 255
    return xy.to(dev).contiguous(), rgb.to(dev).contiguous(), (h, w)


def build_model(mode, dev):
    from collision_handling_in_instantngp_amd import models
    models.should_use_hash_function = (mode == "hash")
    torch.manual_seed(65535)
    net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=N_MIN, n_max=N_MAX,
                                          MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                          HPD_out_features=T, feature_dim=F, topk_k=K_TOP).to(dev)
    net.return_indices = False          # the (P,L,4,K) int64 diagnostics tensor (2 GiB/step) is not part of fwd+bwd
    net.dense_probs = False             # the dense (P,L,4,T) tensor is 128 MiB *per pixel* at this shape
    if mode == "gngf_frozen":
        for p in net.HPD.parameters():
            p.requires_grad = False
        net.compute_pbar = False        # the KL/JS term has no trainable input when the HPD is frozen
    if mode == "gngf_learning":
        net.coord_bounds = (1.0, 338.0 / 507.0)
    return net, models


def make_step(net, models, mode, xy, target, world, exchange=True):
    """One training step: forward, loss, backward (+ the gradient exchange when world > 1 and `exchange`)."""
    from collision_handling_in_instantngp_amd import train, parallel
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    params = [p for p in net.parameters() if p.requires_grad]
    empty = torch.tensor([], device=xy.device)
    one = torch.ones((), device=xy.device)          # the seed of backward(): loss.backward() would fill a fresh one per step

    def step():
        for p in params:
            p.grad = None
        rgb, probs, _idx, _c = net(xy, 1.0)
        if mode == "gngf_learning":
            mse, kls, coll = loss_fn(rgb, target, T, probs, empty, empty)
            loss = train.assemble_loss(mse, kls, coll, 1, 1, 1e-3)
        else:
            loss = loss_fn._mse(rgb, target)      # frozen HPD / hash: the KL-JS and collision terms carry no gradient
        loss.backward(gradient=one)
        if world > 1 and exchange:
            parallel.allreduce_gradients(net, world)
    return step


def graphed(step, warm=3):
    """Capture one whole step (forward + backward, every launch of ours and torch's) into a hipGraph: the step is
    ~15 short kernels, so eager launch gaps and Python dispatch are a visible fraction of a ~1 ms step."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    # thread_local: other threads of the process (the RCCL watchdog at world > 1) may query events while we capture
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step()
    return g.replay


def timed(step, steps, warmup, world):
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
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    return dt


ENTRY_NAMES = {"gngf_bin_pixels": "bin_pixels", "gngf_vertex_grid_fwd": "vertex_fwd", "gngf_encode_tiled_fwd": "encode_fwd:tiled",
               "gngf_encode_tiled_bwd": "encode_bwd:tiled", "gngf_vertex_grid_bwd_sorted": "vertex_bwd", "gngf_vertex_grid_bwd": "vertex_bwd",
               "gngf_decoder_fwd": "decoder_fwd", "gngf_decoder_bwd": "decoder_bwd", "gngf_decoder_reduce": "decoder_reduce", "gngf_mse_fwd": "mse_fwd", "gngf_mse_bwd": "mse_bwd",
               "gngf_encode_fwd": "encode_fwd:direct", "gngf_encode_bwd": "encode_bwd:direct"}


def kernel_times_in_step(eager_step, n=20, warm=3):
    """Average launch duration of every C-ABI entry point INSIDE the training step: HIP events recorded on the stream
    each kernel is launched on (torch's current stream at the call; the helper stream for the vertex stage), bracketing
    every call of n eagerly launched steps.  `decoder_bwd` = decoder_bwd_kernel + decoder_reduce_kernel,
    `encode_bwd:tiled` = tiled_bwd_kernel + gather_partials_kernel (one entry point each)."""
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
    out = {}
    for name, pairs in prof.items():
        key = ENTRY_NAMES.get(name, name)
        ms = [a.elapsed_time(b) for a, b in pairs]
        out[key] = out.get(key, 0.0) + sum(ms) / n * 1e-3          # seconds per step (entry points called once per step)
    return out


def kernel_times(net, models, mode, xy, n=20):
    """Fallback: HIP-event timing (on the launch stream = torch's current stream) of each hot kernel launched alone."""
    from collision_handling_in_instantngp_amd import ops
    dev = xy.device
    n_ls = net._n_ls_flat(dev)
    tables = net.encoding.packed_tables()
    P = xy.shape[0]
    if mode == "hash":
        ti = w = None
        vstride = 0
    else:
        with torch.no_grad():
            _tv, ti, w, vstride, _NV, _o = net._frozen_vertex_table(0) if net.hpd_is_frozen() else (None,) * 6
        if ti is None:
            return {}
    genc = torch.randn((P, L * F), device=dev)
    out = {}

    def ev(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    for name, fn in ops.encode_kernels(xy, n_ls, net._n_ls_host, tables, ti, w, vstride, genc).items():
        out[name] = ev(fn)
    if ops.decoder_fused_ok((ops.ACT_RELU, ops.ACT_RELU, ops.ACT_SIGMOID), net._decoder_params()):
        drgb = torch.randn((P, 3), device=dev) / P
        for name, fn in ops.decoder_kernels(genc, net._decoder_params(), False, drgb).items():
            out[name] = ev(fn)
    return out


def cpu_baseline(mode, sample_pixels):
    """The CPU oracle timed on this host's cores (kind "port"): forward + backward of the same path (encoder with the
    index table given, decoder, MSE gradient) on a bounded sample of the same workload.  Uses the C/OpenMP
    restatement (oracle/gngf_oracle_c.c) when built, else the numpy one.  Checker code is only timed here."""
    from oracle import gngf_oracle as orc, c_oracle
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
    if mode != "hash":
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
            if mode == "hash":
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
            "sample": f"{n} x {sample_pixels} strawberry pixels, fwd+bwd (encoder + decoder + MSE grad), {mode} indexing with the "
                      f"index table given, {impl}, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="gngf_frozen", choices=["gngf_frozen", "gngf_learning", "hash"])
    ap.add_argument("--pixels", type=int, default=P_PER_GPU, help="pixels per GPU per step")
    ap.add_argument("--no-extra-modes", action="store_true")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2 ** 20)
    ap.add_argument("--ramp-steps", type=int, default=60, help="untimed steps before the W warm-up steps (clock ramp)")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: 'gloo' runs the N>1 code path with ranks sharing one GPU")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.cuda.set_device(0)
            dist.init_process_group(a.backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        from collision_handling_in_instantngp_amd import parallel
        parallel.enable_vertex_grid_exchange(world)
    P = a.pixels
    xy, target, (h, w) = strawberry_batch(P, rank, dev)

    results = {}
    in_graph_ms = None
    order = [a.mode] + ([m for m in ("gngf_learning", "hash", "gngf_frozen") if m != a.mode] if not a.no_extra_modes else [])
    kt = {}
    for mode in order:
        head = mode == a.mode
        steps, warmup = (a.steps, a.warmup) if head else ((2, 1) if mode == "gngf_learning" else (max(5, a.steps // 2), 2))
        if mode == "gngf_learning" and head:
            steps, warmup = min(a.steps, 5), min(a.warmup, 1)
        net, models = build_model(mode, dev)
        step = make_step(net, models, mode, xy, target, world)
        launch = "eager"
        if a.graph and mode != "gngf_learning":
            # world > 1: the vertex stage of the encoder backward is deferred behind the dG exchange, so forward + backward
            # hold no collective and replay from one hipGraph; the exchange (RCCL) and the vertex stage follow eagerly.
            try:
                if world > 1:
                    from collision_handling_in_instantngp_amd import parallel
                    parallel.defer_vertex_stage(True)
                    replay = graphed(make_step(net, models, mode, xy, target, world, exchange=False))

                    def step(replay=replay, net=net):
                        replay()
                        parallel.allreduce_gradients(net, world, keep_tables_flag=True)
                    launch = "hipGraph + eager exchange"
                else:
                    step = graphed(step)
                    launch = "hipGraph"
            except Exception as e:  # pragma: no cover
                print(f"[bench] graph capture failed ({e!r}); running eagerly", file=sys.stderr)
                if world > 1:
                    parallel.defer_vertex_stage(False)
                step = make_step(net, models, mode, xy, target, world)
        if mode != "gngf_learning":
            # Clock ramp: the chip reaches its steady matrix-core clock only after ~15 ms of sustained work (the same kernel
            # is ~10 % slower before; tools/perf_decoder_warm.py), and W warm-up steps of 0.65 ms do not get there.  Untimed
            # steps first (every rank runs the same number, so collectives stay matched), then the W + K of the contract.
            for _ in range(a.ramp_steps):
                step()
        dt = timed(step, steps, warmup, world)
        results[mode] = {"mpix_s": P * world * steps / dt / 1e6, "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
                         "launch": launch}
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
        if head and world == 1 and a.graph and mode != "gngf_learning":
            # the same step with the optimizer in the graph (get_optimizer's Adam as one launch): reported, not the metric
            try:
                from collision_handling_in_instantngp_amd import train
                opt = train.get_optimizer(net, 1e-2, 1e-3, 1e-3, 0.0, 0.0, 1e-6)
                plain = make_step(net, models, mode, xy, target, world)

                def with_opt(plain=plain, opt=opt):
                    plain()
                    opt.step()
                dto = timed(graphed(with_opt), steps, warmup, world)
                results[mode]["with_adam_ms_per_step"] = dto / steps * 1e3
                del opt
            except Exception as e:  # pragma: no cover
                results[mode]["with_adam_ms_per_step"] = repr(e)
        if head and rank == 0 and world == 1:
            try:
                kt = kernel_times_in_step(make_step(net, models, mode, xy, target, world), n=(3 if mode == "gngf_learning" else 20),
                                          warm=(0 if mode == "gngf_learning" else 3))
            except Exception as e:  # pragma: no cover
                kt = {"error": repr(e)}
        elif head and rank == 0:
            try:
                kt = kernel_times(net, models, mode, xy)
            except Exception as e:  # pragma: no cover
                kt = {"error": repr(e)}
        models.should_use_hash_function = False
        if world > 1:
            from collision_handling_in_instantngp_amd import parallel as _par
            _par.defer_vertex_stage(False)
        del net, step
        torch.cuda.empty_cache()

    if rank == 0:
        head = results[a.mode]
        b_fwd, b_bwd = algorithmic_bytes(a.mode)
        times = {k: v for k, v in kt.items() if isinstance(v, float)}
        dec_flops = 2 * (L * F * 64 + 64 * 64 + 64 * 3)          # per pixel, forward; backward (dX + dW) = 2x
        traffic = {}
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tr_path):
            traffic = json.load(open(tr_path))

        def roof_of(name):
            t = times[name]
            base = name.split(":")[0]
            if base in ("encode_fwd", "encode_bwd"):
                per_px = b_fwd if base == "encode_fwd" else b_bwd
                ach = per_px * P / t / 1e9
                return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": (traffic.get(name) or {}).get("hbm_bytes_per_launch"),
                        "avg_launch_ms": t * 1e3, "algorithmic_bytes_per_pixel": per_px, "pixels_per_launch": P}
            if base in ("decoder_fwd", "decoder_bwd"):
                fl = dec_flops * (1 if base == "decoder_fwd" else 2)
                ach = fl * P / t / 1e12
                return {"bound": "mfma", "kernel": name, "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": (traffic.get(name) or {}).get("hbm_bytes_per_launch"),
                        "avg_launch_ms": t * 1e3, "algorithmic_flops_per_pixel": fl, "pixels_per_launch": P}
            return None

        roof = roof_enc = None
        ranked = sorted((k for k in times if roof_of(k) is not None), key=times.get, reverse=True)
        if ranked:
            roof = roof_of(ranked[0])                             # dominant kernel of the step
            enc_k = [k for k in ranked if k.startswith("encode_")]
            if enc_k:
                roof_enc = roof_of(enc_k[0])                      # dominant ENCODER kernel (the HBM-bound part of the path)
        line = {
            "metric": "Mpixels/sec fwd+bwd at L=16,F=2,T=2^19", "value": head["mpix_s"], "unit": "Mpixel/s",
            "n_gpus": world, "steps": head["steps"], "warmup": head["warmup"], "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2: strawberry.jpeg 339x508 pixel list shuffled+repeated to 2^20 px/GPU, L=16 F=2 T=2^19 "
                                   f"K=4 N 16->512, {a.mode} indexing, random-init weights, MSE loss, fwd+bwd (no optimizer)",
                       "mode": a.mode, "pixels_per_gpu": P, "parallelism": f"dp{world}",
                       "untimed_ramp_steps_before_warmup": a.ramp_steps},
            "modes": results, "kernel_ms": {k: (v * 1e3 if isinstance(v, float) else v) for k, v in kt.items()},
            "roofline": roof, "roofline_encoder": roof_enc,
        }
        if roof is not None and roof.get("kernel") == "decoder_bwd" and in_graph_ms:
            # same kernel, timed by its own device-clock stamps inside the replayed graph (agrees with rocprofv3's average)
            roof["in_graph_launch_ms"] = in_graph_ms
            roof["in_graph_frac"] = roof["algorithmic_flops_per_pixel"] * P / (in_graph_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.mode, a.cpu_sample)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
