"""Data-parallel training over RCCL/xGMI (one process per GPU): pixels shard across ranks, parameters replicate,
one gradient exchange per step (the reference has no counterpart).

Exchange layout, sized for point-to-point xGMI links:
  * the hash-table gradient is ONE contiguous (L,T,F) buffer (`encoding._grad_base`, see ops.TableViewFunction):
    a single large all-reduce, no per-level launches, no flatten copy;
  * every other gradient (decoder, HPD) is packed into one small flat bucket.
Losses are means over the LOCAL batch, so summed gradients are divided by world size (equal shards)."""
import torch
import torch.distributed as dist

from . import ops


_COALESCED_AVG = True


def all_reduce_sum(t, group=None):
    """Sum-all-reduce of a device tensor: RCCL when the backend is nccl; with gloo (CPU rehearsals / single-GPU tests
    of the multi-rank logic) the tensor is staged through host memory."""
    if dist.get_backend(group) == "nccl" or not t.is_cuda:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)


def enable_vertex_grid_exchange(net, world: int, group=None):
    """Sets up `net.dp` (ops.DataParallel: the data-parallel state of THIS model).  Exchange the dense per-level VERTEX-GRID
    gradient (sum_l (N_l+2)^2 * F floats: 5.7 MB at N 16->512) inside the
    encoder backward instead of the (L,T,F) table gradient (64 MiB at T = 2^19) afterwards: the vertex stage
    dG -> dE is linear and identical on every rank, so reducing dG first gives the same table gradient with ~11x
    fewer bytes on the xGMI links.  Levels that run in the direct form (the finest levels of very fine grids) keep the
    table-gradient exchange, restricted to their own contiguous slice of the (L,T,F) buffer."""
    dp = net.dp
    dp.tables_reduced = 0              # bookkeeping of a previous configuration does not carry over
    dp.deferred = None
    dp.world, dp.group = int(world), group
    if world <= 1:
        dp.exchange = dp.mean = dp.max_ = None
        return
    dp.exchange = lambda t: average_tensors([t], world, group)
    # GNGF with a trainable HPD: the KL/JS loss is a nonlinear function of the batch-mean distribution, so p-bar is
    # averaged over ranks in the forward (exact single-GPU equivalence), and every rank builds its per-vertex table over
    # the same vertex rectangle (max of the shards' coordinate bounds).
    dp.mean = dp.exchange

    def _max(t):
        if dist.get_backend(group) == "nccl" or not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        else:
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
            t.copy_(h)
    dp.max_ = _max


def shard_batch(n_items: int, rank: int, world: int):
    """Contiguous equal shards (the last rank takes the remainder)."""
    per = n_items // world
    lo = rank * per
    hi = n_items if rank == world - 1 else lo + per
    return lo, hi


def _flat_alias(grads):
    """One flat tensor aliasing `grads` when they are consecutive contiguous slices of a single storage (the fused
    decoder backward lays its six gradients out that way), else None."""
    if not grads:
        return None
    g0 = grads[0]
    base, off = g0.untyped_storage().data_ptr(), g0.storage_offset()
    for g in grads:
        if g.untyped_storage().data_ptr() != base or g.storage_offset() != off or not g.is_contiguous() or g.dtype != g0.dtype:
            return None
        off += g.numel()
    return torch.as_strided(g0, (off - g0.storage_offset(),), (1,), g0.storage_offset())


def defer_vertex_stage(net, on: bool = True):
    """Keep collectives OUT of the backward pass: the encoder backward stops after the pixel stage, and
    allreduce_gradients() / finish_backward() exchange dG and run the vertex stage afterwards.  With it the whole
    forward + backward is collective-free and can be replayed from one hipGraph on every rank.  Applies to index sources
    without trainable per-vertex weights (hash, frozen HPD); a trainable HPD keeps the in-backward exchange."""
    dp = net.dp
    dp.defer_vertex = bool(on)
    if not on:
        dp.deferred = None
        dp.tables_reduced = 0


def average_tensors(tensors, world: int, group=None):
    """In-place mean over ranks of several device tensors as ONE exchange.  RCCL: the all-reduces are issued inside a
    coalescing group (one fused launch instead of one per tensor; a small collective costs tens of microseconds of
    latency whatever its size) with the AVG reduction (no separate scaling kernels).  gloo (CPU rehearsals, ranks
    sharing a GPU in tests): one sum-all-reduce per tensor, staged through the host, then scaled."""
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    if not tensors or world <= 1:
        return
    global _COALESCED_AVG
    if _COALESCED_AVG and dist.get_backend(group) == "nccl" and all(t.is_cuda for t in tensors):
        try:
            if len(tensors) == 1:
                dist.all_reduce(tensors[0], op=dist.ReduceOp.AVG, group=group)
                return
            with dist._coalescing_manager(group=group, device=tensors[0].device, async_ops=True) as cm:
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
            cm.wait()
            return
        except (RuntimeError, TypeError, AttributeError, ValueError) as e:
            # an RCCL / torch build without AVG or the coalescing API rejects the call before anything is enqueued, and
            # it does so on every rank alike: fall back (for good) to one sum-all-reduce per tensor
            import sys
            print(f"[parallel] coalesced AVG exchange unavailable ({e!r}); using per-tensor all-reduce", file=sys.stderr)
            _COALESCED_AVG = False
    for t in tensors:
        all_reduce_sum(t, group)
        t.mul_(1.0 / world)


def wait_for_gradients(net):
    """Makes the current stream wait for an exchange started with allreduce_gradients(..., overlap=True): call it before
    anything reads the gradients (optimizer.step()) or overwrites the exchanged buffers (the next backward pass; for a step
    replayed from a hipGraph, the next replay).  No-op when nothing is in flight.  The host does not block."""
    dp = getattr(net, "dp", None)
    if dp is not None and dp.comm_done is not None:
        torch.cuda.current_stream().wait_event(dp.comm_done)
        dp.comm_done = None


def allreduce_gradients(net, world: int, group=None, keep_tables_flag: bool = False, overlap: bool = False):
    """All gradient traffic of one step, as one coalesced exchange: the deferred vertex-grid gradient (if the encoder
    backward left one, see defer_vertex_stage), the table-gradient slice of the direct-form levels, and every other
    gradient as one flat buffer; the deferred vertex stage runs behind it.  keep_tables_flag: the step is replayed from
    a hipGraph, so the staged-level count recorded at capture time stays valid for every replay.
    overlap: the exchange and the deferred vertex stage are issued on the model's own communication stream (behind everything
    already on the current stream) and the call returns at once: whatever the caller launches next that does not touch the
    gradients — the next batch's host-side preparation, its upload, its binning (ops.TiledWorkspace on the next coordinates) —
    runs beside the RCCL transfer; wait_for_gradients(net) joins.  (The next step's forward itself cannot run ahead: it reads
    the weights the optimizer is about to update with these gradients.)"""
    if world <= 1:
        return
    if overlap and torch.cuda.is_available():
        dp = getattr(net, "dp", None)
        if dp is not None:
            wait_for_gradients(net)                  # (an earlier exchange nobody joined)
            if dp.comm_stream is None:
                dp.comm_stream = torch.cuda.Stream()
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(dp.comm_stream):
                dp.comm_stream.wait_event(ready)
                allreduce_gradients(net, world, group, keep_tables_flag, overlap=False)
                done = torch.cuda.Event()
                done.record()
            dp.comm_done = done
            return
    tensors = []
    dp = getattr(net, "dp", None)                # (a plain nn.Module without the encoder's state: dense exchange only)
    deferred = dp.deferred if dp is not None else None
    if deferred is not None:
        tensors.append(deferred[6])              # dG of the staged levels
    handled = set()
    enc = getattr(net, "encoding", None)
    base = getattr(enc, "_grad_base", None) if enc is not None else None
    if base is None and enc is not None and getattr(enc, "_grad_base_fp32", None) is not None \
            and any(getattr(m.weight, "grad_fp32", None) is not None for m in enc._hash_tables):
        base = enc._grad_base_fp32            # fp16 tables with ops.FP16_TABLE_GRAD_FP32: the fp32 buffer IS the gradient
    tables_done = int(dp.tables_reduced) if dp is not None else 0     # leading levels already reduced through dG by the encoder backward
    if dp is not None and not keep_tables_flag:
        dp.tables_reduced = 0
    zero = getattr(dp, "zero", None) if dp is not None else None
    scatter = None
    if base is not None:
        handled = {id(m.weight) for m in enc._hash_tables}
        if tables_done < base.shape[0]:
            if zero and zero["Ls"] == tables_done and base.is_contiguous():
                scatter = base[tables_done:].reshape(-1)          # sharded update: reduce-scatter instead of all-reduce (below)
            else:
                tensors.append(base[tables_done:])   # (L,T,F): the direct-form levels only
    rest = [p for p in net.parameters() if p.requires_grad and p.grad is not None and id(p) not in handled]
    flat = None
    in_place = False
    if rest:
        flat = _flat_alias([p.grad for p in rest])
        in_place = flat is not None
        if not in_place:
            flat = torch.cat([p.grad.reshape(-1) for p in rest])
        tensors.append(flat)
    average_tensors(tensors, world, group)
    if scatter is not None:
        _reduce_scatter_mean(scatter, zero, world, group)
    if deferred is not None:
        ops.run_deferred_vertex_stage(dp, exchanged=True)
    if flat is not None and not in_place:
        off = 0
        for p in rest:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


# ------------------------------------------------------------------------------------------------ sharded update of the direct levels
def shard_direct_levels(net, world: int, rank: int, pixels_per_rank: int, group=None):
    """Round 5 — the answer to "exchange ~ step" at BASELINE config 5 (8 GPUs, F = 4, T = 2^24, fp16 tables:
    the four direct levels' slice of the fp32 table gradient is 1 GiB, its ring all-reduce ~1.75 ms beside a ~1.75 ms step, and
    the dense Adam over all 2 GiB of tables 5.5 ms on EVERY rank).  ZeRO-1 restricted to the levels that need it:

      * the staged levels stay as they are (their gradient travels as the small vertex-grid gradient and every rank updates them);
      * the direct levels' gradient slice is REDUCE-SCATTERED (each rank receives the mean of its 1/n of the rows: (n-1)/n x S on the
        links instead of 2 (n-1)/n x S), each rank's optimizer updates only those rows (`_adam_range` on the level parameters:
        train.FusedAdam offsets its segment records; moments and the fp32 master copy of the other rows are never touched), and
        the updated PARAMETER rows are all-gathered (fp16 storage: half the bytes of the gradient) — gather_direct_levels(),
        after optimizer.step().
    Bytes per rank and step at config 5, n = 8: 0.875 x 1 GiB + 0.875 x 0.5 GiB = 1.31 GiB instead of 1.75 GiB, and the Adam launch
    covers 1/8 of the direct levels (+ the staged ones) instead of all of them.  Which levels are direct follows from the plan of
    a `pixels_per_rank` batch (ops.EncodePlan), as in the encoder.  Returns (first direct level, lo, hi): this rank's element range
    inside the flat direct slice.  Undo: shard_direct_levels(net, 1, 0, pixels_per_rank)."""
    enc = net.encoding
    L, T, F = enc.packed_tables().shape
    Ls = ops.EncodePlan(int(pixels_per_rank), net._n_ls_host, F).Ls
    dp = net.dp
    levels = list(enc._hash_tables)
    for m in levels:
        m.weight._adam_range = None
    dp.zero = None
    n_d = (L - Ls) * T * F
    if world <= 1 or n_d == 0:
        return Ls, 0, n_d
    per = -(-n_d // world)
    per = -(-per // 1024) * 1024                                  # whole 4 KiB blocks: every shard starts 16-byte aligned in every dtype
    lo, hi = min(n_d, rank * per), min(n_d, (rank + 1) * per)
    for l in range(Ls, L):
        a, b = (l - Ls) * T * F, (l - Ls + 1) * T * F             # this level inside the flat direct slice
        levels[l].weight._adam_range = (max(lo, a) - a, max(min(hi, b), max(lo, a)) - a)
    dp.zero = {"Ls": Ls, "lo": lo, "hi": hi, "per": per, "n": n_d, "world": int(world), "rank": int(rank), "group": group}
    return Ls, lo, hi


def _reduce_scatter_mean(flat, z, world, group):
    """mean over the ranks of `flat` (the direct levels' gradient slice), delivered to this rank's [lo, hi) only"""
    if dist.get_backend(group) == "nccl" and flat.is_cuda and z["per"] * world == z["n"]:
        dist.reduce_scatter_tensor(flat[z["lo"]:z["hi"]], flat, op=dist.ReduceOp.AVG, group=group)     # in place: RCCL's own layout
        return
    # gloo rehearsal (ranks sharing a GPU in tests), or a slice that does not divide evenly: the same values by all-reduce
    all_reduce_sum(flat, group)
    flat.mul_(1.0 / world)


def gather_direct_levels(net, group=None):
    """after optimizer.step() with shard_direct_levels(): every rank receives the parameter rows the others updated"""
    z = getattr(net.dp, "zero", None)
    if not z:
        return
    world = z["world"]
    flat = net.encoding.packed_tables()[z["Ls"]:].reshape(-1)
    with torch.no_grad():
        if dist.get_backend(group) == "nccl" and flat.is_cuda and z["per"] * world == z["n"]:
            dist.all_gather_into_tensor(flat, flat[z["lo"]:z["hi"]].clone(), group=group)
            return
        mine = flat[z["lo"]:z["hi"]].detach()
        pad = torch.zeros((z["per"],), dtype=flat.dtype, device="cpu")
        pad[:mine.numel()] = mine.cpu()
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        for r, part in enumerate(parts):
            a, b = min(z["n"], r * z["per"]), min(z["n"], (r + 1) * z["per"])
            if b > a and r != z["rank"]:
                flat[a:b].copy_(part[:b - a].to(flat.device))


def broadcast_parameters(net, src: int = 0, group=None):
    """Replicas must start identical (same seed does it too; this makes it explicit)."""
    for p in net.parameters():
        dist.broadcast(p.data, src=src, group=group)
    for b in net.buffers():
        dist.broadcast(b.data, src=src, group=group)
