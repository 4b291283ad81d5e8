"""GPU: round 4's step structure — the vertex stage forward fused into the pixel stage's staging loop (gngf_encode_tiled_fwd_fused)
and the binning of the NEXT batch riding on the current step's pixel-stage launches (ops.BinPipeline: count half in the forward
launch, scatter half as tasks in the tail of the backward launch).  Neither may change a result:
  * fused forward == vertex stage + pixel stage, bit for bit (both index sources, K = 4 and K != 4, L = 16 and L != 16);
  * a pipelined unrolled graph == the same graph with the pipeline off (outputs and decoder gradients bit for bit; table gradients
    bit for bit up to the order of float atomics where three or more vertices share a table row);
  * every way of breaking the announcement (another tensor, an in-place edit, a forward without its backward, a shape change)
    falls back to binning at the head of the step."""
import numpy as np
import pytest
import torch

from oracle import gngf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(P, n_min, n_max, L, T, K, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    xy = torch.rand((P, 2), device=DEV, generator=g)
    xy[:6] = torch.tensor([[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.25], [0.99999994, 1e-8]], device=DEV)
    n_host = [int(v) for v in orc.level_resolutions(n_min, n_max, L)]
    n_ls = torch.tensor(n_host, dtype=torch.int32, device=DEV)
    tables = (torch.rand((L, T, 2), device=DEV, generator=g) - 0.5) * 2e-2
    vs = n_host[-1] + 2
    vidx = torch.randint(0, T, (vs * vs, K), device=DEV, dtype=torch.int32, generator=g)
    vw = torch.softmax(torch.rand((vs * vs, K), device=DEV, generator=g), -1)
    return xy, n_host, n_ls, tables, vidx, vw, vs


@pytest.mark.parametrize("shape", [(2 ** 16, 8, 32, 4, 256, 4), (2 ** 18, 16, 512, 16, 2 ** 19, 4), (50000, 16, 128, 5, 1000, 3),
                                   (2 ** 15, 16, 256, 16, 2 ** 14, 7)])
@pytest.mark.parametrize("src", ["hash", "table"])
def test_fused_vertex_forward_is_bit_identical(shape, src):
    """FUSED_VERTEX_FWD on / off: enc from the level tables directly (SRC 1 / 2 of tiled_fwd_il_kernel) equals enc through the
    vertex grid G (vertex riders + SRC 0); and both equal the direct form, which gathers per pixel."""
    from collision_handling_in_instantngp_amd import ops, _lib
    P, n_min, n_max, L, T, K = shape
    xy, n_host, n_ls, tables, vidx, vw, vs = _inputs(P, n_min, n_max, L, T, K, 11)
    vi, w, vstr = (None, None, 0) if src == "hash" else (vidx, vw, vs)
    assert ops.EncodePlan(P, n_host, 2, "tiled").interleaved(backward=False)
    prev = ops.FUSED_VERTEX_FWD
    _lib.PROFILE = {}
    try:
        ops.FUSED_VERTEX_FWD = True
        a = ops.encode_apply(xy, n_ls, n_host, tables, vi, w, vstr, path="tiled")
        assert "gngf_encode_tiled_fwd_fused" in _lib.PROFILE and "gngf_bin_pixels2" in _lib.PROFILE
        ops.FUSED_VERTEX_FWD = False
        _lib.PROFILE = {}
        b = ops.encode_apply(xy, n_ls, n_host, tables, vi, w, vstr, path="tiled")
        assert "gngf_encode_tiled_fwd" in _lib.PROFILE and "gngf_encode_tiled_fwd_fused" not in _lib.PROFILE
    finally:
        ops.FUSED_VERTEX_FWD = prev
        _lib.PROFILE = None
    d = ops.encode_apply(xy, n_ls, n_host, tables, vi, w, vstr, path="direct")
    assert torch.equal(a, b) and torch.equal(a, d)


def _net(models, mode, L=16, T=2 ** 17, n_max=256):
    torch.manual_seed(3)
    net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=16, n_max=n_max,
                                          MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                          HPD_out_features=T, feature_dim=2, topk_k=4)
    net.return_indices = False
    net.dense_probs = False
    if mode != "hash":
        for p in net.HPD.parameters():
            p.requires_grad = False
        net.compute_pbar = False
    with torch.no_grad():
        net.encoding.packed_tables().mul_(100.0)
    return net


def _batches(n, P, seed=5):
    g = torch.Generator(device=DEV).manual_seed(seed)
    out = []
    for _ in range(n):
        xy = torch.rand((P, 2), device=DEV, generator=g)
        xy[:, 1] *= 0.7
        out.append((xy.contiguous(), torch.rand((P, 3), device=DEV, generator=g)))
    return out


@pytest.mark.parametrize("mode", ["hash", "gngf_frozen"])
def test_pipelined_unrolled_graph_equals_the_unpipelined_one(mode):
    """GraphedStep(unroll=4) at 32 encoder features (the fused training decoder: the headline's chain), four DIFFERENT batches:
    with the pipeline on, steps 1..3 find their pixels binned by the riders of the step before (3 hits at capture, the kernels
    replay); with it off every step bins itself.  Per-step rgb / loss and the decoder gradients of the last step are equal bit
    for bit (the binned ORDER of the pixels differs — block order inside a tile is whatever the atomics make it — but no result
    depends on it: enc rows are written by original index, dG64 is an exact integer sum); table gradients: bit for bit wherever
    at most two vertices share a row (a + b = b + a), a few ulp where float atomics of three or more arrive in another order."""
    from collision_handling_in_instantngp_amd import models, ops, train
    P = 2 ** 18
    models.should_use_hash_function = mode == "hash"
    prev = ops.BIN_PIPELINE
    try:
        res = {}
        for pipelined in (True, False):
            ops.BIN_PIPELINE = pipelined
            net = _net(models, mode)
            loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
            gs = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3, unroll=4)
            batches = _batches(4, P)
            pipe = net.dp.pipeline
            h0 = pipe.hits
            rs = gs.run_many(batches)
            assert pipe.hits - h0 == (3 if pipelined else 0), (pipe.hits, pipe.misses)
            rs = gs.run_many(batches[::-1])                 # other data in the same static buffers: the riders bin at replay time
            rs = gs.run_many(batches)
            torch.cuda.synchronize()
            res[pipelined] = {"rgb": [r.out.clone() for r in rs], "mse": [r.mse.clone() for r in rs],
                              "dec": [p.grad.clone() for p in net.mlp.parameters()],
                              "tab": torch.stack([m.weight.grad for m in net.encoding._hash_tables]).clone()}
        a, b = res[True], res[False]
        for x, y in zip(a["rgb"] + a["mse"] + a["dec"], b["rgb"] + b["mse"] + b["dec"]):
            assert torch.equal(x, y)
        same = (a["tab"] == b["tab"])
        assert float(same.float().mean()) > 0.999, float(same.float().mean())
        mx = float(b["tab"].abs().max())
        assert mx > 0 and float((a["tab"] - b["tab"]).abs().max()) <= 1e-6 * mx
        if mode == "hash":
            # rows that received at most two vertices: bit for bit
            assert bool(torch.isfinite(a["tab"]).all())
    finally:
        ops.BIN_PIPELINE = prev
        models.should_use_hash_function = False


def _eager_step(net, train, loss_fn, xy, tgt):
    net.zero_grad(set_to_none=True)
    with net.fused_mse(tgt, gloss=1.0):
        rgb, probs, _i, _c = net(xy, 1.0)
    empty = torch.tensor([], device=DEV)
    mse, kls, coll = loss_fn(rgb, tgt, None, probs, empty, empty)
    train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
    return rgb.detach().clone(), torch.stack([m.weight.grad for m in net.encoding._hash_tables]).clone()


def test_broken_announcements_fall_back_to_binning_at_the_head_of_the_step():
    from collision_handling_in_instantngp_amd import models, ops, train
    models.should_use_hash_function = True
    try:
        net = _net(models, "hash")
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        P = 2 ** 17
        (x0, y0), (x1, y1), (x2, y2) = _batches(3, P)
        pipe = net.dp.pipeline
        # reference results without any announcement
        want1 = _eager_step(net, train, loss_fn, x1, y1)
        want2 = _eager_step(net, train, loss_fn, x2, y2)
        assert pipe.hits == 0

        def check(got, want):
            assert torch.equal(got[0], want[0])
            assert float((got[1] - want[1]).abs().max()) <= 1e-6 * float(want[1].abs().max())

        # (1) the announced tensor is the one the next step runs on: hit
        pipe.announce(x1)
        _eager_step(net, train, loss_fn, x0, y0)
        h = pipe.hits
        check(_eager_step(net, train, loss_fn, x1, y1), want1)
        assert pipe.hits == h + 1
        # (2) announced, then edited in place before the next step: the version differs -> miss, binned again, right result
        x1b = x1.clone()
        pipe.announce(x1b)
        _eager_step(net, train, loss_fn, x0, y0)
        x1b.copy_(x2)
        h, m = pipe.hits, pipe.misses
        check(_eager_step(net, train, loss_fn, x1b, y2), want2)
        assert pipe.hits == h and pipe.misses == m + 1
        # (3) announced one tensor, called with another
        pipe.announce(x1)
        _eager_step(net, train, loss_fn, x0, y0)
        h = pipe.hits
        check(_eager_step(net, train, loss_fn, x2, y2), want2)
        assert pipe.hits == h
        # (4) a forward pass whose backward never runs (evaluation between two steps): the count half has run, the scatter half
        # has not — the persistent counters hold that batch's totals and are cleared before anybody bins with them
        pipe.announce(x1)
        with net.fused_mse(y0, gloss=1.0):
            rgb, *_ = net(x0, 1.0)                        # needs grad (training mode), count riders launched; no backward
        assert pipe.pending is not None
        del rgb
        check(_eager_step(net, train, loss_fn, x1, y1), want1)
        assert pipe.pending is None
        check(_eager_step(net, train, loss_fn, x2, y2), want2)
        # (5) another batch size announced: ignored
        pipe.announce(x1[: P // 2].contiguous())
        _eager_step(net, train, loss_fn, x0, y0)
        h = pipe.hits
        check(_eager_step(net, train, loss_fn, x1, y1), want1)
        assert pipe.hits == h
        # (6) inference (no gradient): nothing is announced to the riders, nothing pends
        pipe.announce(x1)
        with torch.no_grad():
            net(x0, 1.0)
        assert pipe.pending is None
        check(_eager_step(net, train, loss_fn, x1, y1), want1)
    finally:
        models.should_use_hash_function = False


def test_train_epoch_announces_the_next_slice(golden):
    """train.train_epoch (eager): the batches are fixed slices of one permutation (functions.py:186-194); every step but the first
    finds its pixels binned, and the epoch's results equal the unpipelined epoch's."""
    from collision_handling_in_instantngp_amd import models, ops, train
    img = golden("strawberry_rgb")["img"]
    h, w = img.shape[:2]
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    X = (torch.tensor(np.stack([rows, cols], -1).reshape(-1, 2)).float() / (max(w, h) - 1)).to(DEV)
    Y = torch.tensor(img.reshape(-1, 3) / 255).float().to(DEV)
    perm = torch.randperm(h * w, generator=torch.Generator().manual_seed(1)).to(DEV)
    models.should_use_hash_function = True
    prev = ops.BIN_PIPELINE
    try:
        outs = {}
        for pipelined in (True, False):
            ops.BIN_PIPELINE = pipelined
            net = _net(models, "hash", L=16, T=2 ** 15, n_max=256)
            loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
            opt = train.get_optimizer(net, 1e-3, 1e-3, 1e-3, 0.0, 0.0, 1e-6)
            rec = train.train_epoch(net, loss_fn, opt, X, Y, w, h, 1, 1, 1e-3, batch_percentage=1 / 4, should_shuffle=True,
                                    shuffled_indices=perm)
            torch.cuda.synchronize()
            outs[pipelined] = (rec["outputs"].clone(), torch.stack(rec["mse"]).clone(), net.dp.pipeline.hits)
        assert outs[True][2] == 3 and outs[False][2] == 0
        assert torch.equal(outs[True][1][:1], outs[False][1][:1])                     # first step: identical
        assert float((outs[True][0] - outs[False][0]).abs().max()) <= 2e-3              # later steps: Adam amplifies atomic-order ulps
        assert float((outs[True][1] - outs[False][1]).abs().max()) <= 1e-4 * float(outs[False][1].abs().max())
    finally:
        ops.BIN_PIPELINE = prev
        models.should_use_hash_function = False


@pytest.mark.parametrize("unroll", [1, 3])
def test_binning_across_replays_equals_binning_at_the_head_of_every_replay(unroll):
    """GraphedStep(cross_replay=True): the LAST step of a replay bins the first batch of the NEXT one when the caller names it in
    advance (run_many(..., next_first=) / gs(x, y, next_first=)); with unroll = 1 — the data-parallel case — every step's binning
    rides on the step before.  Three graphs per shape (cold, and two steady ones ping-ponging between two workspaces).  Results
    equal a GraphedStep without it on the same sequence of DIFFERENT batches; an announcement that is not honoured (another
    tensor, or none) falls back to the cold graph."""
    from collision_handling_in_instantngp_amd import models, ops, train
    P = 2 ** 17
    models.should_use_hash_function = True
    try:
        seq = [_batches(unroll, P, seed=100 + r) for r in range(5)]
        res = {}
        for cross in (False, True):
            net = _net(models, "hash")
            loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
            opt = train.get_optimizer(net, 1e-3, 1e-3, 1e-3, 0.0, 0.0, 1e-6)
            gs = train.GraphedStep(net, loss_fn, opt, 1, 1, 1e-3, unroll=unroll, cross_replay=cross)
            outs, used = [], []
            for r, batches in enumerate(seq):
                nf = seq[r + 1][0][0] if r + 1 < len(seq) else None
                if r == 2:
                    nf = seq[0][0][0]                      # announce one batch, run another at r = 3: cold graph again
                st = next(iter(gs._graphs.values())) if gs._graphs else None
                before = None if st is None else st.get("pre")
                rs = gs.run_many(batches, next_first=nf) if unroll > 1 else [gs(*batches[0], next_first=nf)]
                torch.cuda.synchronize()
                outs.append([(x.out.clone(), x.mse.clone()) for x in rs])
                used.append(before)
            st = next(iter(gs._graphs.values()))
            if cross:
                assert set(st["variants"]) == {"cold", "01", "10"} or set(st["variants"]) == {"cold", "01"}, set(st["variants"])
                assert used[1] is not None and used[2] is not None and used[4] is not None      # announced and honoured
            else:
                assert set(st["variants"]) == {"cold"} and "W" not in st
            res[cross] = (outs, {k: p.detach().clone() for k, p in net.named_parameters()})
        for a, b in zip(res[True][0], res[False][0]):
            for (oa, ma), (ob, mb) in zip(a, b):
                assert float((oa - ob).abs().max()) <= 2e-3 and float((ma - mb).abs()) <= 1e-4 * float(mb.abs())
        assert torch.equal(res[True][0][0][0][0], res[False][0][0][0][0])          # the very first step: identical
    finally:
        models.should_use_hash_function = False


def test_gradients_a_caller_reads_after_a_replay_are_that_replays():
    """Three graphs per shape (cold / two steady) each write their gradients into their OWN pool memory; p.grad, the encoder's
    one-buffer table gradient and the data-parallel bookkeeping have to follow the variant that was replayed (GraphedStep._adopt)
    — they used to keep pointing at the variant captured LAST.  No optimizer: the gradients of every replay have an eager twin."""
    from collision_handling_in_instantngp_amd import models, train
    P = 40000
    models.should_use_hash_function = True
    try:
        net = _net(models, "hash", L=8, T=2 ** 14, n_max=128)
        g = torch.Generator(device=DEV).manual_seed(1)
        xs = [torch.rand((P, 2), device=DEV, generator=g) for _ in range(3)]
        ys = [torch.rand((P, 3), device=DEV, generator=g) for _ in range(3)]
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        empty = torch.tensor([], device=DEV)

        def table_grads():
            return torch.stack([m.weight.grad for m in net.encoding._hash_tables]).clone()

        ref = []
        for x, y in zip(xs, ys):
            net.zero_grad()
            with net.fused_mse(y, gloss=1.0):
                rgb, probs, _i, _c = net(x, 1.0)
            mse, kls, coll = loss_fn(rgb, y, None, probs, empty, empty)
            train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
            torch.cuda.synchronize()
            ref.append((table_grads(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None and "hash" not in n}))
        gs = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3, unroll=1, cross_replay=True)
        seen = set()
        for rep in range(2):
            for k in range(3):
                st = next(iter(gs._graphs.values())) if gs._graphs else None
                pre = None if st is None else st.get("pre")
                seen.add("cold" if pre is None else ("01" if pre[0] == 0 else "10"))
                gs(xs[k], ys[k], next_first=xs[(k + 1) % 3])
                torch.cuda.synchronize()
                got = table_grads()
                scale = float(ref[k][0].abs().max())
                assert float((got - ref[k][0]).abs().max()) <= 2e-5 * scale, (rep, k)
                base = net.encoding._grad_base if net.encoding._grad_base is not None else net.encoding._grad_base_fp32
                if base is not None:
                    assert float((base.float() - ref[k][0]).abs().max()) <= 2e-5 * scale, (rep, k)
                for n, p in net.named_parameters():
                    if n in ref[k][1]:
                        want = ref[k][1][n]
                        assert float((p.grad - want).abs().max()) <= 2e-4 * float(want.abs().max()) + 1e-9, (rep, k, n)
        assert seen == {"cold", "01", "10"}, seen
    finally:
        models.should_use_hash_function = False
