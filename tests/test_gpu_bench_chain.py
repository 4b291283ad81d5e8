"""GPU: the EXACT chain of launches bench.py times, checked against the CPU oracle at full size, plus the cases VERDICT r3 / ADVICE r3
named as uncovered:
  * cfg2 / cfg3, P = 2^20, the step as bench.py builds it (bench.build_model + train.GraphedStep + net.fused_mse(gloss=1), both index
    sources): rgb, ALL 16 x 2^19 table-gradient rows and the six decoder gradients against oracle/gngf_oracle_c.c (pinned by
    tests/test_oracle_c.py to the numpy oracle, which is pinned to the reference's goldens);
  * the encoder chain alone on the step's own d enc: every table-gradient row against the double-precision sum of the reference's
    fp32 terms, with a RELATIVE bound on rows far below the maximum (the fixed-point grid has one scale per launch);
  * the 4096^2 shape through the MODEL (fused decoder => a bound on |d enc| exists) — the shape whose staged levels do not fit the
    level-interleaved image, so the generic pixel-stage kernels run and must start from a zeroed vertex grid;
  * should_softmax_topk_features in {None, False} at model level (golden G17, written by the reference) and at the kernel
    (gngf_blend_fwd / _bwd codes 1, 2 against oracle.blend_weights and its backward)."""
import numpy as np
import pytest
import torch

from conftest import parity_close
from oracle import c_oracle, gngf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
P = 2 ** 20


def _poison_allocator(nbytes=3 << 30):
    """Fills the caching allocator's free blocks with NaN so that any buffer the step leaves uninitialised (torch.empty) shows:
    a gradient that reads it is NaN instead of whatever the previous test left there."""
    blocks = []
    for sz in (nbytes // 2, nbytes // 4, nbytes // 8, 64 << 20, 16 << 20, 16 << 20, 4 << 20, 4 << 20):
        blocks.append(torch.full((sz // 4,), float("nan"), dtype=torch.float32, device=DEV))
    del blocks
    torch.cuda.synchronize()


def _numpy_state(net, L):
    sd = {k: v.detach().float().cpu().numpy() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    tables = np.ascontiguousarray(np.stack([sd[f"encoding._hash_tables.{l}.weight"] for l in range(L)]))
    dw = [np.ascontiguousarray(sd[f"mlp.{i}.0.weight"]) for i in range(3)]
    db = [np.ascontiguousarray(sd[f"mlp.{i}.0.bias"]) for i in range(3)]
    return tables, dw, db


def _oracle_step(x, y, n_ls, tables, dw, db, vidx=None, vw=None, vstride=0, genc_override=None):
    """forward + MSE gradient + backward of the path on the host (C/OpenMP oracle): rgb, d enc, decoder gradients, and the table
    gradient as the double-precision sum of the reference's fp32 terms."""
    enc = c_oracle.encode_fwd(x, tables, n_ls, vidx, vw, vstride)
    rgb, h1, h2 = c_oracle.decoder_fwd(enc, dw, db)
    drgb = ((2.0 / rgb.size) * (rgb - y)).astype(np.float32)                 # MSELoss backward seeded with 1 (utils.py:99)
    genc, gdec = c_oracle.decoder_bwd(enc, h1, h2, rgb, drgb, dw)
    g_used = genc if genc_override is None else genc_override
    dt64 = c_oracle.encode_bwd_f64(x, tables.shape, n_ls, np.ascontiguousarray(g_used), vidx, vw, vstride)
    return {"enc": enc, "rgb": rgb, "genc": genc, "gdec": gdec, "dt64": dt64,
            "mse": float(np.mean((rgb.astype(np.float64) - y) ** 2))}


@pytest.mark.skipif(not c_oracle.available(), reason="oracle/libgngf_oracle_c.so not built (make -C oracle)")
@pytest.mark.parametrize("mode", ["hash", "gngf_frozen"])
def test_bench_step_at_full_size_matches_the_oracle(mode):
    """VERDICT r3 item 2(b).  The model, the batch and the step are built by bench.py's own functions; the graph is replayed twice
    (the second replay is what a timed step is).  The encoder runs binning -> tiled_fwd_il -> decoder_train -> tiled_bwd_il (dG64)
    -> vertex_bwd_hash64 | vertex_bwd_sorted<FROM64>."""
    import bench
    from collision_handling_in_instantngp_amd import models, ops, train
    cfg = bench.MODES[mode]
    c = bench.SHAPES[cfg]
    L, T, F = c["L"], c["T"], c["F"]
    xy, target, bounds = bench.make_batch(cfg, P, 0, torch.device(DEV))
    _poison_allocator()
    trace = []
    prev_trace, ops.PIXEL_BWD_TRACE = ops.PIXEL_BWD_TRACE, trace
    try:
        net, _ = bench.build_model(mode, torch.device(DEV), bounds)
        # tables at 100x the reference's init (+-1e-4): at the init scale the decoder output barely depends on the encoder and the
        # table gradient of every row is the same few bits; the kernels do not care, the comparison does
        with torch.no_grad():
            net.encoding.packed_tables().mul_(100.0)
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        gs = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3, unroll=2)          # as bench.py: several steps per replayed graph
        res = gs.run_many([(xy, target)] * 2)
        res = gs.run_many([(xy, target)] * 2)
        torch.cuda.synchronize()
        r = res[-1]
        assert len(trace) >= 2 and all(t_["bound"] and t_["interleaved"] and t_["Ls"] == L and t_["P"] == P for t_ in trace), trace[:2]
        if mode == "hash":      # round 5: the store pass of tiled_bwd_il adds to the table-gradient rows itself (no fixed-point grid, no vertex launch)
            assert all(t_["direct_hash"] and t_["hash_fuse"] and not t_["dG64"] and not t_["fp32_grid"] for t_ in trace), trace[:2]
        else:                   # vertex_bwd_sorted reads the fixed-point grid itself
            assert all(t_["dG64"] and not t_["fp32_grid"] and not t_["direct_hash"] for t_ in trace), trace[:2]
        tables, dw, db = _numpy_state(net, L)
        n_ls = np.array(net._n_ls_host, np.int32)
        vidx = vw = None
        vstride = 0
        if mode == "gngf_frozen":
            tv, ti, w, vstride, NV, order = net._frozen_vertex_table(ops.BLEND_CODES[True])
            vidx, vw = np.ascontiguousarray(ti.cpu().numpy()), np.ascontiguousarray(w.cpu().numpy())
        x_np, y_np = np.ascontiguousarray(xy.cpu().numpy()), np.ascontiguousarray(target.cpu().numpy())
        want = _oracle_step(x_np, y_np, n_ls, tables, dw, db, vidx, vw, vstride)
        parity_close(r.out, want["rgb"], 0, 1e-5, f"bench step {mode}: rgb, all 2^20 pixels vs C oracle")
        parity_close(r.mse, want["mse"], 1e-5, 0, f"bench step {mode}: MSE value")
        got_dt = torch.stack([net.encoding._hash_tables[l].weight.grad for l in range(L)]).double().cpu().numpy()
        assert np.isfinite(got_dt).all()
        mx = float(np.abs(want["dt64"]).max())
        parity_close(got_dt, want["dt64"], 0, 1e-5 * mx, f"bench step {mode}: table gradient, all {L} x 2^19 rows vs C oracle (atol 1e-5 of max)")
        touched = np.abs(want["dt64"]) > 0
        assert np.array_equal(got_dt != 0, touched) or (np.abs(got_dt[~touched]).max() == 0), "rows no pixel touches stay exactly zero"
        names = ["mlp.0.0.weight", "mlp.0.0.bias", "mlp.1.0.weight", "mlp.1.0.bias", "mlp.2.0.weight", "mlp.2.0.bias"]
        params = dict(net.named_parameters())
        for nm, wg in zip(names, want["gdec"]):
            scale = float(np.abs(wg).max()) + 1e-30
            parity_close(params[nm].grad, wg, 1e-3, 2e-5 * scale, f"bench step {mode}: grad {nm} vs C oracle")
    finally:
        ops.PIXEL_BWD_TRACE = prev_trace
        models.should_use_hash_function = False


@pytest.mark.skipif(not c_oracle.available(), reason="oracle/libgngf_oracle_c.so not built (make -C oracle)")
@pytest.mark.parametrize("mode", ["hash", "gngf_frozen"])
def test_fixed_point_chain_on_the_steps_own_gradient_keeps_small_rows(mode):
    """VERDICT r3 weak #3.  The encoder backward of the bench's chain on the d enc the step itself produced (captured with a tensor
    hook), against the double-precision sum of the reference's fp32 terms g * c (* w) over all 2^20 pixels.  One scale per launch
    means a quantum of 2^-40 of the BATCH maximum per term: rows at 1e-4 .. 1e-7 of the largest row must still carry a relative
    accuracy of ~1e-4 (they hold >= 5 digits), which a bound relative to the maximum alone would not show."""
    import bench
    from collision_handling_in_instantngp_amd import models, ops
    cfg = bench.MODES[mode]
    c = bench.SHAPES[cfg]
    L, T, F = c["L"], c["T"], c["F"]
    xy, target, bounds = bench.make_batch(cfg, P, 0, torch.device(DEV))
    _poison_allocator()
    trace, captured = [], []
    prev_trace, ops.PIXEL_BWD_TRACE = ops.PIXEL_BWD_TRACE, trace
    real_decoder_apply = ops.decoder_apply

    def spy(enc, *a, **kw):
        enc.register_hook(lambda g: captured.append(g.detach().clone()))
        return real_decoder_apply(enc, *a, **kw)
    try:
        net, _ = bench.build_model(mode, torch.device(DEV), bounds)
        with torch.no_grad():
            net.encoding.packed_tables().mul_(100.0)
        ops.decoder_apply = spy
        with net.fused_mse(target, gloss=1.0):
            rgb, _probs, _idx, _c = net(xy, 1.0)
        loss = ops.mse_loss(rgb, target)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.decoder_apply = real_decoder_apply
        ops.PIXEL_BWD_TRACE = prev_trace
        models.should_use_hash_function = False
    assert len(captured) == 1 and len(trace) == 1 and trace[0]["interleaved"] and trace[0]["bound"], trace
    assert trace[0]["direct_hash"] if mode == "hash" else trace[0]["dG64"], trace
    genc = np.ascontiguousarray(captured[0].cpu().numpy())
    n_ls = np.array(net._n_ls_host, np.int32)
    vidx = vw = None
    vstride = 0
    if mode == "gngf_frozen":
        tv, ti, w, vstride, NV, order = net._frozen_vertex_table(ops.BLEND_CODES[True])
        vidx, vw = np.ascontiguousarray(ti.cpu().numpy()), np.ascontiguousarray(w.cpu().numpy())
    x_np = np.ascontiguousarray(xy.cpu().numpy())
    want = c_oracle.encode_bwd_f64(x_np, (L, T, F), n_ls, genc, vidx, vw, vstride)
    got = torch.stack([net.encoding._hash_tables[l].weight.grad for l in range(L)]).double().cpu().numpy()
    mx = float(np.abs(want).max())
    parity_close(got, want, 0, 2e-6 * mx, f"dG64 chain {mode}: table gradient on the step's own d enc, all rows (atol 2e-6 of max)")
    # Small rows.  A table row is an fp32 sum (float atomics of per-vertex sums — the reference's index_put accumulates in fp32
    # too), so what can be asked of ANY fp32 implementation is an error bounded by a few fp32 roundings of the row's ABSOLUTE mass
    # A = sum |term| (a row that is small because large terms cancel keeps their rounding).  The fixed-point grid must not add to
    # that: with one scale per launch (quantum 2^-40 of the batch's max |d enc| per term) a row whose mass is 1e-7 .. 1e-4 of the
    # largest mass still has to come out within that bound — a grid that flushed small sums would miss it by orders of magnitude.
    # The comparison is with the EXACT gradient of the fp32 inputs (products formed in double): the fixed-point terms are the
    # exactly rounded products RNE(g c 2^S), the reference's are fp32-rounded products (the stated deviation, covered by the
    # absolute check above).
    want = c_oracle.encode_bwd_f64(x_np, (L, T, F), n_ls, genc, vidx, vw, vstride, exact_products=True)
    mass = c_oracle.encode_bwd_f64(x_np, (L, T, F), n_ls, np.abs(genc), vidx, vw, vstride, exact_products=True)     # c, w >= 0
    amx = float(mass.max())
    small = (mass < 1e-4 * amx) & (mass > 1e-7 * amx)
    assert small.sum() > (1000 if mode == "hash" else 100), int(small.sum())
    # hash: a row is the float-atomic sum of one to a few per-vertex sums (3 roundings); vertex-table source: thousands of
    # (vertex, k) entries meet in a row through a DPP segmented scan, a chain across the waves of a workgroup and float atomics
    # across workgroups (tens of roundings)
    k = 4e-7 if mode == "hash" else 4e-6
    ratio = np.abs(got[small] - want[small]) / mass[small]
    print(f"[dG64 {mode}] rows with mass in (1e-7, 1e-4) x max: {int(small.sum())}, worst |err| / mass {ratio.max():.2e} (bound {k:g}), "
          f"median {np.median(ratio):.2e}")
    assert float(ratio.max()) <= k, float(ratio.max())
    ratio_all = np.abs(got - want)[mass > 0] / mass[mass > 0]
    assert float(ratio_all.max()) <= k, float(ratio_all.max())
    from conftest import PARITY
    PARITY.record(f"dG64 chain {mode}: |err| / (row's absolute mass), rows with mass 1e-7..1e-4 of the largest", ratio, np.zeros_like(ratio), 0, k)
    # VERDICT r4 item 6 (second half): the criterion above is relative to the row's ABSOLUTE MASS — the right one for an fp32 sum with
    # cancellation — which hides how the few rows that are small THROUGH CANCELLATION look relative to their own value; that number
    # is recorded next to it (not bounded: a row whose terms cancel to 1e-3 of their mass carries 1e3 x the rounding of its terms, in
    # the reference's own index_put as here): worst |err| / |value| over the same rows, and how many exceed 1e-4
    nz = small & (np.abs(want) > 0)
    rel_val = np.abs(got[nz] - want[nz]) / np.abs(want[nz])
    cancel = mass[nz] / np.abs(want[nz])
    worst = int(np.argmax(rel_val))
    print(f"[dG64 {mode}] same rows, relative to their own VALUE: worst |err| / |value| {rel_val.max():.2e} (that row's mass / |value| = "
          f"{cancel[worst]:.1e}), {int((rel_val > 1e-4).sum())} of {int(nz.sum())} rows above 1e-4, median {np.median(rel_val):.2e}")
    PARITY.record(f"dG64 chain {mode}: |err| / |value| of the same small rows (recorded, not bounded: rows small through cancellation)",
                  rel_val, np.zeros_like(rel_val), 0, float("inf"))
    assert float(np.median(rel_val)) <= 1e-6


@pytest.mark.parametrize("mode", ["hash", "gngf_frozen"])
@pytest.mark.parametrize("interleaved_off", [False, True])
def test_model_gradients_when_the_generic_pixel_stage_runs(mode, interleaved_off):
    """ADVICE r3 (high).  With a bound on |d enc| (every step through the fused decoder has one) the Python side used to hand the
    C side an UNINITIALISED fp32 vertex grid next to the fixed-point one whenever the level-interleaved kernel did not apply —
    the 4096^2 shape (297 KB image), or any shape after gngf_set_tiled_interleaved(0) — and the generic kernels accumulated
    into it.  Now the launcher's own decision (gngf_tiled_interleaved_applies) picks the buffers.  The allocator is poisoned with
    NaN first; the table gradient of the tiled dispatch must equal the direct form's (float atomics: fp32 round-off)."""
    import bench
    from collision_handling_in_instantngp_amd import models, ops, _lib
    shape = "cfg2" if interleaved_off else "cfg4"
    c = bench.SHAPES[shape]
    L, T, F = c["L"], c["T"], c["F"]
    xy, target, bounds = bench.make_batch(shape, P, 0, torch.device(DEV))
    models.should_use_hash_function = mode == "hash"
    prev_il = _lib.query("gngf_set_tiled_interleaved", 0 if interleaved_off else 1)
    trace = []
    prev_trace, ops.PIXEL_BWD_TRACE = ops.PIXEL_BWD_TRACE, trace
    try:
        torch.manual_seed(7)
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=c["n_min"], n_max=c["n_max"],
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=T, feature_dim=F, topk_k=4).to(DEV)
        net.return_indices = False
        net.dense_probs = False
        if mode == "gngf_frozen":
            for p_ in net.HPD.parameters():
                p_.requires_grad = False
            net.compute_pbar = False
        with torch.no_grad():
            net.encoding.packed_tables().mul_(100.0)
        grads = {}
        for path in ("tiled", "direct"):
            net.zero_grad()
            ops.ENCODE_PATH = path
            _poison_allocator()
            try:
                with net.fused_mse(target, gloss=1.0):
                    rgb, _p, _i, _c = net(xy, 1.0)
                ops.mse_loss(rgb, target).backward()
            finally:
                ops.ENCODE_PATH = "auto"
            torch.cuda.synchronize()
            grads[path] = torch.stack([net.encoding._hash_tables[l].weight.grad for l in range(L)]).clone()
        assert len(trace) == 1 and not trace[0]["interleaved"] and not trace[0]["dG64"] and trace[0]["bound"], trace
        assert 0 < trace[0]["Ls"] <= L
        gt, gd = grads["tiled"], grads["direct"]
        assert bool(torch.isfinite(gt).all()), "the tiled dispatch read an uninitialised buffer (NaN-poisoned allocator)"
        mx = float(gd.abs().max())
        assert mx > 0
        # (vertex-table source with a freshly initialised HPD: a million entries meet in a few dozen rows, and the DIRECT form
        # adds them one float atomic at a time — its own fp32 accumulation error is what the looser bound covers)
        tol = 2e-5 if mode == "hash" else 5e-3
        parity_close(gt, gd, 1e-3, tol * mx, f"{shape} {mode} interleaved_off={interleaved_off}: model table gradient, tiled (generic kernels) vs direct form")
    finally:
        ops.PIXEL_BWD_TRACE = prev_trace
        _lib.query("gngf_set_tiled_interleaved", prev_il)
        models.should_use_hash_function = False


# ------------------------------------------------------------------------------------------------ blend variants
def t(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


@pytest.mark.parametrize("tag,flag", [("raw", None), ("norm", False)])
def test_blend_variants_at_model_level_match_reference(golden, tag, flag, encode_path):
    """G17, written by the reference: should_softmax_topk_features in {None, False} (models.py:212-217) through
    GeneralNeuralGaugeFields.forward — the fast path (per-vertex table + gngf_blend_fwd / _bwd codes 1, 2), direct and tiled."""
    from collision_handling_in_instantngp_amd import models, train
    g = golden(f"G17_blend_{tag}")
    models.should_use_hash_function = False
    prev = models.should_softmax_topk_features
    models.should_softmax_topk_features = flag
    try:
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=256, num_levels=4, n_min=8, n_max=32,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=256, feature_dim=2, topk_k=4)
        sd = net.state_dict()
        net.load_state_dict({k: (t(g["init_" + k.replace(".", "_")]) if "init_" + k.replace(".", "_") in g else v) for k, v in sd.items()})
        img = golden("strawberry_rgb")["img"]
        h, w = img.shape[:2]
        rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        X = (torch.tensor(np.stack([rows, cols], -1).reshape(-1, 2)).float() / (max(w, h) - 1)).to(DEV)
        Y = torch.tensor(img.reshape(-1, 3) / 255).float().to(DEV)
        sl = t(g["perm"])
        rgb, probs, idx, _ = net(X[sl], 1 / 3)
        empty = torch.tensor([], device=DEV)
        mse, kls, coll = train.Loss(delta=1, gamma=-2, epsilon=1)(rgb, Y[sl], probs.shape[-1], probs, empty, empty)
        loss = train.assemble_loss(mse, kls, coll, 1, 1, 1e-3)
        loss.backward()
        parity_close(rgb, g["rgb"], 2e-5, 2e-6, f"G17 {tag} rgb")
        parity_close(mse, g["mse"], 2e-5, 0, f"G17 {tag} mse")
        parity_close(kls, g["kls"], 2e-4, 1e-9, f"G17 {tag} JS/KL")
        parity_close(loss, g["loss"], 2e-5, 0, f"G17 {tag} loss")
        assert (idx.cpu().numpy() == g["idx"]).mean() > 0.995
        n = 0
        for k_, p_ in net.named_parameters():
            gk = "grad_" + k_.replace(".", "_")
            if gk in g:
                scale = np.abs(g[gk]).max() + 1e-30
                parity_close(p_.grad, g[gk], 5e-3, 2e-4 * scale, f"G17 {tag} grad " + k_)
                n += 1
        assert n >= 4 + 8 + 6
        encode_path.assert_chain()
    finally:
        models.should_softmax_topk_features = prev


@pytest.mark.parametrize("flag", [True, None, False])
def test_blend_kernels_all_codes_vs_oracle(flag):
    """gngf_blend_fwd / _bwd (ops.BlendFunction) against oracle.blend_weights and the oracle's blend backward (the dp branch of
    oracle.encoding_backward), for the three values of should_softmax_topk_features; K in {1, 4, 7}, rows incl. ragged counts."""
    from collision_handling_in_instantngp_amd import ops
    rng = np.random.default_rng(17)
    for U, K in ((1, 4), (1000, 4), (4099, 7), (257, 1)):
        q = rng.random((U, K)).astype(np.float32) * (0.5 if K > 1 else 1.0) + 1e-3
        d = rng.standard_normal((U, K)).astype(np.float32)
        qt = t(q).requires_grad_(True)
        w = ops.BlendFunction.apply(qt, ops.BLEND_CODES[flag])
        w.backward(t(d))
        parity_close(w, orc.blend_weights(q, flag), 2e-6, 1e-9, f"blend fwd code {ops.BLEND_CODES[flag]} ({U}x{K})")
        q64, d64 = q.astype(np.float64), d.astype(np.float64)
        if flag is None:
            want = d64
        elif flag:
            w64 = np.exp(q64 - q64.max(-1, keepdims=True))
            w64 /= w64.sum(-1, keepdims=True)
            want = w64 * (d64 - (w64 * d64).sum(-1, keepdims=True))
        else:
            s = q64.sum(-1, keepdims=True)
            want = d64 / s - (d64 * q64).sum(-1, keepdims=True) / (s * s)
        # (K = 1 with the normalised blend: w = q / q = 1, the exact gradient is 0 and what is left is fp32 cancellation noise of
        # d / s - d q / s^2: the tolerance is relative to the size of those two terms)
        scale = float(np.abs(d64 / (q64.sum(-1, keepdims=True) if flag is False else 1.0)).max())
        parity_close(qt.grad, want, 2e-5, 2e-6 * scale, f"blend bwd code {ops.BLEND_CODES[flag]} ({U}x{K})")


@pytest.mark.parametrize("half", [False, True])
def test_sixty_four_feature_step_with_the_clear_inside_the_decoder_backward(half):
    """BASELINE config 5's width (L F = 64: no fused training kernel): with net.fused_mse the table-gradient buffer is left for the
    decoder BACKWARD kernel to clear on its way (gngf_decoder_bwd(..., zero_fill); ops.DECODER_BWD_CLEARS) instead of for rider
    workgroups of the binning launch.  Same gradients either way, on a NaN-poisoned allocator; fp32 and fp16 table storage."""
    from collision_handling_in_instantngp_amd import models, ops
    Pn = 2 ** 17
    g = torch.Generator(device=DEV).manual_seed(9)
    xy = torch.rand((Pn, 2), device=DEV, generator=g)
    target = torch.rand((Pn, 3), device=DEV, generator=g)
    models.should_use_hash_function = True
    prev = (ops.DECODER_BWD_CLEARS, ops.FP16_TABLE_GRAD_FP32)
    try:
        ops.FP16_TABLE_GRAD_FP32 = False
        torch.manual_seed(5)
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=2 ** 16, num_levels=16, n_min=16, n_max=1024,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=2 ** 16, feature_dim=4, topk_k=4,
                                              table_dtype=(torch.float16 if half else torch.float32)).to(DEV)
        net.return_indices = False
        with torch.no_grad():
            net.encoding.packed_tables().mul_(50.0)
        out = {}
        for on in (True, False):
            ops.DECODER_BWD_CLEARS = on
            net.zero_grad()
            _poison_allocator(1 << 30)
            with net.fused_mse(target, gloss=1.0):
                rgb, _p, _i, _c = net(xy, 1.0)
            ops.mse_loss(rgb, target).backward()
            torch.cuda.synchronize()
            out[on] = (rgb.detach().clone(), torch.stack([m.weight.grad.float() for m in net.encoding._hash_tables]).clone(),
                       [p_.grad.clone() for p_ in net.mlp.parameters()])
        assert torch.equal(out[True][0], out[False][0])
        assert bool(torch.isfinite(out[True][1]).all())
        mx = float(out[False][1].abs().max())
        assert mx > 0
        parity_close(out[True][1], out[False][1], 1e-3, (2e-3 if half else 2e-5) * mx, f"64-feature step (fp16 tables: {half}): table gradient, clear inside decoder_bwd vs riders")
        for a, b in zip(out[True][2], out[False][2]):
            assert torch.equal(a, b)
    finally:
        ops.DECODER_BWD_CLEARS, ops.FP16_TABLE_GRAD_FP32 = prev
        models.should_use_hash_function = False
