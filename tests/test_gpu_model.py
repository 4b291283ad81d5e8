"""GPU end-to-end parity: GeneralNeuralGaugeFields (HIP path) vs the reference's own outputs (G7 goldens):
rgb, loss terms, every parameter gradient, parameters after 1..3 Adam steps — hash and GNGF modes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def t(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


def close(a, b, rtol, atol, msg=""):
    """assert_allclose + a row in the achieved-error report (tests/conftest.py: ParityRecorder)"""
    import inspect
    from conftest import parity_close
    if not msg:
        ctx = inspect.stack()[1].code_context
        msg = (ctx[0].strip() if ctx else "")[:100]
    parity_close(a, b, rtol, atol, msg)


def strawberry(golden):
    img = golden("strawberry_rgb")["img"]
    h, w = img.shape[:2]
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    X = torch.tensor(np.stack([rows, cols], -1).reshape(-1, 2)).float() / (max(w, h) - 1)   # main.py:50-51
    Y = torch.tensor(img.reshape(-1, 3) / 255).float()                                      # utils.py:61
    return X.to(DEV), Y.to(DEV), h, w


def build(golden, mode):
    from collision_handling_in_instantngp_amd import models, train
    g = golden(f"G7_end_to_end_{mode}")
    models.should_use_hash_function = (mode == "hash")
    net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=256, num_levels=4, n_min=8, n_max=32,
                                          MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                          HPD_out_features=256, feature_dim=2, topk_k=4)
    sd = net.state_dict()
    new = {}
    for k in sd:
        gk = "init_" + k.replace(".", "_")
        new[k] = t(g[gk]) if gk in g else sd[k]
    net.load_state_dict(new)
    return net, g, models, train


@pytest.mark.parametrize("mode", ["hash", "gngf"])
def test_three_training_steps_match_reference(golden, mode, encode_path):
    net, g, models, train = build(golden, mode)
    try:
        X, Y, h, w = strawberry(golden)
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        opt = train.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
        B = 4096
        perm = t(g["perm"])
        for step in range(3):
            sl = perm[step * B:(step + 1) * B]
            opt.zero_grad()
            rgb, probs, idx, counts = net(X[sl], 1 / 3, should_calc_counts=False)
            empty = torch.tensor([], device=DEV)
            mse, kls, coll = loss_fn(rgb, Y[sl], None if probs is None else probs.shape[-1], probs, empty, empty)
            loss = train.assemble_loss(mse, kls, coll, 1, 1, 1e-3)
            loss.backward()
            s = f"s{step}_"
            # tolerances widen with the step: fp32 round-off is amplified by Adam's 1/sqrt(v) at eps = 1e-15
            rt = [2e-5, 2e-3, 1e-2][step]
            close(rgb, g[s + "rgb"], rt, 1e-6 * (1 + 50 * step), "rgb")
            close(mse, g[s + "mse"], rt, 0, "mse")
            close(loss, g[s + "loss"], rt, 0, "loss")
            if mode == "hash":
                assert np.array_equal(idx.cpu().numpy(), g[s + "idx"])
                assert probs is None and counts == []
            else:
                assert idx.shape == (B, 4, 4, 4) and idx.dtype == torch.int64
                assert (idx.cpu().numpy() == g[s + "idx"]).mean() > 0.995
                assert probs.shape == (B, 4, 4, 256)
                close(kls, g[s + "kls"], 10 * rt, 1e-9, "kls")
            if step == 0:
                for k_, p_ in net.named_parameters():
                    gk = s + "grad_" + k_.replace(".", "_")
                    if gk in g:
                        scale = np.abs(g[gk]).max() + 1e-30
                        close(p_.grad, g[gk], 5e-3, 2e-4 * scale, k_)
            before = {k_: p_.detach().clone() for k_, p_ in net.named_parameters()} if step == 0 else None
            opt.step()
            if step == 0:
                # Adam's first step moves a weight by lr * g / (|g| + eps), eps = 1e-15: exactly -lr * sign(g) wherever the
                # gradient is not round-off noise, and not at all where it is exactly zero.  Checked entry by entry where
                # the reference's |g| is well above the noise floor, with a tolerance << lr: a tensor that was not updated,
                # or moved the wrong way, fails.
                checked = 0
                for k_, p_ in net.named_parameters():
                    gk, gg = s + "param_" + k_.replace(".", "_"), s + "grad_" + k_.replace(".", "_")
                    if gk not in g or gg not in g:
                        continue
                    lr = 1e-4 if "hash_tables" in k_ else 1e-3
                    wd = 0.0 if "hash_tables" in k_ else 1e-6
                    init = g["init_" + k_.replace(".", "_")]
                    g_eff = g[gg] + wd * init                                 # torch.optim.Adam: L2-style weight decay
                    floor = 1e-2 * np.abs(g_eff).max()
                    mask = np.abs(g_eff) > floor
                    assert mask.any(), k_
                    moved = (p_.detach() - before[k_]).cpu().numpy()
                    want_moved = g[gk] - init
                    assert np.all(np.abs(want_moved[mask]) > 0.9 * lr), k_     # the reference took a full step there
                    close(moved[mask], want_moved[mask], 0, 0.02 * lr, f"Adam step 1, {k_}: update where |g| > 1e-2 max|g| (lr {lr:g})")
                    untouched = (g[gg] == 0) & (p_.grad.detach().cpu().numpy() == 0)
                    if wd == 0.0 and untouched.any():
                        assert np.all(moved[untouched] == 0), k_               # zero gradient, no weight decay: bit-identical
                    checked += int(mask.sum())
                assert checked > 1000
        encode_path.assert_chain(min_launches=3)
    finally:
        models.should_use_hash_function = False


def _cfg1_net(models, g, **kw):
    net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=256, num_levels=4, n_min=8, n_max=32,
                                          MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                          HPD_out_features=256, feature_dim=2, topk_k=4, **kw)
    sd = net.state_dict()
    net.load_state_dict({k: (t(g["init_" + k.replace(".", "_")]) if "init_" + k.replace(".", "_") in g else v) for k, v in sd.items()})
    return net


def test_keep_topk_only_two_steps_match_reference(golden, encode_path):
    """G15: should_keep_topk_only=True (reference models.py:478-484; half of its grid, params.py:58-75): `probs` is the
    (P,L,4,K) top-K tensor, the loss's distribution term runs with N = K (functions.py:226-232).  Two optimisation steps
    written by the reference itself: outputs, loss terms, every gradient, parameters after the first step."""
    from collision_handling_in_instantngp_amd import models, train
    g = golden("G15_keep_topk_only")
    models.should_use_hash_function = False
    net = _cfg1_net(models, g, should_keep_topk_only=True)
    X, Y, h, w = strawberry(golden)
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    opt = train.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
    B = 4096
    perm = t(g["perm"])
    empty = torch.tensor([], device=DEV)
    for step in range(2):
        sl = perm[step * B:(step + 1) * B]
        opt.zero_grad()
        rgb, probs, idx, counts = net(X[sl], 1 / 3, should_calc_counts=False)
        assert tuple(probs.shape) == (B, 4, 4, 4) and probs.requires_grad and tuple(idx.shape) == (B, 4, 4, 4)
        mse, kls, coll = loss_fn(rgb, Y[sl], probs.shape[-1], probs, empty, empty)
        loss = train.assemble_loss(mse, kls, coll, 1, 1, 1e-3)
        loss.backward()
        s = f"s{step}_"
        rt = [2e-5, 2e-3][step]
        close(rgb, g[s + "rgb"], rt, 1e-6 * (1 + 50 * step), "G15 rgb")
        same = (idx.cpu().numpy() == g[s + "idx"]).all(-1)
        assert same.mean() > 0.995
        close(probs.detach().cpu().numpy()[same], g[s + "probs"][same], 10 * rt, 1e-9, "G15 top-K probabilities returned as probs")
        close(mse, g[s + "mse"], rt, 0, "G15 mse")
        close(kls, g[s + "kls"], 10 * rt, 1e-9, "G15 JS/KL term with N = K")
        close(loss, g[s + "loss"], rt, 0, "G15 loss")
        if step == 0:
            for k_, p_ in net.named_parameters():
                gk = s + "grad_" + k_.replace(".", "_")
                if gk in g:
                    scale = np.abs(g[gk]).max() + 1e-30
                    close(p_.grad, g[gk], 5e-3, 2e-4 * scale, "G15 grad " + k_)
        opt.step()
    encode_path.assert_chain(min_launches=2)


@pytest.mark.parametrize("mode", ["hash", "gngf"])
def test_batchnorm_coordinates_match_reference(golden, mode):
    """G16: should_batchnorm_data=True (reference models.py:394-397; main.py:50 then feeds RAW pixel coordinates): nn.BatchNorm1d
    in training mode centres the coordinates on 0, so grid vertices are negative.  Hash: the direct-form kernels hash any
    integer vertex (indices bit-exact).  GNGF: the per-instance formulation on the module-boundary kernels."""
    from collision_handling_in_instantngp_amd import models, train
    g = golden(f"G16_batchnorm_{mode}")
    models.should_use_hash_function = mode == "hash"
    models.should_batchnorm_data = True
    try:
        net = _cfg1_net(models, g)
        net.train()
        img = golden("strawberry_rgb")["img"]
        h, w = img.shape[:2]
        rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        Xraw = torch.tensor(np.stack([rows, cols], -1).reshape(-1, 2)).float().to(DEV)          # main.py:50 skipped
        Y = torch.tensor(img.reshape(-1, 3) / 255).float().to(DEV)
        sl = t(g["perm"])
        rgb, probs, idx, _ = net(Xraw[sl], 1 / 3)
        empty = torch.tensor([], device=DEV)
        mse, kls, coll = train.Loss(delta=1, gamma=-2, epsilon=1)(rgb, Y[sl], None if probs is None else probs.shape[-1], probs, empty, empty)
        loss = train.assemble_loss(mse, kls, coll, 1, 1, 1e-3)
        loss.backward()
        close(rgb, g["rgb"], 2e-5, 2e-6, f"G16 batchnorm {mode} rgb")
        close(mse, g["mse"], 2e-5, 0, f"G16 batchnorm {mode} mse")
        close(loss, g["loss"], 2e-5, 0, f"G16 batchnorm {mode} loss")
        if mode == "hash":
            assert np.array_equal(idx.cpu().numpy(), g["idx"])                                  # negative vertices hashed bit-exactly
        else:
            assert (idx.cpu().numpy() == g["idx"]).mean() > 0.995
            assert tuple(probs.shape) == (2048, 4, 4, 256)
            close(kls, g["kls"], 2e-4, 1e-9, "G16 batchnorm gngf JS/KL")
            close(probs.sum(0).sum(1) / (probs.shape[0] * probs.shape[2]), g["pbar"], 1e-4, 1e-9, "G16 batchnorm gngf p-bar")
        for k_, p_ in net.named_parameters():
            gk = "grad_" + k_.replace(".", "_")
            if gk in g:
                scale = np.abs(g[gk]).max() + 1e-30
                close(p_.grad, g[gk], 5e-3, 2e-4 * scale, f"G16 batchnorm {mode} grad " + k_)
            else:
                assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, k_                # BatchNorm's affine parameters: no gradient path
        for k_, v_ in net.state_dict().items():
            if k_.startswith("_batch_norm.") and v_.dtype.is_floating_point:
                close(v_, g["after_" + k_.replace(".", "_")], 1e-5, 1e-6, "G16 " + k_)           # running statistics updated as the reference's
    finally:
        models.should_batchnorm_data = False
        models.should_use_hash_function = False


def test_state_dict_keys_and_shapes_match_reference_contract(golden):
    net, g, models, train = build(golden, "gngf")
    keys = set(net.state_dict().keys())
    want = {k[len("init_"):] for k in g if k.startswith("init_")}
    assert {k.replace(".", "_") for k in keys} == want
    assert net.encoding._hash_tables[0].weight.shape == (256, 2)


def test_compact_distribution_equals_dense(golden):
    """VertexDistribution.pbar (never expanding per-instance rows) == batch mean of the dense (P,L,4,T) tensor."""
    net, g, models, train = build(golden, "gngf")
    X, Y, h, w = strawberry(golden)
    sl = t(g["perm"])[:2048]
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    outs = {}
    for dense in (True, False):
        net.dense_probs = dense
        net.zero_grad()
        rgb, probs, idx, _ = net(X[sl], 1 / 3)
        empty = torch.tensor([], device=DEV)
        mse, kls, coll = loss_fn(rgb, Y[sl], probs.shape[-1], probs, empty, empty)
        train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
        outs[dense] = (kls.detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters() if p.grad is not None})
    close(outs[False][0], outs[True][0], 1e-4, 1e-9)
    for k in outs[True][1]:
        scale = np.abs(outs[True][1][k]).max() + 1e-30
        close(outs[False][1][k], outs[True][1][k], 2e-3, 1e-4 * scale, k)


@pytest.mark.parametrize("graph", [False, True])
def test_training_curve_matches_reference_three_epochs(golden, graph):
    """cfg1 (BASELINE configs[0]: strawberry.jpeg, params-ID 4061, L=4 T=2^8 K=4, 3 batches/epoch): MSE and PSNR of the
    first three epochs vs the reference's own train_step run (G8), called with the reference's positional signature and
    read through its 9-tuple.  PSNR must agree within 0.01 dB (north-star bar).  graph=True: every step replayed from a
    hipGraph (train.GraphedStep), previous-epoch collision statistics fed back as in functions.py:654-672."""
    from collision_handling_in_instantngp_amd import data, train
    net, g7, models, _ = build(golden, "gngf")
    g = golden("G8_train_curve")
    img = golden("strawberry_rgb")["img"]
    X, Y, h, w = strawberry(golden)
    shuffled = t(g["shuffled"].astype(np.int64))
    reordered = torch.empty_like(shuffled)
    reordered[shuffled] = torch.arange(shuffled.numel(), device=DEV)
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    opt = train.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
    prev_c = prev_m = None
    for e in range(3):
        r = train.train_step(net, loss_fn, opt, X, Y, w, h, 256, 4, 1, 1, 1e-3, 1 / 3, 4, False, False, True, shuffled, reordered,
                             prev_c, prev_m, graph=graph)
        loss, show, prev_c, prev_m, counts, mse, kls, colls, ipl = r
        assert show.shape == (h, w, 3) and show.dtype == np.int32 and kls.shape == (4,) and colls.shape == (4,)
        assert counts == [] and ipl == []
        psnr = train.calc_psnr(show, img)
        assert abs(psnr - float(g["psnr"][e])) < 0.01, (e, psnr, float(g["psnr"][e]))
        close(np.array(psnr), g["psnr"][e], 0, 0.01, f"G8 cfg1 PSNR epoch {e} (graph={graph})")
        close(np.array(mse), g["mse"][e], 2e-3, 0, f"G8 cfg1 MSE epoch {e} (graph={graph})")
        if e == 0:          # later epochs feed back collision statistics, which the reference takes over uninitialised slots
            close(np.array(loss), g["loss"][e], 2e-3, 0, f"G8 cfg1 loss epoch {e} (graph={graph})")


@pytest.mark.parametrize("mode", ["hash", "gngf"])
def test_diagnostics_contract_matches_reference(golden, mode):
    """forward(..., should_calc_counts=True) -> counts_per_level, and calc_hash_collisions (reference models.py:530-619)."""
    from collision_handling_in_instantngp_amd import models
    g = golden("G10_diagnostics")
    models.should_use_hash_function = (mode == "hash")
    try:
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=256, num_levels=4, n_min=8, n_max=32,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=256, feature_dim=2, topk_k=4)
        if mode == "gngf":
            sd = net.state_dict()
            for k in list(sd):
                gk = "gngf_init_" + k.replace(".", "_")
                if gk in g:
                    sd[k] = t(g[gk])
            net.load_state_dict(sd)
        X, Y, h, w = strawberry(golden)
        sel = t(g["sel"])
        rgb, probs, idx, counts = net(X[sel], 1 / 3, should_calc_counts=True)
        ref_idx = g[f"{mode}_idx"]
        if mode == "hash":
            assert np.array_equal(idx.cpu().numpy(), ref_idx)
        same = np.array_equal(idx.cpu().numpy(), ref_idx)
        assert len(counts) == 4
        if same:                      # identical indices (always in hash mode; in GNGF mode unless a top-K tie resolved differently)
            for l, c in enumerate(counts):
                ks = np.array(sorted(c.keys()), dtype=np.int64)
                assert np.array_equal(ks, g[f"{mode}_counts_keys_{l}"])
                assert np.array_equal(np.array([c[k] for k in ks]), g[f"{mode}_counts_vals_{l}"])
        coll, minc = net.calc_hash_collisions(t(ref_idx))
        np.testing.assert_allclose(coll.cpu().numpy().astype(np.float64), g[f"{mode}_collisions"].astype(np.float64))
        np.testing.assert_allclose(minc.cpu().numpy().astype(np.float64), g[f"{mode}_min_collisions"].astype(np.float64))
    finally:
        models.should_use_hash_function = False


@pytest.mark.parametrize("mode", ["hash", "gngf", "gngf_frozen"])
def test_tracked_collision_statistic_equals_the_one_taken_from_the_index_tensor(golden, mode):
    """calc_hash_collisions only needs the distinct slots per (level, rank) among the batch's corners
    (reference models.py:568-619, called at functions.py:327).  With start_collision_tracking() the forward passes mark them from
    the per-vertex table and a touched-vertex map (csrc/stats.hip) — no (P,L,4,K) int64 tensor.  Equal, batch by batch and
    accumulated over three batches, to calc_hash_collisions on the concatenated index tensors of the same passes; and
    train_step(return_indices=False) hands the reference's collision tensors back."""
    from collision_handling_in_instantngp_amd import models, train
    models.should_use_hash_function = (mode == "hash")
    try:
        torch.manual_seed(11)
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=2 ** 12, num_levels=8, n_min=8, n_max=96,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=2 ** 12, feature_dim=2, topk_k=4)
        if mode == "gngf_frozen":
            for p in net.HPD.parameters():
                p.requires_grad = False
            net.dense_probs = False
            net.compute_pbar = False
        X, Y, h, w = strawberry(golden)
        net.return_indices = True
        net.start_collision_tracking()
        idxs = []
        for lo in (0, 50000, 120000):
            with torch.no_grad():
                _rgb, _p, idx, _c = net(X[lo:lo + 30000].contiguous(), 1 / 3)
            idxs.append(idx)
            got, got_min = net.tracked_hash_collisions()
            want, want_min = net.calc_hash_collisions(torch.cat(idxs))
            assert torch.equal(got.cpu(), want.cpu()) and torch.equal(got_min.cpu(), want_min.cpu()), (lo, got, want)
            if mode == "gngf":                       # a trainable HPD: the table changes between batches, each batch marks its own
                with torch.no_grad():
                    net.HPD.module_list[3][0].bias.add_(torch.randn(2 ** 12, device=DEV))
        assert float(want.sum()) > 0
        net.start_collision_tracking()               # a new epoch starts from empty maps
        with torch.no_grad():
            _rgb, _p, idx, _c = net(X[:30000].contiguous(), 1 / 3)
        got, _ = net.tracked_hash_collisions()
        assert torch.equal(got.cpu(), net.calc_hash_collisions(idx)[0].cpu())
        net.stop_collision_tracking()
        # the reference's train_step contract with return_indices = False
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        opt = train.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
        perm = torch.randperm(h * w, generator=torch.Generator().manual_seed(3)).to(DEV)
        outs = {}
        for ret in (True, False):
            net.return_indices = ret
            for graph in ((False, True) if mode != "gngf" else (False,)):
                r = train.train_step(net, loss_fn, opt, X, Y, w, h, 2 ** 12, 4, 1, 1, 1e-3, batch_percentage=1 / 3, num_levels=8,
                                     shuffled_indices=perm, graph=graph)
                outs[(ret, graph)] = (r[2], r[3])
                assert r[2].numel() == 8 and r[3].numel() == 8
        if mode != "gngf":                            # (a trainable HPD moves between the epochs: the statistic moves with it)
            base = outs[(True, False)]
            for k_, v_ in outs.items():
                assert torch.equal(v_[0].cpu().double(), base[0].cpu().double()) and torch.equal(v_[1].cpu(), base[1].cpu()), k_
    finally:
        models.should_use_hash_function = False


def test_keep_topk_only_bw_and_leaky_variants_run_and_differentiate(golden):
    """constructor switches of the reference: should_keep_topk_only (probs = (P,L,4,K)), should_bw (1 output), LeakyReLU."""
    from collision_handling_in_instantngp_amd import models, train
    X, Y, h, w = strawberry(golden)
    models.should_leaky_relu = True
    try:
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=512, num_levels=5, n_min=8, n_max=64,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=512, feature_dim=4, topk_k=3, should_keep_topk_only=True, should_bw=True)
        rgb, probs, idx, counts = net(X[:3000], 1.0)
        assert rgb.shape == (3000, 1) and probs.shape == (3000, 5, 4, 3) and idx.shape == (3000, 5, 4, 3)
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        empty = torch.tensor([], device=DEV)
        mse, kls, coll = loss_fn(rgb, Y[:3000, :1], probs.shape[-1], probs, empty, empty)
        train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
        for k, p in net.named_parameters():
            if k.startswith("_batch_norm"):        # unused unless should_batchnorm_data (as in the reference)
                continue
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
        assert float(net.HPD.module_list[3][0].weight.grad.abs().sum()) > 0
    finally:
        models.should_leaky_relu = False


@pytest.mark.parametrize("mode", ["hash", "gngf"])
def test_edge_batches_empty_single_and_noncontiguous(golden, mode):
    """empty batch, one pixel, a non-contiguous coordinate view and coordinates that require grad (the reference's
    `rearranged_grid_coords.requires_grad_()` leaves coordinates without a gradient: _scale_to_grid is no_grad)."""
    net, g, models, train = build(golden, mode)
    try:
        X, Y, h, w = strawberry(golden)
        rgb, probs, idx, counts = net(X[:0], 1.0)
        assert rgb.shape == (0, 3) and counts == []
        rgb1, *_ = net(X[5:6], 1.0)
        big = torch.stack([X[:1000, 0], torch.zeros(1000, device=DEV), X[:1000, 1]], 1)
        view = big[:, ::2]                                      # stride-2 view of the coordinates
        assert not view.is_contiguous()
        xg = X[:1000].clone().requires_grad_()
        a, *_ = net(view, 1.0)
        b, *_ = net(xg, 1.0)
        assert torch.equal(a, b)
        assert torch.allclose(a[5], rgb1[0], rtol=0, atol=1e-7)
        b.sum().backward()
        assert xg.grad is None
    finally:
        models.should_use_hash_function = False


def test_fp16_table_storage_model_matches_fp32_model_on_rounded_tables(golden):
    """table_dtype=torch.float16 (extension for BASELINE config 5): same outputs as an fp32 model holding the rounded
    tables; table gradients come back in fp16, everything else in fp32; state-dict keys unchanged."""
    from collision_handling_in_instantngp_amd import models
    X, Y, h, w = strawberry(golden)
    models.should_use_hash_function = True
    try:
        kw = dict(input_dim=2, hash_table_size=2 ** 14, num_levels=8, n_min=16, n_max=256, MLP_hidden_layers_widths=[64, 64],
                  HPD_hidden_layers_widths=[32, 64, 128], HPD_out_features=2 ** 14, feature_dim=4, topk_k=4)
        torch.manual_seed(3)
        n16 = models.GeneralNeuralGaugeFields(**kw, table_dtype=torch.float16)
        n32 = models.GeneralNeuralGaugeFields(**kw)
        sd = {k: (v.float() if "hash_tables" in k else v) for k, v in n16.state_dict().items()}
        n32.load_state_dict(sd)
        assert set(n16.state_dict()) == set(n32.state_dict())
        assert n16.encoding._hash_tables[0].weight.dtype == torch.float16
        xb, yb = X[:40000], Y[:40000]
        outs = []
        for net in (n16, n32):
            rgb, _, idx, _ = net(xb, 1.0)
            # fp16 gradients need loss scaling like any fp16 training (an MSE-mean gradient of ~5e-7 is subnormal in fp16)
            (torch.nn.functional.mse_loss(rgb, yb) * 16384.0).backward()
            outs.append((rgb.detach(), net.encoding._hash_tables[7].weight.grad, net.mlp[0][0].weight.grad))
        assert torch.allclose(outs[0][0], outs[1][0], rtol=0, atol=1e-6)
        assert outs[0][1].dtype == torch.float16 and outs[1][1].dtype == torch.float32
        scale = float(outs[1][1].abs().max())
        assert float((outs[0][1].float() - outs[1][1]).abs().max()) <= 2e-3 * scale
        assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-7)
    finally:
        models.should_use_hash_function = False


@pytest.mark.parametrize("graph", [False, True])
def test_headline_shape_hash_training_curve_matches_reference(golden, graph):
    """L=16, F=2, T=2^19, N 16->512 (the headline table shape), hash indexing, strawberry.jpeg: two epochs of three
    1/3-image batches vs the reference's own train_step run on CPU (G11): PSNR within 0.01 dB.  The 64 MiB tables are
    regenerated from the same seeded CPU generator the golden script used."""
    from collision_handling_in_instantngp_amd import models, train
    g = golden("G11_hash_L16_T19_curve")
    img = golden("strawberry_rgb")["img"]
    X, Y, h, w = strawberry(golden)
    Lv, T, Fd = 16, 2 ** 19, 2
    models.should_use_hash_function = True
    try:
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=Lv, n_min=16, n_max=512,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=T, feature_dim=Fd, topk_k=4)
        gen = torch.Generator().manual_seed(65535 + 11)
        tabs = ((torch.rand((Lv, T, Fd), generator=gen) * 2 - 1) * 1e-4).to(DEV)
        sd = net.state_dict()
        for l in range(Lv):
            sd[f"encoding._hash_tables.{l}.weight"] = tabs[l]
        for k in list(sd):
            gk = "init_mlp_" + k[len("mlp."):].replace(".", "_") if k.startswith("mlp.") else None
            if gk and gk in g:
                sd[k] = t(g[gk])
        net.load_state_dict(sd)
        shuffled = t(g["shuffled"].astype(np.int64))
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        opt = train.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
        assert isinstance(opt, train.FusedAdam)               # hash mode too (the int64 _prime_numbers is in no group)
        for e in range(2):
            r = train.train_step(net, loss_fn, opt, X, Y, w, h, T, 4, 1, 1, 1e-3, batch_percentage=1 / 3, num_levels=Lv,
                                 should_shuffle=True, shuffled_indices=shuffled, graph=graph)
            loss, show, coll, minc, _cnt, mse, kls, colls, _ipl = r
            assert kls is None and colls is None and coll.shape == (Lv,)
            psnr = train.calc_psnr(show, img)
            assert abs(psnr - float(g["psnr"][e])) < 0.01, (e, psnr, float(g["psnr"][e]))
            close(np.array(psnr), g["psnr"][e], 0, 0.01, f"G11 headline-shape hash PSNR epoch {e} (graph={graph})")
            close(np.array(mse), g["mse"][e], 2e-3, 0, f"G11 headline-shape hash MSE epoch {e} (graph={graph})")
    finally:
        models.should_use_hash_function = False


@pytest.mark.parametrize("mode", ["hash", "gngf_frozen"])
def test_hipgraph_replay_of_a_step_equals_eager(golden, mode):
    """train.GraphedStep replays one captured forward + backward (helper stream, hint hand-off, per-call workspaces
    included): the replayed gradients must equal the eager ones, for the captured batch and for a different one."""
    from collision_handling_in_instantngp_amd import models, train
    models.should_use_hash_function = (mode == "hash")
    try:
        torch.manual_seed(1)
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=2 ** 15, num_levels=8, n_min=16, n_max=128,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=2 ** 15, feature_dim=2, topk_k=4)
        net.return_indices = False
        net.dense_probs = False
        if mode != "hash":
            for p in net.HPD.parameters():
                p.requires_grad = False
            net.compute_pbar = False
        X, Y, h, w = strawberry(golden)
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        gs = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3)
        for lo in (0, 60000):
            xy, tgt = X[lo:lo + 60000].contiguous(), Y[lo:lo + 60000].contiguous()
            net.zero_grad(set_to_none=True)
            rgb, probs, _i, _c = net(xy, 1.0)
            mse, kls, coll = loss_fn(rgb, tgt, None, probs, torch.tensor([], device=DEV), torch.tensor([], device=DEV))
            assert kls is None
            train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
            eager = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
            eager_rgb = rgb.detach().clone()
            r = gs(xy, tgt)
            r = gs(xy, tgt)
            torch.cuda.synchronize()
            assert torch.equal(r.out, eager_rgb)
            for k, p in net.named_parameters():
                if k in eager:
                    scale = float(eager[k].abs().max()) + 1e-30
                    assert float((p.grad - eager[k]).abs().max()) <= 2e-5 * scale, k
        assert len(gs._graphs) == 1                      # one capture serves both batches
    finally:
        models.should_use_hash_function = False


def test_unrolled_graph_of_three_steps_equals_three_eager_steps(golden):
    """GraphedStep(unroll=3): forward + backward + Adam of three consecutive batches in ONE replayed graph == three eager steps
    (parameters after the third step, and the per-step outputs)."""
    import copy
    from collision_handling_in_instantngp_amd import models, train
    models.should_use_hash_function = True
    try:
        torch.manual_seed(2)
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=2 ** 14, num_levels=8, n_min=16, n_max=128,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=2 ** 14, feature_dim=2, topk_k=4)
        net.return_indices = False
        ref = copy.deepcopy(net)
        start = {k: p.detach().clone() for k, p in net.named_parameters()}
        X, Y, h, w = strawberry(golden)
        batches = [(X[lo:lo + 40000].contiguous(), Y[lo:lo + 40000].contiguous()) for lo in (0, 40000, 80000)]
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        empty = torch.tensor([], device=DEV)
        opt_ref = train.get_optimizer(ref, 1e-2, 1e-3, 1e-3, 0.0, 0.0, 1e-6)
        outs = []
        for _round in range(2):
            for xy, tgt in batches:
                opt_ref.zero_grad(set_to_none=True)
                rgb, probs, _i, _c = ref(xy, 1.0)
                mse, kls, coll = loss_fn(rgb, tgt, None, probs, empty, empty)
                train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
                opt_ref.step()
                outs.append(rgb.detach().clone())
        opt = train.get_optimizer(net, 1e-2, 1e-3, 1e-3, 0.0, 0.0, 1e-6)
        gs = train.GraphedStep(net, loss_fn, opt, 1, 1, 1e-3, unroll=3)
        got = []
        for _round in range(2):                          # the first call captures and replays; the second only replays
            rs = gs.run_many(batches)
            torch.cuda.synchronize()
            got += [r.out.clone() for r in rs]
        # Adam's first steps turn a last-bit difference of a tiny gradient (float atomics land in a different order from run
        # to run) into a difference of up to lr in that entry: compare outputs, and parameters in the mean against their movement
        assert torch.equal(got[0], outs[0])
        for i, (g_, w_) in enumerate(zip(got, outs)):
            assert float((g_ - w_).abs().max()) <= 2e-3, i
            assert float((g_ - w_).abs().mean()) <= 2e-5, i
        for (k, p), (_k, q), (_k0, q0) in zip(net.named_parameters(), ref.named_parameters(), start.items()):
            if not q.requires_grad:
                continue
            moved = float((q - q0).abs().mean())
            assert float((p - q).abs().mean()) <= 0.01 * moved, k           # parameters no loss term reaches stay put in both
        moved_any = [k for (k, q), q0 in zip(ref.named_parameters(), start.values()) if not torch.equal(q, q0)]
        assert any("hash" in k.lower() or "table" in k.lower() or "MRHE" in k for k in moved_any) and len(moved_any) >= 7, moved_any
        with pytest.raises(ValueError):
            gs(*batches[0])
    finally:
        models.should_use_hash_function = False


@pytest.mark.parametrize("K", [0, 3])
def test_distinct_slot_counts_kernel_vs_torch_unique(K):
    """csrc/stats.hip (bit-map pass) == torch.unique(...).numel() per level and top-K rank, T not a multiple of 32"""
    from collision_handling_in_instantngp_amd import models
    L, T, P = 5, 1000 + 7, 20000
    g = torch.Generator().manual_seed(11 + K)
    shape = (P, L, 4) if K == 0 else (P, L, 4, K)
    idx = torch.randint(0, T, shape, generator=g)
    idx[:, 1] = idx[:, 1] % 3                                  # a level that uses three slots only
    idx[:, 2] = torch.arange(P).reshape((P, 1) if K == 0 else (P, 1, 1)) % T   # every slot used
    got = models.GeneralNeuralGaugeFields._distinct_slot_counts(idx.to(DEV), L, T).cpu()
    Kk = max(K, 1)
    assert got.shape == (Kk, L)
    for k in range(Kk):
        for l in range(L):
            sl = idx[:, l] if K == 0 else idx[:, l, :, k]
            assert int(got[k, l]) == int(torch.unique(sl).numel()), (k, l)
    assert int(got[0, 1]) == 3 and int(got[0, 2]) == T


def test_kept_logits_equal_recomputed_logits_in_the_hpd_backward(golden):
    """ops.HPD_Z_CACHE_BYTES: chunks whose logits are kept from the forward give the same gradients as chunks that
    recompute them (several chunks, budget for some of them only, and none)"""
    from collision_handling_in_instantngp_amd import ops, models as M
    net, g, models, train = build(golden, "gngf")
    X, Y, h, w = strawberry(golden)
    sl = t(g["perm"])[:4096]
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    net.dense_probs = False
    T = net._hash_table_size
    old = (M.HPD_CHUNK_BYTES, ops.HPD_Z_CACHE_BYTES, ops.HPD_Z_CACHE_RESERVE)
    outs = []
    try:
        M.HPD_CHUNK_BYTES = 64 * 4 * T                        # 64-row chunks: many chunks at this small shape
        ops.HPD_Z_CACHE_RESERVE = 0
        for budget in (0, 3 * 64 * 4 * T, 1 << 40):           # recompute everything / keep three chunks / keep all
            ops.HPD_Z_CACHE_BYTES = budget
            net.zero_grad()
            rgb, probs, idx, _ = net(X[sl], 1 / 3)
            empty = torch.tensor([], device=DEV)
            mse, kls, coll = loss_fn(rgb, Y[sl], probs.shape[-1], probs, empty, empty)
            train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
            outs.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    finally:
        M.HPD_CHUNK_BYTES, ops.HPD_Z_CACHE_BYTES, ops.HPD_Z_CACHE_RESERVE = old
    assert any(k.startswith("HPD") for k in outs[0])
    for other in outs[1:]:
        for k in outs[0]:                                     # float atomics (split-K GEMMs) reorder sums from run to run
            scale = float(outs[0][k].abs().max()) + 1e-30
            close(other[k], outs[0][k].cpu().numpy(), 1e-4, 2e-5 * scale, k)


@pytest.mark.parametrize("fp32_grads", [False, True])
def test_fp16_table_model_trains_with_fused_adam_like_the_fp32_master_model(golden, fp32_grads):
    """cfg5 flavour end to end (F = 4, fp16 table storage): get_optimizer returns FusedAdam, whose fp32 master copy follows
    the trajectory of an fp32 model started from the same (fp16-representable) tables; MultiResHashEncoding.forward accepts
    the fp16 tables at the per-instance boundary.  fp32_grads: ops.FP16_TABLE_GRAD_FP32 — the table gradient reaches the
    optimizer as the fp32 buffer it was accumulated in (`param.grad_fp32`, no `.grad`) instead of an fp16 copy."""
    from collision_handling_in_instantngp_amd import models, train, ops
    X, Y, h, w = strawberry(golden)
    models.should_use_hash_function = True
    ops.FP16_TABLE_GRAD_FP32 = fp32_grads
    try:
        kw = dict(input_dim=2, hash_table_size=2 ** 14, num_levels=8, n_min=16, n_max=256, MLP_hidden_layers_widths=[64, 64],
                  HPD_hidden_layers_widths=[32, 64, 128], HPD_out_features=2 ** 14, feature_dim=4, topk_k=4)
        torch.manual_seed(3)
        n16 = models.GeneralNeuralGaugeFields(**kw, table_dtype=torch.float16)
        with torch.no_grad():
            for m in n16.encoding._hash_tables:
                m.weight.mul_(100.0)
        n32 = models.GeneralNeuralGaugeFields(**kw)
        n32.load_state_dict({k: (v.float() if "hash_tables" in k else v) for k, v in n16.state_dict().items()})
        o16 = train.get_optimizer(n16, 1e-3, 1e-3, 1e-3, 0, 1e-6, 1e-6)
        o32 = train.get_optimizer(n32, 1e-3, 1e-3, 1e-3, 0, 1e-6, 1e-6)
        assert isinstance(o16, train.FusedAdam) and isinstance(o32, train.FusedAdam)
        scale = 4096.0
        o16.grad_scale = scale
        xb, yb = X[:40000], Y[:40000]
        losses = {16: [], 32: []}
        for _ in range(5):
            for net, opt, sc, tag in ((n16, o16, scale, 16), (n32, o32, 1.0, 32)):
                opt.zero_grad()
                rgb, _, _, _ = net(xb, 1.0)
                loss = torch.nn.functional.mse_loss(rgb, yb)
                (loss * sc).backward()
                if tag == 16:
                    w5 = net.encoding._hash_tables[5].weight
                    if fp32_grads:
                        assert w5.grad is None and w5.grad_fp32.dtype == torch.float32 and w5.grad_fp32.shape == w5.shape
                        assert float(w5.grad_fp32.abs().max()) > 0
                    else:
                        assert w5.grad.dtype == torch.float16 and getattr(w5, "grad_fp32", None) is None
                opt.step()
                losses[tag].append(float(loss))
        if fp32_grads:
            o16.zero_grad()
            assert n16.encoding._hash_tables[5].weight.grad_fp32 is None
        assert losses[16][-1] < losses[16][0]
        close(np.array(losses[16]), np.array(losses[32]), 2e-3, 0, "fp16-table model vs fp32 model: MSE over 5 Adam steps")
        w16 = n16.encoding._hash_tables[5].weight
        master = o16.state[w16]["master"]
        assert w16.dtype == torch.float16 and torch.equal(w16.detach(), master.half())
        ref = n32.encoding._hash_tables[5].weight.detach()
        moved = (ref - n16.state_dict()["encoding._hash_tables.5.weight"].float()).abs().max()
        close(master, ref.cpu().numpy(), 0, 2e-3 * 5, "fp32 master of the fp16 tables vs fp32 tables after 5 steps (lr 1e-3)")
        # the per-instance module boundary on fp16 tables
        idx = torch.randint(0, 2 ** 14, (512, 8, 4), device=DEV)
        feats = n16.encoding(idx, None)
        assert feats.shape == (512, 4, 8, 4) and feats.dtype == torch.float32
        want = torch.stack([n16.encoding._hash_tables[l].weight.detach().float()[idx[:, l]] for l in range(8)], 1)   # (P,L,4,F)
        assert torch.equal(feats, want.permute(0, 3, 1, 2))
    finally:
        models.should_use_hash_function = False
        ops.FP16_TABLE_GRAD_FP32 = False


def test_three_models_interleaved_do_not_share_state(golden):
    """Every piece of per-step state lives on the model / the autograd context (ops.DataParallel, ops.StepLink, the fp32
    gradient sink), not in the process: three live models — fp32 tables, fp16 tables with the fp32 gradient hand-over, and one
    with a deferred vertex stage (data parallel) — run their forward and backward passes INTERLEAVED, each inside
    fused_mse(target, gloss=1.0) (decoder slab reduction riding on the encoder backward), and every model's gradients equal the
    ones it produces alone."""
    from collision_handling_in_instantngp_amd import models, ops, train
    models.should_use_hash_function = True
    try:
        gen = torch.Generator().manual_seed(3)
        P = 2 ** 15
        xs = [torch.rand((P, 2), generator=gen).to(DEV) for _ in range(3)]
        ys = [torch.rand((P, 3), generator=gen).to(DEV) for _ in range(3)]

        def make(kind):
            torch.manual_seed(11)
            net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=2 ** 14, num_levels=16, n_min=16, n_max=256,
                                                  MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                                  HPD_out_features=2 ** 14, feature_dim=2, topk_k=4,
                                                  table_dtype=(torch.float16 if kind == "fp16" else torch.float32))
            net.return_indices = False
            if kind == "fp16":
                with torch.no_grad():
                    for m in net.encoding._hash_tables:
                        m.weight.mul_(100.0)
                net.encoding.grad_fp32_handover = True
            if kind == "deferred":
                net.dp.exchange = lambda tns: None            # a world of one: the exchange leaves dG as it is
                net.dp.defer_vertex = True
            return net

        def fwd(net, k):
            with net.fused_mse(ys[k], gloss=1.0):
                rgb, *_ = net(xs[k], 1.0)
            return ops.mse_loss(rgb, ys[k])

        def finish(net, kind):
            if kind == "deferred":
                assert net.dp.deferred is not None and net.dp.tables_reduced == 16
                assert ops.run_deferred_vertex_stage(net.dp)
            else:
                assert net.dp.deferred is None
            torch.cuda.synchronize()
            out = {}
            for n_, p_ in net.named_parameters():
                gr = p_.grad if p_.grad is not None else getattr(p_, "grad_fp32", None)
                if gr is not None:
                    out[n_] = gr.detach().float().clone()
            return out

        kinds = ("fp32", "fp16", "deferred")
        alone = []
        for k, kind in enumerate(kinds):
            net = make(kind)
            fwd(net, k).backward()
            alone.append(finish(net, kind))
            del net
        nets = [make(kind) for kind in kinds]
        losses = [fwd(net, k) for k, net in enumerate(nets)]          # three forward passes in flight
        for k in (2, 0, 1):                                           # backward passes in another order
            losses[k].backward()
        for k, kind in enumerate(kinds):
            got = finish(nets[k], kind)
            assert set(got) == set(alone[k]) and len(got) >= 16 + 6, (kind, sorted(got))
            assert (kind == "fp16") == (nets[k].encoding._hash_tables[0].weight.grad is None)      # handed over as grad_fp32
            for n_ in got:
                scale = float(alone[k][n_].abs().max())
                assert scale > 0, (kind, n_)
                close(got[n_], alone[k][n_].cpu().numpy(), 0, 1e-5 * scale, f"interleaved == alone ({kind}: {n_})")
    finally:
        models.should_use_hash_function = False


def test_learning_step_gradients_do_not_depend_on_how_the_hpd_backward_is_run(golden):
    """Round 5: at a shape whose chunks are whole 128-row tiles (T = 2^14, last hidden width 128, 128-row chunks: many chunks + a ragged
    last one) one learning step gives the same gradients with the d-logits formed inside the dW / dh GEMMs (gngf_hpd_bwd_dot +
    gngf_hpd_bwd_prepare + gngf_hpd_bwd_fused) as with the three separate entry points, chunks pipelined over two streams or one after
    the other, two planes or three — the paths `ops.HpdVertexFunction.backward` chooses between (reference: autograd through
    models.py:84-85,105-123 and utils.py:138,159)."""
    from collision_handling_in_instantngp_amd import ops, models as M, train
    X, Y, h, w = strawberry(golden)
    M.should_use_hash_function = False
    kw = dict(input_dim=2, hash_table_size=2 ** 14, num_levels=8, n_min=16, n_max=128, MLP_hidden_layers_widths=[64, 64],
              HPD_hidden_layers_widths=[32, 64, 128], HPD_out_features=2 ** 14, feature_dim=2, topk_k=4)
    torch.manual_seed(11)
    net = M.GeneralNeuralGaugeFields(**kw).to(DEV)
    net.dense_probs = False
    T = 2 ** 14
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    xb, yb = X[:30000], Y[:30000]
    old = (M.HPD_CHUNK_BYTES, ops.HPD_BWD_FUSED, ops.HPD_PIPELINE, ops.HPD_BWD_TWO_PLANES)
    outs = {}
    try:
        M.HPD_CHUNK_BYTES = 128 * 4 * T
        for fused, pipe, two in ((0, 1, 1), (1, 1, 1), (1, 0, 1), (1, 1, 0), (0, 0, 1)):
            ops.HPD_BWD_FUSED, ops.HPD_PIPELINE, ops.HPD_BWD_TWO_PLANES = bool(fused), bool(pipe), bool(two)
            net.zero_grad()
            rgb, probs, idx, _ = net(xb, 1 / 3)
            empty = torch.tensor([], device=DEV)
            mse, kls, coll = loss_fn(rgb, yb, probs.shape[-1], probs, empty, empty)
            train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
            torch.cuda.synchronize()
            outs[(fused, pipe, two)] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
            st = net.hpd_stats
            assert st["chunks"] >= 3 and st["rows_total"] % 128 != 0, st        # several whole chunks and a ragged one
    finally:
        M.HPD_CHUNK_BYTES, ops.HPD_BWD_FUSED, ops.HPD_PIPELINE, ops.HPD_BWD_TWO_PLANES = old
    ref = outs[(0, 1, 1)]
    assert any(k.startswith("HPD") for k in ref)
    for key, other in outs.items():
        for k in ref:
            scale = float(ref[k].abs().max()) + 1e-30
            close(other[k], ref[k].cpu().numpy(), 1e-4, 2e-5 * scale, f"{key} {k}")
