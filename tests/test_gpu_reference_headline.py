"""GPU parity against outputs of the REFERENCE ITSELF at the shapes the headline is quoted on (goldens G12-G14, written by
oracle/make_goldens.py in the build container):
  G12  GNGF indexing at L=16, F=2, T=2^19, K=4, N 16->512 on 8 strawberry pixels (reference models.py:90-123, 5-19, 394-484):
       top-K membership per vertex, top-K probabilities, rgb, loss terms, p-bar, every gradient
  G13  the reference's five checkpoint files (functions.py:761-781) loaded and resumed; the -hwp constructor path
       (models.py:364-371): frozen HPD from HPD_model.pt, cached per-vertex table and its invalidation
  G14  cfg3's own pixels (macaw.jpg), hash indexing at the headline table shape
Large tensors (64 MiB tables, the 256 MiB last HPD layer) are regenerated from the seeded CPU generators the golden script used."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, parity_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
SEED = 65535


def t(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


def image_xy(img):
    h, w = img.shape[:2]
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    X = torch.tensor(np.stack([rows, cols], -1).reshape(-1, 2)).float() / (max(w, h) - 1)   # main.py:50-51
    Y = torch.tensor(img.reshape(-1, 3) / 255).float()                                      # utils.py:61
    return X.to(DEV), Y.to(DEV), h, w


def seeded_tables(L, T, Fd, seed):
    gen = torch.Generator().manual_seed(seed)
    return (torch.rand((L, T, Fd), generator=gen) * 2 - 1) * 1e-4


def seeded_hpd_last_layer(T, fan_in=128, seed=SEED + 12):
    gen = torch.Generator().manual_seed(seed)
    bound = 1.0 / np.sqrt(fan_in)
    W = (torch.rand((T, fan_in), generator=gen) * 2 - 1) * bound
    b = (torch.rand((T,), generator=gen) * 2 - 1) * bound
    return W, b


def headline_net(models, T=2 ** 19, L=16, Fd=2, K=4, **kw):
    return models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=16, n_max=512,
                                           MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                           HPD_out_features=T, feature_dim=Fd, topk_k=K, **kw)


@pytest.mark.parametrize("fused_train", [False, True])
def test_gngf_headline_shape_matches_reference_T19(golden, fused_train, encode_path):
    """fused_train: the same step inside net.fused_mse(target, gloss=1.0), i.e. through gngf_decoder_train (forward + MSE
    gradient + backward of the decoder in ONE launch — the kernel bench.py times): the reference's own gradients check it
    (models.py:469-470, utils.py:99)."""
    import contextlib
    from collision_handling_in_instantngp_amd import models, ops, train, _lib
    g = golden("G12_gngf_T19_reference")
    L, T, Fd, K = 16, 2 ** 19, 2, 4
    models.should_use_hash_function = False
    net = headline_net(models)
    sd = net.state_dict()
    tabs = seeded_tables(L, T, Fd, SEED + 13)
    for l in range(L):
        sd[f"encoding._hash_tables.{l}.weight"] = tabs[l].to(DEV)
    W, b = seeded_hpd_last_layer(T)
    sd["HPD.module_list.3.0.weight"], sd["HPD.module_list.3.0.bias"] = W.to(DEV), b.to(DEV)
    for k in list(sd):
        gk = "init_" + k.replace(".", "_")
        if gk in g:
            sd[k] = t(g[gk])
    net.load_state_dict(sd)
    net.dense_probs = "auto"           # (P,L,4,T) is 1 GiB, but the per-vertex rows would be 340 GiB: must come back compact
    X, Y, h, w = image_xy(golden("strawberry_rgb")["img"])
    sel = t(g["sel"])
    x = X[sel]
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    net.zero_grad()
    ysel = Y[sel].contiguous()
    _lib.PROFILE = {}
    try:
        with (net.fused_mse(ysel, gloss=1.0) if fused_train else contextlib.nullcontext()):
            rgb, probs, idx, counts = net(x, 1.0, should_calc_counts=False)
        assert isinstance(probs, models.VertexDistribution) and tuple(probs.shape) == (8, L, 4, T)
        empty = torch.tensor([], device=DEV)
        mse, kls, coll = loss_fn(rgb, ysel, probs.shape[-1], probs, empty, empty)
        loss = train.assemble_loss(mse, kls, coll, 1, 1, 1e-3)
        loss.backward()
        torch.cuda.synchronize()
        names = set(_lib.PROFILE)
    finally:
        _lib.PROFILE = None
    assert ("gngf_decoder_train" in names) == fused_train and ("gngf_decoder_bwd" in names) == (not fused_train), names
    assert ("gngf_encode_tiled_bwd" in names) == (encode_path.path == "tiled"), names
    encode_path.assert_chain()          # "tiled": tiled_bwd_il -> dG64 -> dg64_to_float -> vertex_bwd_sorted (d w needed: trainable HPD)

    # --- top-K membership per (pixel, level, corner): equal as a set unless the reference's own K-th / (K+1)-th gap is a tie
    ref_idx, ref_tp, nxt = g["topk_idx"].astype(np.int64), g["topk_probs"], g["next_prob"]
    got_idx = idx.cpu().numpy()
    assert got_idx.shape == ref_idx.shape and idx.dtype == torch.int64
    same_set = (np.sort(got_idx, -1) == np.sort(ref_idx, -1)).all(-1)
    rel_gap = (ref_tp[..., K - 1] - nxt) / ref_tp[..., K - 1]
    assert np.all(same_set | (rel_gap < 1e-5)), f"{(~same_set).sum()} top-K sets differ outside ties"
    ordered = (got_idx == ref_idx).all(-1)
    inner_gap = np.min((ref_tp[..., :-1] - ref_tp[..., 1:]) / ref_tp[..., :-1], -1)
    assert np.all(ordered | (inner_gap < 1e-5) | ~same_set)
    print(f"[G12] top-K sets equal: {same_set.mean():.4f}, same order: {ordered.mean():.4f}, smallest relative gap at the K boundary {rel_gap.min():.2e}")
    # --- top-K probabilities (per-vertex table expanded to the reference's (P,L,4,K) layout)
    n_ls = net._n_ls_flat(x.device)
    vstride, NV = net._vertex_extent(x)
    _, _, tp = ops.expand_vertex_table(x, n_ls, vstride, NV, src_val=probs.topk_probs.detach())
    m = same_set & ordered
    # p = exp(z - max) / sum: a relative error of p IS an absolute error of the logit difference, and the logits of far
    # vertices are in the hundreds (raw integer coordinates up to 512 enter the HPD): 1e-4 relative = 1e-4 absolute on
    # z ~ 3e2, i.e. a few fp32 ulps of a 128-term dot product summed in a different order than the reference's MKL GEMM
    # VERDICT r4 item 6 — the one tolerance of the suite that stood above north_star's 1e-5 (rtol 2e-4 until round 4), now pinned
    # to the reference itself (golden G12_spread, oracle/make_goldens.py::g12s):
    #   * the reference's own spread over 8 threads / 1 thread / mkldnn off is EXACTLY 0 at this shape (so no slack comes from there);
    #   * the same forward pass through the reference's modules in float64 gives the exact top-K probabilities for these weights,
    #     and the fp32 reference is 1.45e-5 (max-abs) away from them: its sequential fp32 accumulation of the 128-term logit
    #     products, |z| ~ 70, one ulp = 7.6e-6.
    # So: against the EXACT values the HIP path must meet north_star's 1e-5 outright, and against the fp32 reference it may differ by
    # what the reference itself differs from exact plus 1e-5 (triangle inequality) — no more.
    sp = golden("G12_spread")
    assert float(sp["topk_probs_spread_abs"]) == 0.0 and bool(sp["ordered_topk_identical"])
    got_tp = tp.cpu().numpy()
    m64 = m & (sp["fp64_topk_idx"].astype(np.int64) == ref_idx).all(-1)
    assert m64.mean() > 0.99
    parity_close(got_tp[m64], sp["fp64_topk_probs"][m64], 0, 1e-5, "G12 top-K probabilities vs the reference evaluated in float64 (T=2^19, atol 1e-5)")
    own = float(sp["ref_fp32_vs_fp64_topk_probs_abs"])
    assert own < 3e-5, own
    parity_close(got_tp[m], ref_tp[m], 0, max(1e-5, 2 * float(sp["topk_probs_spread_abs"]), own + 1e-5),
                 "G12 top-K probabilities vs the fp32 reference (atol = the reference's own distance from float64 + 1e-5)")
    # --- outputs and loss terms
    parity_close(rgb, g["rgb"], 0, 1e-5, "G12 rgb")
    parity_close(mse, g["mse"], 1e-5, 0, "G12 mse")
    parity_close(kls, g["kls"], 1e-2, 2e-8, "G12 JS/KL term per level")
    parity_close(loss, g["loss"], 1e-5, 0, "G12 loss")
    slots = t(g["slots"])
    pbar = probs.pbar.detach()
    parity_close(pbar[:, slots], g["pbar_at_slots"], 1e-4, 1e-10, "G12 p-bar at top-K + sampled slots")
    # (the reference's own fp32 softmax rows sum to 1 +- 2e-5 at T = 2^19; the streaming form here stays within 1.2e-6 of 1)
    parity_close(pbar.double().sum(1), g["pbar_rowsum"], 5e-5, 0, "G12 p-bar row sums")
    assert float((pbar.double().sum(1) - 1).abs().max()) < 5e-6
    parity_close(pbar.max(1).values, g["pbar_max"], 1e-4, 0, "G12 p-bar row maxima")
    # --- gradients
    for k_, p_ in net.named_parameters():
        key = "grad_" + k_.replace(".", "_")
        if key in g:
            scale = float(np.abs(g[key]).max()) + 1e-30
            parity_close(p_.grad, g[key], 1e-3, 1e-4 * scale, "G12 " + key)
    for name in ("weight", "bias"):
        p_ = getattr(net.HPD.module_list[3][0], name)
        key = f"grad_HPD_module_list_3_0_{name}"
        scale = float(np.abs(g[key + "_at_slots"]).max()) + 1e-30
        parity_close(p_.grad[slots], g[key + "_at_slots"], 1e-3, 1e-4 * scale, "G12 " + key + " at slots")
        parity_close(p_.grad.double().abs().sum(), g[key + "_abs_sum"], 1e-3, 0, "G12 " + key + " abs sum")
    dt = torch.stack([net.encoding._hash_tables[l].weight.grad for l in range(L)]).cpu().numpy()
    want = np.zeros((L, T, Fd), np.float32)
    nz = g["dtables_nz"]
    want[nz[:, 0], nz[:, 1]] = g["dtables_val"]
    if same_set.all():
        parity_close(dt, want, 1e-3, 1e-5 * float(np.abs(want).max()), "G12 table gradient (all 16 x 2^19 rows)")
    else:                                   # a tie resolved differently moves gradient between the tied rows: compare the rest
        tied = np.unique(np.concatenate([got_idx[~same_set].ravel(), ref_idx[~same_set].ravel()]))
        keep = np.ones(T, bool)
        keep[tied] = False
        parity_close(dt[:, keep], want[:, keep], 1e-3, 1e-5 * float(np.abs(want).max()), "G12 table gradient (untied rows)")


def test_reference_written_checkpoint_loads_and_resumes(golden):
    """whole_model.pt + whole_opt.pt written by the reference's torch.save calls (functions.py:768-769) load into this
    package's model and FusedAdam; the resumed step lands on the reference's parameters."""
    from collision_handling_in_instantngp_amd import data, models, train
    g = golden("G13_hwp_and_checkpoints")
    folder = os.path.join(GOLDEN, "ref_ckpt_cfg1")
    models.should_use_hash_function = False
    net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=256, num_levels=4, n_min=8, n_max=32,
                                          MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                          HPD_out_features=256, feature_dim=2, topk_k=4)
    opt = train.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
    assert isinstance(opt, train.FusedAdam)
    data.load_checkpoint(net, folder, optimizer=opt, parts=("model",), map_location=DEV)
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    X, Y, h, w = image_xy(golden("strawberry_rgb")["img"])
    sl = t(g["perm"])[2 * 4096:3 * 4096]
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    opt.zero_grad()
    rgb, probs, idx, _ = net(X[sl], 1 / 3)
    empty = torch.tensor([], device=DEV)
    mse, kls, coll = loss_fn(rgb, Y[sl], probs.shape[-1], probs, empty, empty)
    train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    opt.step()
    parity_close(rgb, g["resume_rgb"], 0, 1e-5, "G13 resumed rgb")
    parity_close(mse, g["resume_mse"], 1e-5, 0, "G13 resumed mse")
    for k_, p_ in net.named_parameters():
        key = "resume_param_" + k_.replace(".", "_")
        if key not in g or k_ not in grads:
            continue
        lr = 1e-4 if "hash_tables" in k_ else 1e-3
        gk = grads[k_]
        mask = (gk.abs() > 1e-2 * gk.abs().max()).cpu().numpy()        # entries whose gradient is well above round-off
        assert mask.mean() > 0.01, k_
        moved = (p_.detach() - before[k_]).cpu().numpy()
        want_moved = g[key] - before[k_].cpu().numpy()
        assert np.abs(want_moved[mask]).max() > 0.2 * lr, k_             # the reference did move these entries
        parity_close(moved[mask], want_moved[mask], 0, 0.05 * lr, f"G13 Adam update after resume, {k_} (lr {lr:g})")
    # the per-module files load as well, in either direction
    net2 = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=256, num_levels=4, n_min=8, n_max=32,
                                           MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                           HPD_out_features=256, feature_dim=2, topk_k=4)
    data.load_checkpoint(net2, folder, parts=("encoding", "HPD", "mlp"), map_location=DEV)
    ref_sd = torch.load(os.path.join(folder, "whole_model.pt"), map_location=DEV)
    for k, v in net2.state_dict().items():
        if not k.startswith("_batch_norm"):
            assert torch.equal(v, ref_sd[k]), k
    assert set(net2.state_dict()) == set(ref_sd)


def test_hwp_constructor_path_frozen_hpd_matches_reference_and_caches_its_table(golden):
    """GeneralNeuralGaugeFields(HPD_weights_path=...) (reference models.py:364-371) from the reference-written HPD_model.pt."""
    from collision_handling_in_instantngp_amd import models, train
    g = golden("G13_hwp_and_checkpoints")
    path = os.path.join(GOLDEN, "ref_ckpt_cfg1", "HPD_model.pt")
    models.should_use_hash_function = False
    net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=256, num_levels=4, n_min=8, n_max=32,
                                          MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                          HPD_out_features=256, feature_dim=2, topk_k=4, HPD_weights_path=path)
    assert net.hpd_is_frozen() and all(not p.requires_grad for p in net.HPD.parameters())
    ref_hpd = torch.load(path, map_location=DEV)
    for k, v in net.HPD.state_dict().items():
        assert torch.equal(v, ref_hpd[k]), k
    sd = net.state_dict()
    for k in list(sd):
        gk = "hwp_init_" + k.replace(".", "_")
        if gk in g:
            sd[k] = t(g[gk])
    net.load_state_dict(sd)
    X, Y, h, w = image_xy(golden("strawberry_rgb")["img"])
    sl = t(g["perm"])[3 * 4096:4 * 4096]
    loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
    empty = torch.tensor([], device=DEV)
    rgb, probs, idx, _ = net(X[sl], 1 / 3)
    assert probs.shape == (4096, 4, 4, 256)                      # the reference returns the dense distribution here too
    mse, kls, coll = loss_fn(rgb, Y[sl], probs.shape[-1], probs, empty, empty)
    train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
    parity_close(rgb, g["hwp_rgb"], 0, 1e-5, "G13 -hwp rgb")
    parity_close(mse, g["hwp_mse"], 1e-5, 0, "G13 -hwp mse")
    parity_close(kls, g["hwp_kls"], 1e-3, 1e-8, "G13 -hwp JS/KL")
    assert (idx.cpu().numpy() == g["hwp_idx"]).mean() > 0.995
    assert all(p.grad is None for p in net.HPD.parameters())
    for k_, p_ in net.named_parameters():
        key = "hwp_grad_" + k_.replace(".", "_")
        if key in g:
            scale = float(np.abs(g[key]).max()) + 1e-30
            parity_close(p_.grad, g[key], 1e-3, 1e-4 * scale, "G13 -hwp " + key)
    # the fast path of a frozen HPD: per-vertex table cached across steps, rebuilt when a frozen weight changes in place
    net.dense_probs = False
    net.compute_pbar = False
    rgb_a, pa, idx_a, _ = net(X[sl], 1 / 3)
    table_a = net._frozen_table
    assert table_a is not None
    assert torch.equal(idx_a, idx)
    parity_close(rgb_a, g["hwp_rgb"], 0, 1e-5, "G13 -hwp rgb from the cached per-vertex table")
    rgb_b, *_ = net(X[sl], 1 / 3)
    assert net._frozen_table is table_a and torch.equal(rgb_a, rgb_b)          # second step: no rebuild
    with torch.no_grad():
        net.HPD.module_list[3][0].bias.add_(torch.linspace(0, 5, 256, device=DEV))
    rgb_c, _, idx_c, _ = net(X[sl], 1 / 3)
    assert net._frozen_table is not table_a                                      # in-place edit -> version bump -> rebuild
    assert not torch.equal(idx_c, idx_a)
    net.HPD.load_state_dict(ref_hpd)                                             # load_state_dict copies in place too
    rgb_d, _, idx_d, _ = net(X[sl], 1 / 3)
    assert torch.equal(idx_d, idx_a) and torch.equal(rgb_d, rgb_a)


@pytest.mark.parametrize("fused_train", [False, True])
def test_macaw_hash_step_at_headline_shape_matches_reference(golden, fused_train, encode_path):
    """BASELINE configs[2]: macaw.jpg, plain spatial hash, L=16 F=2 T=2^19 — one forward + MSE + backward on 4096 of its pixels.
    fused_train: inside net.fused_mse(target, gloss=1.0) — the decoder's forward, loss gradient and backward in one launch."""
    import contextlib
    from collision_handling_in_instantngp_amd import models, train, _lib
    g = golden("G14_macaw_hash")
    img = golden("macaw_rgb")["img"]
    assert tuple(g["hw"]) == img.shape[:2]
    L, T, Fd = 16, 2 ** 19, 2
    models.should_use_hash_function = True
    try:
        net = headline_net(models)
        sd = net.state_dict()
        tabs = seeded_tables(L, T, Fd, SEED + 14) * 100.0
        for l in range(L):
            sd[f"encoding._hash_tables.{l}.weight"] = tabs[l].to(DEV)
        for k in list(sd):
            gk = "init_mlp_" + k[len("mlp."):].replace(".", "_") if k.startswith("mlp.") else None
            if gk and gk in g:
                sd[k] = t(g[gk])
        net.load_state_dict(sd)
        X, Y, h, w = image_xy(img)
        sel = t(g["sel"])
        net.zero_grad()
        ysel = Y[sel].contiguous()
        _lib.PROFILE = {}
        try:
            with (net.fused_mse(ysel, gloss=1.0) if fused_train else contextlib.nullcontext()):
                rgb, probs, idx, _ = net(X[sel], 1.0)
            mse, _, _ = train.Loss(delta=1, gamma=-2, epsilon=1)(rgb, ysel, None, None, None, None)
            mse.backward()
            torch.cuda.synchronize()
            names = set(_lib.PROFILE)
        finally:
            _lib.PROFILE = None
        assert ("gngf_decoder_train" in names) == fused_train, names
        assert ("gngf_encode_tiled_bwd" in names) == (encode_path.path == "tiled"), names
        encode_path.assert_chain()      # "tiled": tiled_bwd_il -> dG64 -> vertex_bwd_hash64, the hash mode's chain in bench.py
        ic = idx.cpu()
        chk = np.array([int(ic.sum()), int((ic * torch.arange(1, 4097)[:, None, None]).sum() % (2 ** 61 - 1))], dtype=np.int64)
        assert np.array_equal(chk, g["idx_checksum"])                            # index work: bit-exact
        parity_close(rgb, g["rgb"], 0, 1e-5, "G14 macaw rgb")
        parity_close(mse, g["mse"], 1e-5, 0, "G14 macaw mse")
        for k_, p_ in net.mlp.named_parameters():
            key = "grad_mlp_" + k_.replace(".", "_")
            scale = float(np.abs(g[key]).max()) + 1e-30
            parity_close(p_.grad, g[key], 1e-3, 1e-4 * scale, "G14 " + key)
        dt = torch.stack([net.encoding._hash_tables[l].weight.grad for l in range(L)]).cpu().numpy()
        want = np.zeros((L, T, Fd), np.float32)
        nz = g["dtables_nz"]
        want[nz[:, 0], nz[:, 1]] = g["dtables_val"]
        parity_close(dt, want, 1e-3, 1e-5 * float(np.abs(want).max()), "G14 macaw table gradient (all rows)")
    finally:
        models.should_use_hash_function = False
