"""Generate tests/golden/*.npz by RUNNING THE PYTHON REFERENCE on CPU (build container only).

TEST INFRASTRUCTURE ONLY.  Requires /root/reference (read-only).  The reference itself never
travels: only inputs and expected outputs (data) are written, as small .npz fixtures, together
with this script.  Re-run:  python oracle/make_goldens.py [--only G4,G7] [--with-train-curve]

Golden groups follow SURVEY.md §8c:
  G1 level_resolutions     models.py:305-317
  G2 corners               models.py:486-502
  G3 spatial_hash          models.py:504-528
  G4 encoding fwd/bwd      models.py:173-229   (hash + GNGF, three blend variants)
  G5 bilinear              models.py:621-655
  G6 hpd                   models.py:45-123, 5-42
  G7 end_to_end            models.py:394-484 + utils.py:91-174 + functions.py:96-127,243-281
  G8 train_curve           functions.py:139-355 (optional, slow)
  G9 decoder mlp           models.py:382-392,469-470
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SEED = 65535


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def np32(t):
    return t.detach().cpu().numpy().copy()  # copy: parameters are later updated in place


def make_net(M, mods, *, hash_mode, T, L, n_min, n_max, F=2, K=4, keep_topk=False, bw=False):
    rh.set_flag(mods, "should_use_hash_function", hash_mode)
    torch.manual_seed(SEED)
    return M.GeneralNeuralGaugeFields(
        input_dim=2, hash_table_size=T, num_levels=L, n_min=n_min, n_max=n_max,
        MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
        HPD_out_features=T, feature_dim=F, topk_k=K, should_keep_topk_only=keep_topk, should_bw=bw)


def load_strawberry():
    """PIL decode (cv2 absent).  X[:,0]=row, X[:,1]=col, row-major (utils.py:56-59); Y=RGB/255 (utils.py:61);
    x /= max(w,h)-1 (main.py:50-51)."""
    from PIL import Image
    img = np.array(Image.open(os.path.join(rh.REF_ROOT, "images", "strawberry.jpeg")).convert("RGB"))
    h, w = img.shape[:2]
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    X = torch.tensor(np.stack([rows, cols], -1).reshape(-1, 2)).float()
    Y = torch.tensor(img.reshape(-1, 3) / 255).float()
    X = X / (max(w, h) - 1)
    return img, X, Y, h, w


def edge_coords(n_rand, gen):
    x = torch.rand(n_rand, 2, generator=gen)
    edges = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.0, 1.0], [1.0, 0.0], [0.5, 0.5], [0.25, 0.75],
                          [1.0 / 8, 3.0 / 8], [1.0 / 32, 31.0 / 32], [338.0 / 507, 1.0], [506.0 / 507, 337.0 / 507],
                          [1.0 / 16, 1.0 / 512], [0.99999994, 0.99999994], [1e-8, 1e-8]])
    return torch.cat([x, edges], 0)


def g1(F_, U_, M, P_):
    mods = (F_, U_, M)
    cases = [(8, 32, 4), (16, 512, 16), (16, 2048, 16), (16, 4096, 16), (16, 8192, 16), (16, 1024, 8), (4, 4096, 12)]
    out = {}
    for i, (a, b, L) in enumerate(cases):
        net = make_net(M, mods, hash_mode=True, T=16, L=L, n_min=a, n_max=b)
        out[f"case{i}"] = np.array([a, b, L])
        out[f"n_ls{i}"] = np32(net._n_ls).reshape(-1)
        print(a, b, L, out[f"n_ls{i}"])
    out["hypercube"] = np32(net._voxels_helper_hypercube)
    save("G1_level_resolutions", **out)


def g2_g3(F_, U_, M, P_):
    mods = (F_, U_, M)
    gen = torch.Generator().manual_seed(SEED)
    x = edge_coords(256, gen)
    out = {"x": np32(x)}
    for tag, (a, b, L) in {"cfg1": (8, 32, 4), "cfg2": (16, 512, 16), "cfg4": (16, 4096, 16)}.items():
        for T in (2 ** 8, 2 ** 19, 1000):
            net = make_net(M, mods, hash_mode=True, T=T, L=L, n_min=a, n_max=b)
            scaled, grid = net._scale_to_grid(x)
            out[f"{tag}_scaled"] = np32(scaled)
            out[f"{tag}_grid"] = np32(grid)
            out[f"{tag}_hash_T{T}"] = np32(net._fast_hash(grid.int()))
    save("G2G3_corners_hash", **out)


def g4(F_, U_, M, P_):
    mods = (F_, U_, M)
    out = {}
    gen = torch.Generator().manual_seed(SEED)
    for tag, (T, L, Fd, K, Pn) in {"small": (256, 4, 2, 4, 64), "mid": (4096, 16, 2, 4, 48), "f4k3": (512, 3, 4, 3, 40)}.items():
        # hash branch (models.py:181-191)
        rh.set_flag(mods, "should_use_hash_function", True)
        torch.manual_seed(SEED)
        enc = M.MultiResHashEncoding(hash_table_size=T, num_levels=L, feature_dim=Fd, topk_k=K)
        tables = torch.stack([enc._hash_tables[l].weight.detach() for l in range(L)])
        out[f"{tag}_tables"] = np32(tables)
        idx = torch.randint(0, T, (Pn, L, 4), generator=gen, dtype=torch.int64)
        idx[: Pn // 4, :, 1] = idx[: Pn // 4, :, 0]  # forced collisions inside the batch
        g = torch.randn(Pn, Fd, L, 4, generator=gen)
        y = enc(idx, None)
        enc.zero_grad()
        y.backward(g)
        out[f"{tag}_hash_idx"] = np32(idx)
        out[f"{tag}_hash_gout"] = np32(g)
        out[f"{tag}_hash_out"] = np32(y)
        out[f"{tag}_hash_dtables"] = np32(torch.stack([enc._hash_tables[l].weight.grad for l in range(L)]))
        # GNGF branch (models.py:193-222), three blend variants
        rh.set_flag(mods, "should_use_hash_function", False)
        idxk = torch.randint(0, T, (Pn, L, 4, K), generator=gen, dtype=torch.int64)
        idxk[: Pn // 4, :, :, 1] = idxk[: Pn // 4, :, :, 0]
        probs = torch.rand(Pn, L, 4, K, generator=gen) * 0.01 + 1e-3
        gk = torch.randn(Pn, Fd, L, 4, generator=gen)
        out[f"{tag}_gngf_idx"] = np32(idxk)
        out[f"{tag}_gngf_probs"] = np32(probs)
        out[f"{tag}_gngf_gout"] = np32(gk)
        for vname, flag in (("softmax", True), ("raw", None), ("norm", False)):
            rh.set_flag(mods, "should_softmax_topk_features", flag)
            p = probs.clone().requires_grad_()
            y = enc(idxk, p)
            enc.zero_grad()
            y.backward(gk)
            out[f"{tag}_gngf_{vname}_out"] = np32(y)
            out[f"{tag}_gngf_{vname}_dtables"] = np32(torch.stack([enc._hash_tables[l].weight.grad for l in range(L)]))
            out[f"{tag}_gngf_{vname}_dprobs"] = np32(p.grad)
        rh.set_flag(mods, "should_softmax_topk_features", True)
    save("G4_encoding", **out)


def g5(F_, U_, M, P_):
    mods = (F_, U_, M)
    gen = torch.Generator().manual_seed(SEED + 5)
    out = {}
    for tag, (a, b, L, Fd) in {"cfg1": (8, 32, 4, 2), "cfg2": (16, 512, 16, 2), "f4": (16, 128, 6, 4)}.items():
        net = make_net(M, mods, hash_mode=True, T=64, L=L, n_min=a, n_max=b, F=Fd)
        x = edge_coords(64, gen)
        scaled, grid = net._scale_to_grid(x)
        feats = torch.randn(x.shape[0], Fd, L, 4, generator=gen).requires_grad_()
        y = net._bilinear_interpolate(scaled, grid, feats)
        g = torch.randn(y.shape, generator=gen)
        y.backward(g)
        out[f"{tag}_x"] = np32(x)
        out[f"{tag}_feats"] = np32(feats)
        out[f"{tag}_out"] = np32(y)
        out[f"{tag}_gout"] = np32(g)
        out[f"{tag}_dfeats"] = np32(feats.grad)
        out[f"{tag}_cfg"] = np.array([a, b, L, Fd])
    save("G5_bilinear", **out)


def hpd_state(hpd):
    return {f"hpd_{k.replace('.', '_')}": np32(v) for k, v in hpd.state_dict().items()}


def g6(F_, U_, M, P_):
    mods = (F_, U_, M)
    rh.set_flag(mods, "should_use_hash_function", False)
    _, X, _, _, _ = load_strawberry()
    net = make_net(M, mods, hash_mode=False, T=256, L=4, n_min=8, n_max=32)
    _, grid = net._scale_to_grid(X)
    verts = torch.unique(grid.permute(0, 2, 3, 1).reshape(-1, 2), dim=0)  # (U,2) fp32 integer coords
    print("unique vertices:", verts.shape)
    out = {"verts": np32(verts)}
    gen = torch.Generator().manual_seed(SEED + 6)
    for T in (256, 2048):
        dprobs = None
        for K in (1, 4, 20):
            torch.manual_seed(SEED + T)
            hpd = M.HashProbDistribution([32, 64, 128], in_features=2, out_features=T, k=K)
            # make the net less degenerate than default init so that top-K is well separated
            with torch.no_grad():
                for p in hpd.parameters():
                    p.mul_(3.0)
            vs = verts if T == 256 else verts[::4]  # keep the big-T fixture small
            out[f"T{T}_verts"] = np32(vs)
            v = vs.clone().requires_grad_()
            probs, tp, ti = hpd(v)
            dq = torch.randn(tp.shape, generator=gen)
            if dprobs is None:  # one direct-gradient tensor per T, shared by every K
                dprobs = torch.randn(probs.shape, generator=gen) * 1e-2
            hpd.zero_grad()
            (tp * dq).sum().add((probs * dprobs).sum()).backward()
            tag = f"T{T}_K{K}"
            if K == 1:
                out.update({f"T{T}_{k}": v_ for k, v_ in hpd_state(hpd).items()})
                out[f"T{T}_probs"] = np32(probs)
                out[f"T{T}_dprobs_in"] = np32(dprobs)
            out[f"{tag}_topk_probs"] = np32(tp)
            out[f"{tag}_topk_idx"] = np32(ti)
            out[f"{tag}_dq_in"] = np32(dq)
            for k_, p_ in hpd.named_parameters():
                out[f"{tag}_grad_{k_.replace('.', '_')}"] = np32(p_.grad)
    save("G6_hpd", **out)


def g9(F_, U_, M, P_):
    mods = (F_, U_, M)
    gen = torch.Generator().manual_seed(SEED + 9)
    out = {}
    for tag, (L, Fd, bw, leaky) in {"cfg1": (4, 2, False, False), "cfg2": (16, 2, False, False),
                                     "bw_leaky": (8, 4, True, True)}.items():
        rh.set_flag(mods, "should_leaky_relu", leaky)
        net = make_net(M, mods, hash_mode=True, T=16, L=L, n_min=8, n_max=64, F=Fd, bw=bw)
        with torch.no_grad():
            for p in net.mlp.parameters():
                p.mul_(2.0)
        x = (torch.randn(200, L * Fd, generator=gen) * 0.5).requires_grad_()
        y = x
        for layer in net.mlp:
            y = layer(y)
        g = torch.randn(y.shape, generator=gen)
        net.zero_grad()
        y.backward(g)
        out[f"{tag}_x"] = np32(x)
        out[f"{tag}_y"] = np32(y)
        out[f"{tag}_gy"] = np32(g)
        out[f"{tag}_dx"] = np32(x.grad)
        for k_, p_ in net.mlp.named_parameters():
            out[f"{tag}_w_{k_.replace('.', '_')}"] = np32(p_)
            out[f"{tag}_g_{k_.replace('.', '_')}"] = np32(p_.grad)
        out[f"{tag}_cfg"] = np.array([L, Fd, int(bw), int(leaky)])
    rh.set_flag(mods, "should_leaky_relu", False)
    save("G9_decoder", **out)


def g7(F_, U_, M, P_):
    """End-to-end: one model, three optimisation steps on three 4096-pixel strawberry batches, both modes."""
    mods = (F_, U_, M)
    img, X, Y, h, w = load_strawberry()
    save("strawberry_rgb", img=img)
    torch.manual_seed(SEED)
    perm = torch.randperm(h * w)
    B = 4096
    for mode in ("gngf", "hash"):
        hash_mode = mode == "hash"
        net = make_net(M, mods, hash_mode=hash_mode, T=256, L=4, n_min=8, n_max=32, K=4)
        loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)
        opt = F_.get_optimizer(net, encoding_lr=1e-4, HPD_lr=1e-3, MLP_lr=1e-3,
                               encoding_weight_decay=0, HPD_weight_decay=1e-6, MLP_weight_decay=1e-6)
        out = {"perm": perm[: 3 * B].numpy().astype(np.int64), "hw": np.array([h, w])}
        for k_, v_ in net.state_dict().items():
            out["init_" + k_.replace(".", "_")] = np32(v_)
        for step in range(3):
            sl = perm[step * B:(step + 1) * B]
            bx, by = X[sl], Y[sl]
            opt.zero_grad()
            rgb, probs, idx, counts = net(bx, 1 / 3, should_calc_counts=False)
            mse, kls, coll = loss_fn(rgb, by, None if hash_mode else probs.shape[-1], probs,
                                     None if hash_mode else torch.tensor([]), None if hash_mode else torch.tensor([]))
            loss = 1 * mse
            if not hash_mode:  # functions.py:243-245 with empty previous_collisions -> "+1" per level
                loss = loss + ((1 * kls) + (1e-3 * coll if coll.nelement() != 0 else 1)).sum(0)
            loss.backward()
            s = f"s{step}_"
            out[s + "rgb"] = np32(rgb)
            out[s + "idx"] = np32(idx)
            out[s + "mse"] = np32(mse)
            out[s + "loss"] = np32(loss)
            if not hash_mode:
                out[s + "kls"] = np32(kls)
                out[s + "pbar"] = np32(probs.sum(0).sum(1) / (probs.shape[0] * probs.shape[2]))  # (L,T)
                tp, ti = torch.topk(probs, 4, dim=-1)
                out[s + "topk_probs"] = np32(tp)
            for k_, p_ in net.named_parameters():
                if p_.grad is not None:
                    out[s + "grad_" + k_.replace(".", "_")] = np32(p_.grad)
            opt.step()
            for k_, p_ in net.named_parameters():
                out[s + "param_" + k_.replace(".", "_")] = np32(p_)
        save(f"G7_end_to_end_{mode}", **out)


def g8(F_, U_, M, P_):
    """First 3 epochs of cfg1 through the reference's own train_step (slow: ~5 min)."""
    mods = (F_, U_, M)
    rh.set_flag(mods, "should_use_hash_function", False)
    img, X, Y, h, w = load_strawberry()
    torch.manual_seed(SEED)
    shape = h * w
    shuffled = torch.randperm(shape).int()
    reordered = torch.zeros((shape,)).int()
    reordered[shuffled] = torch.arange(shape).int()
    cfg = F_.get_grid_search_configs(P_.grid_search_configs)[4061]
    print(len(F_.get_grid_search_configs(P_.grid_search_configs)), cfg)
    net = make_net(M, mods, hash_mode=False, T=256, L=4, n_min=8, n_max=32, K=cfg["topk_k"])
    loss_fn = U_.Loss(delta=1, gamma=cfg["loss_gamma"], epsilon=1)
    opt = F_.get_optimizer(net, 1e-4, cfg["HPD_lr"], cfg["MLP_lr"], 0, 1e-6, 1e-6)
    prev_c, prev_m = torch.tensor([]), torch.tensor([])
    rec = {"cfg_keys": np.array(list(cfg.keys())), "cfg_vals": np.array([float(v) for v in cfg.values()])}
    mses, psnrs, losses = [], [], []
    for e in range(3):
        r = F_.train_step(net, loss_fn, opt, X.clone(), Y.clone(), w, h, 256, cfg["topk_k"], cfg["l_mse"],
                          cfg["l_js_kl"], cfg["l_collisions"], batch_percentage=P_.batch_size, num_levels=4,
                          should_shuffle=True, shuffled_indices=shuffled, reordered_indices=reordered,
                          previous_collisions=prev_c, previous_min_possible_collisions=prev_m)
        loss_item, show, prev_c, prev_m, _, mse, kls, colls, _ = r
        psnr = F_.calc_psnr(show, img)
        print(e, loss_item, mse, psnr)
        mses.append(mse); psnrs.append(psnr); losses.append(loss_item)
    rec.update(mse=np.array(mses), psnr=np.array(psnrs), loss=np.array(losses), shuffled=shuffled.numpy())
    save("G8_train_curve", **rec)


def g10(F_, U_, M, P_):
    """Diagnostics contract: counts_per_level (models.py:530-566) and calc_hash_collisions (models.py:568-619)."""
    mods = (F_, U_, M)
    img, X, Y, h, w = load_strawberry()
    torch.manual_seed(SEED)
    sl = torch.randperm(h * w)[:4096]
    out = {"sel": sl.numpy().astype(np.int64)}
    for mode in ("hash", "gngf"):
        net = make_net(M, mods, hash_mode=(mode == "hash"), T=256, L=4, n_min=8, n_max=32, K=4)
        if mode == "gngf":
            for k_, v_ in net.state_dict().items():
                if k_.startswith("HPD."):
                    out["gngf_init_" + k_.replace(".", "_")] = np32(v_)
        rgb, probs, idx, counts = net(X[sl], 1 / 3, should_calc_counts=True)
        for l, c in enumerate(counts):
            ks = np.array(sorted(c.keys()), dtype=np.int64)
            out[f"{mode}_counts_keys_{l}"] = ks
            out[f"{mode}_counts_vals_{l}"] = np.array([c[k] for k in ks], dtype=np.int64)
        coll, minc = net.calc_hash_collisions(idx)
        out[f"{mode}_collisions"] = np32(coll)
        out[f"{mode}_min_collisions"] = np32(minc)
        out[f"{mode}_idx"] = np32(idx)
    save("G10_diagnostics", **out)


def seeded_tables(L, T, Fd, seed=SEED + 11):
    """Level tables too large for a fixture (64 MiB at T=2^19) are regenerated from a seeded CPU generator on both sides."""
    gen = torch.Generator().manual_seed(seed)
    return (torch.rand((L, T, Fd), generator=gen) * 2 - 1) * 1e-4


def g11(F_, U_, M, P_):
    """Headline shape in hash mode (BASELINE configs[2] flavour on strawberry): L=16, F=2, T=2^19, N 16->512, two epochs
    of three 1/3-image batches through the reference's own train_step; MSE / PSNR trajectory."""
    mods = (F_, U_, M)
    rh.set_flag(mods, "should_use_hash_function", True)
    img, X, Y, h, w = load_strawberry()
    L, T, Fd = 16, 2 ** 19, 2
    net = make_net(M, mods, hash_mode=True, T=T, L=L, n_min=16, n_max=512, F=Fd)
    tabs = seeded_tables(L, T, Fd)
    with torch.no_grad():
        for l in range(L):
            net.encoding._hash_tables[l].weight.copy_(tabs[l])
    out = {}
    for k_, v_ in net.mlp.state_dict().items():
        out["init_mlp_" + k_.replace(".", "_")] = np32(v_)
    torch.manual_seed(SEED)
    shape = h * w
    shuffled = torch.randperm(shape).int()
    reordered = torch.zeros((shape,)).int()
    reordered[shuffled] = torch.arange(shape).int()
    loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)
    opt = F_.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
    mses, psnrs = [], []
    for e in range(2):
        r = F_.train_step(net, loss_fn, opt, X.clone(), Y.clone(), w, h, T, 4, 1, 1, 1e-3, batch_percentage=P_.batch_size,
                          num_levels=L, should_shuffle=True, shuffled_indices=shuffled, reordered_indices=reordered)
        loss_item, show, _c, _m, _cnt, mse, _kl, _cl, _ipl = r
        psnr = F_.calc_psnr(show, img)
        print(e, mse, psnr)
        mses.append(mse); psnrs.append(psnr)
    out.update(mse=np.array(mses), psnr=np.array(psnrs), shuffled=shuffled.numpy())
    rh.set_flag(mods, "should_use_hash_function", False)
    save("G11_hash_L16_T19_curve", **out)


def seeded_hpd_last_layer(T, fan_in=128, seed=SEED + 12):
    """The (T, 128) last HPD layer is 256 MiB at T=2^19 — too large for a fixture: both sides regenerate it from a seeded CPU
    generator with nn.Linear's default bound 1/sqrt(fan_in)."""
    gen = torch.Generator().manual_seed(seed)
    bound = 1.0 / np.sqrt(fan_in)
    W = (torch.rand((T, fan_in), generator=gen) * 2 - 1) * bound
    b = (torch.rand((T,), generator=gen) * 2 - 1) * bound
    return W, b


def load_image(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(rh.REF_ROOT, "images", name)).convert("RGB"))


def g12(F_, U_, M, P_):
    """HEADLINE shape, GNGF indexing, through the reference itself (models.py:90-123, 5-19, 394-484): L=16, F=2, T=2^19,
    K=4, N 16->512, P=8 strawberry pixels, one forward + Loss + backward (~35 s, ~12 GiB).  Pins HPD / top-K at T=2^19.
    Large tensors are stored sparsely / sampled: table gradients as (level, slot, value) triples, the last HPD layer's
    gradient and p-bar at the slots any top-K touches plus a seeded sample of other slots."""
    mods = (F_, U_, M)
    img, X, Y, h, w = load_strawberry()
    L, T, Fd, K, Pn = 16, 2 ** 19, 2, 4, 8
    net = make_net(M, mods, hash_mode=False, T=T, L=L, n_min=16, n_max=512, F=Fd, K=K)
    tabs = seeded_tables(L, T, Fd, SEED + 13)
    W_last, b_last = seeded_hpd_last_layer(T)
    with torch.no_grad():
        for l in range(L):
            net.encoding._hash_tables[l].weight.copy_(tabs[l])
        net.HPD.module_list[3][0].weight.copy_(W_last)
        net.HPD.module_list[3][0].bias.copy_(b_last)
    out = {"cfg": np.array([L, T, Fd, K, 16, 512])}
    for k_, v_ in net.state_dict().items():
        if k_.startswith("mlp.") or (k_.startswith("HPD.") and not k_.startswith("HPD.module_list.3.")):
            out["init_" + k_.replace(".", "_")] = np32(v_)
    gen = torch.Generator().manual_seed(SEED + 12)
    sel = torch.randperm(h * w, generator=gen)[:Pn]
    sel[0] = 0                                   # the (0,0) corner pixel: vertices (0,0),(1,0),(0,1),(1,1) at every level
    sel[1] = h * w - 1                           # the far corner
    out["sel"] = sel.numpy().astype(np.int64)
    bx, by = X[sel], Y[sel]
    loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)
    net.zero_grad()
    import time
    t0 = time.time()
    rgb, probs, idx, counts = net(bx, 1.0, should_calc_counts=False)
    mse, kls, coll = loss_fn(rgb, by, probs.shape[-1], probs, torch.tensor([]), torch.tensor([]))
    loss = 1 * mse + ((1 * kls) + 1).sum(0)      # functions.py:243-245 with empty previous_collisions
    loss.backward()
    print("reference fwd+loss+bwd at T=2^19, P=8: %.1f s" % (time.time() - t0))
    out["rgb"] = np32(rgb)
    out["mse"] = np32(mse)
    out["kls"] = np32(kls)
    out["loss"] = np32(loss)
    out["topk_idx"] = np32(idx).astype(np.int32)                       # (P,L,4,K)
    tp, ti = torch.topk(probs.detach(), K + 1, dim=-1)
    assert torch.equal(ti[..., :K], idx)
    out["topk_probs"] = np32(tp[..., :K])
    out["next_prob"] = np32(tp[..., K])                                # the (K+1)-th probability: tie detector for the checker
    pbar = probs.detach().sum(0).sum(1) / (probs.shape[0] * probs.shape[2])      # (L,T)
    touched = torch.unique(idx.reshape(-1))
    extra = torch.randperm(T, generator=gen)[:4096]
    slots = torch.unique(torch.cat([touched, extra]))
    out["slots"] = slots.numpy().astype(np.int64)
    out["pbar_at_slots"] = np32(pbar[:, slots])
    out["pbar_rowsum"] = np32(pbar.double().sum(1))
    out["pbar_max"] = np32(pbar.max(1).values)
    for k_, p_ in net.named_parameters():
        if p_.grad is None:
            continue
        key = "grad_" + k_.replace(".", "_")
        if k_.startswith("encoding."):
            continue
        if k_ == "HPD.module_list.3.0.weight":
            out[key + "_at_slots"] = np32(p_.grad[slots])
            out[key + "_abs_sum"] = np32(p_.grad.double().abs().sum())
        elif k_ == "HPD.module_list.3.0.bias":
            out[key + "_at_slots"] = np32(p_.grad[slots])
            out[key + "_abs_sum"] = np32(p_.grad.double().abs().sum())
        else:
            out[key] = np32(p_.grad)
    dt = torch.stack([net.encoding._hash_tables[l].weight.grad for l in range(L)])   # (L,T,F), almost all zero
    nz = torch.nonzero(dt.abs().sum(-1))
    out["dtables_nz"] = nz.numpy().astype(np.int32)                   # (n,2) = (level, slot)
    out["dtables_val"] = np32(dt[nz[:, 0], nz[:, 1]])
    print("table grad rows:", nz.shape[0], "slots kept:", slots.numel())
    save("G12_gngf_T19_reference", **out)


def g13(F_, U_, M, P_):
    """-hwp mode (models.py:364-371) and the reference's five checkpoint files (functions.py:761-781), cfg1 shape: a model is
    trained for two steps and saved BY THE REFERENCE (torch.save of state dicts = data); a second reference model is built
    with HPD_weights_path= (frozen HPD) and takes one step.  Written: tests/golden/ref_ckpt_cfg1/*.pt + G13 npz."""
    mods = (F_, U_, M)
    img, X, Y, h, w = load_strawberry()
    torch.manual_seed(SEED)
    perm = torch.randperm(h * w)
    B = 4096
    net = make_net(M, mods, hash_mode=False, T=256, L=4, n_min=8, n_max=32, K=4)
    loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)
    opt = F_.get_optimizer(net, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
    for step in range(2):
        sl = perm[step * B:(step + 1) * B]
        opt.zero_grad()
        rgb, probs, idx, _ = net(X[sl], 1 / 3)
        mse, kls, coll = loss_fn(rgb, Y[sl], probs.shape[-1], probs, torch.tensor([]), torch.tensor([]))
        (1 * mse + ((1 * kls) + 1).sum(0)).backward()
        opt.step()
    folder = os.path.join(OUT, "ref_ckpt_cfg1")
    os.makedirs(folder, exist_ok=True)
    torch.save(net.state_dict(), os.path.join(folder, "whole_model.pt"))            # functions.py:768
    torch.save(opt.state_dict(), os.path.join(folder, "whole_opt.pt"))              # :769
    torch.save(net.encoding.state_dict(), os.path.join(folder, "encoding_model.pt"))  # :773
    torch.save(net.HPD.state_dict(), os.path.join(folder, "HPD_model.pt"))          # :776
    torch.save(net.mlp.state_dict(), os.path.join(folder, "MLP_model.pt"))          # :780
    out = {"perm": perm[: 4 * B].numpy().astype(np.int64)}
    # resumed step of the saved model+optimizer (whole_model.pt + whole_opt.pt loaded into fresh objects)
    net_r = make_net(M, mods, hash_mode=False, T=256, L=4, n_min=8, n_max=32, K=4)
    net_r.load_state_dict(torch.load(os.path.join(folder, "whole_model.pt")))
    opt_r = F_.get_optimizer(net_r, 1e-4, 1e-3, 1e-3, 0, 1e-6, 1e-6)
    opt_r.load_state_dict(torch.load(os.path.join(folder, "whole_opt.pt")))
    sl = perm[2 * B:3 * B]
    opt_r.zero_grad()
    rgb, probs, idx, _ = net_r(X[sl], 1 / 3)
    mse, kls, coll = loss_fn(rgb, Y[sl], probs.shape[-1], probs, torch.tensor([]), torch.tensor([]))
    (1 * mse + ((1 * kls) + 1).sum(0)).backward()
    opt_r.step()
    out["resume_rgb"] = np32(rgb)
    out["resume_mse"] = np32(mse)
    for k_, p_ in net_r.named_parameters():
        out["resume_param_" + k_.replace(".", "_")] = np32(p_)
    # -hwp: frozen HPD loaded from HPD_model.pt, fresh tables/decoder (seeded), one training step
    torch.manual_seed(SEED + 1)
    net_f = M.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=256, num_levels=4, n_min=8, n_max=32,
                                       MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                       HPD_out_features=256, feature_dim=2, topk_k=4,
                                       HPD_weights_path=os.path.join(folder, "HPD_model.pt"))
    assert all(not p.requires_grad for p in net_f.HPD.parameters())
    for k_, v_ in net_f.state_dict().items():
        if not k_.startswith("HPD."):
            out["hwp_init_" + k_.replace(".", "_")] = np32(v_)
    opt_f = torch.optim.Adam([{"params": net_f.encoding.parameters(), "lr": 1e-4, "weight_decay": 0},
                              {"params": net_f.mlp.parameters(), "lr": 1e-3, "weight_decay": 1e-6}], betas=(0.9, 0.99), eps=1e-15)
    sl = perm[3 * B:4 * B]
    opt_f.zero_grad()
    rgb, probs, idx, _ = net_f(X[sl], 1 / 3)
    mse, kls, coll = loss_fn(rgb, Y[sl], probs.shape[-1], probs, torch.tensor([]), torch.tensor([]))
    (1 * mse + ((1 * kls) + 1).sum(0)).backward()
    out["hwp_rgb"] = np32(rgb)
    out["hwp_mse"] = np32(mse)
    out["hwp_kls"] = np32(kls)
    out["hwp_idx"] = np32(idx).astype(np.int32)
    for k_, p_ in net_f.named_parameters():
        if p_.grad is not None:
            out["hwp_grad_" + k_.replace(".", "_")] = np32(p_.grad)
    assert all(p.grad is None for p in net_f.HPD.parameters())
    save("G13_hwp_and_checkpoints", **out)


def g14(F_, U_, M, P_):
    """cfg3's own pixels: macaw.jpg decoded (PIL) + one hash-mode step at the headline table shape on 4096 of its pixels."""
    mods = (F_, U_, M)
    img = load_image("macaw.jpg")
    save("macaw_rgb", img=img)
    h, w = img.shape[:2]
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    X = torch.tensor(np.stack([rows, cols], -1).reshape(-1, 2)).float() / (max(w, h) - 1)
    Y = torch.tensor(img.reshape(-1, 3) / 255).float()
    L, T, Fd = 16, 2 ** 19, 2
    net = make_net(M, mods, hash_mode=True, T=T, L=L, n_min=16, n_max=512, F=Fd)
    tabs = seeded_tables(L, T, Fd, SEED + 14)
    with torch.no_grad():
        for l in range(L):
            net.encoding._hash_tables[l].weight.copy_(tabs[l])
            net.encoding._hash_tables[l].weight.mul_(100.0)      # 1e-2-scale features: the image signal reaches the output
    out = {"hw": np.array([h, w])}
    for k_, v_ in net.mlp.state_dict().items():
        out["init_mlp_" + k_.replace(".", "_")] = np32(v_)
    gen = torch.Generator().manual_seed(SEED + 14)
    sel = torch.randperm(h * w, generator=gen)[:4096]
    out["sel"] = sel.numpy().astype(np.int64)
    loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)
    net.zero_grad()
    rgb, _, idx, _ = net(X[sel], 1.0)
    mse, _, _ = loss_fn(rgb, Y[sel], None, None, None, None)
    mse.backward()
    out["rgb"] = np32(rgb)
    out["mse"] = np32(mse)
    out["idx_checksum"] = np.array([int(idx.sum()), int((idx * torch.arange(1, 4097)[:, None, None]).sum() % (2 ** 61 - 1))], dtype=np.int64)
    for k_, p_ in net.mlp.named_parameters():
        out["grad_mlp_" + k_.replace(".", "_")] = np32(p_.grad)
    dt = torch.stack([net.encoding._hash_tables[l].weight.grad for l in range(L)])
    nz = torch.nonzero(dt.abs().sum(-1))
    out["dtables_nz"] = nz.numpy().astype(np.int32)
    out["dtables_val"] = np32(dt[nz[:, 0], nz[:, 1]])
    rh.set_flag(mods, "should_use_hash_function", False)
    save("G14_macaw_hash", **out)


def g3b(F_, U_, M, P_):
    """Hash indices at the large-table shapes of BASELINE configs 4 and 5 (T = 2^22 with N 16->4096; T = 2^24 with N 16->8192)."""
    mods = (F_, U_, M)
    gen = torch.Generator().manual_seed(SEED + 3)
    x = edge_coords(512, gen)
    out = {"x": np32(x)}
    for tag, (a, b, L, T) in {"cfg4": (16, 4096, 16, 2 ** 22), "cfg5": (16, 8192, 16, 2 ** 24)}.items():
        net = make_net(M, mods, hash_mode=True, T=T, L=L, n_min=a, n_max=b)
        scaled, grid = net._scale_to_grid(x)
        out[f"{tag}_n_ls"] = np32(net._n_ls).reshape(-1)
        out[f"{tag}_grid"] = np32(grid)
        out[f"{tag}_hash"] = np32(net._fast_hash(grid.int()))
        out[f"{tag}_cfg"] = np.array([a, b, L, T])
    rh.set_flag(mods, "should_use_hash_function", False)
    save("G3b_hash_large_tables", **out)


def g15(F_, U_, M, P_):
    """should_keep_topk_only=True (models.py:478-484; half of the reference's grid, params.py:58-75): `probs` is the (P,L,4,K)
    top-K tensor and the loss's distribution term runs with N = K (functions.py:226-232).  cfg1 shape, two optimisation steps."""
    mods = (F_, U_, M)
    img, X, Y, h, w = load_strawberry()
    torch.manual_seed(SEED + 15)
    perm = torch.randperm(h * w)
    B = 4096
    net = make_net(M, mods, hash_mode=False, T=256, L=4, n_min=8, n_max=32, K=4, keep_topk=True)
    loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)
    opt = F_.get_optimizer(net, encoding_lr=1e-4, HPD_lr=1e-3, MLP_lr=1e-3,
                           encoding_weight_decay=0, HPD_weight_decay=1e-6, MLP_weight_decay=1e-6)
    out = {"perm": perm[: 2 * B].numpy().astype(np.int64), "hw": np.array([h, w])}
    for k_, v_ in net.state_dict().items():
        out["init_" + k_.replace(".", "_")] = np32(v_)
    for step in range(2):
        sl = perm[step * B:(step + 1) * B]
        bx, by = X[sl], Y[sl]
        opt.zero_grad()
        rgb, probs, idx, counts = net(bx, 1 / 3, should_calc_counts=False)
        assert tuple(probs.shape) == (B, 4, 4, 4)
        mse, kls, coll = loss_fn(rgb, by, probs.shape[-1], probs, torch.tensor([]), torch.tensor([]))
        loss = 1 * mse + ((1 * kls) + (1e-3 * coll if coll.nelement() != 0 else 1)).sum(0)
        loss.backward()
        s_ = f"s{step}_"
        out[s_ + "rgb"], out[s_ + "probs"], out[s_ + "idx"] = np32(rgb), np32(probs), np32(idx)
        out[s_ + "mse"], out[s_ + "kls"], out[s_ + "loss"] = np32(mse), np32(kls), np32(loss)
        for k_, p_ in net.named_parameters():
            if p_.grad is not None:
                out[s_ + "grad_" + k_.replace(".", "_")] = np32(p_.grad)
        opt.step()
        for k_, p_ in net.named_parameters():
            out[s_ + "param_" + k_.replace(".", "_")] = np32(p_)
    save("G15_keep_topk_only", **out)


def g16(F_, U_, M, P_):
    """should_batchnorm_data=True (models.py:394-397; main.py:50 then feeds RAW pixel coordinates): the coordinates pass through
    nn.BatchNorm1d in training mode, so grid vertices are centred on 0 (negative).  cfg1 shape, both index sources, one step."""
    mods = (F_, U_, M)
    img, X, Y, h, w = load_strawberry()
    Xraw = X * (max(w, h) - 1)                               # main.py:50-51 skipped under should_batchnorm_data
    torch.manual_seed(SEED + 16)
    perm = torch.randperm(h * w)
    B = 2048
    rh.set_flag(mods + (P_,), "should_batchnorm_data", True)
    try:
        for mode in ("gngf", "hash"):
            hash_mode = mode == "hash"
            net = make_net(M, mods, hash_mode=hash_mode, T=256, L=4, n_min=8, n_max=32, K=4)
            net.train()
            loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)
            out = {"perm": perm[:B].numpy().astype(np.int64), "hw": np.array([h, w])}
            for k_, v_ in net.state_dict().items():
                out["init_" + k_.replace(".", "_")] = np32(v_)
            bx, by = Xraw[perm[:B]], Y[perm[:B]]
            rgb, probs, idx, counts = net(bx, 1 / 3, should_calc_counts=False)
            mse, kls, coll = loss_fn(rgb, by, None if hash_mode else probs.shape[-1], probs,
                                     None if hash_mode else torch.tensor([]), None if hash_mode else torch.tensor([]))
            loss = 1 * mse
            if not hash_mode:
                loss = loss + ((1 * kls) + (1e-3 * coll if coll.nelement() != 0 else 1)).sum(0)
            loss.backward()
            out["rgb"], out["idx"], out["mse"], out["loss"] = np32(rgb), np32(idx), np32(mse), np32(loss)
            xb = (bx - bx.mean(0)) / torch.sqrt(bx.var(0, unbiased=False) + 1e-5)      # (not through the module: no second statistics update)
            out["idx_min_vertex"] = np.array(float(torch.floor(xb * 8).min()))
            if not hash_mode:
                out["kls"] = np32(kls)
                out["pbar"] = np32(probs.sum(0).sum(1) / (probs.shape[0] * probs.shape[2]))
                tp, _ti = torch.topk(probs, 4, dim=-1)
                out["topk_probs"] = np32(tp)
            for k_, p_ in net.named_parameters():
                if p_.grad is not None:
                    out["grad_" + k_.replace(".", "_")] = np32(p_.grad)
            for k_, v_ in net.state_dict().items():
                if k_.startswith("_batch_norm."):
                    out["after_" + k_.replace(".", "_")] = np32(v_)
            save(f"G16_batchnorm_{mode}", **out)
    finally:
        rh.set_flag(mods + (P_,), "should_batchnorm_data", False)
        rh.set_flag(mods, "should_use_hash_function", False)


def g17(F_, U_, M, P_):
    """should_softmax_topk_features in {None, False} at MODEL level (models.py:212-217, params.py:14): the K gathered rows are
    blended with the raw top-K probabilities (None) or with the probabilities normalised by their sum (False) instead of their
    softmax.  cfg1 shape, GNGF indexing, one step: outputs, loss terms, every gradient (the HPD receives the blend's backward)."""
    mods = (F_, U_, M)
    img, X, Y, h, w = load_strawberry()
    torch.manual_seed(SEED + 17)
    perm = torch.randperm(h * w)
    B = 2048
    try:
        for tag, flag in (("raw", None), ("norm", False)):
            rh.set_flag(mods + (P_,), "should_softmax_topk_features", flag)
            net = make_net(M, mods, hash_mode=False, T=256, L=4, n_min=8, n_max=32, K=4)
            # the default table init (+-1e-4) makes the blend invisible in rgb: scale the tables so that the variants differ
            with torch.no_grad():
                for l in range(4):
                    net.encoding._hash_tables[l].weight.mul_(3000.0)
            loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)
            out = {"perm": perm[:B].numpy().astype(np.int64), "hw": np.array([h, w])}
            for k_, v_ in net.state_dict().items():
                out["init_" + k_.replace(".", "_")] = np32(v_)
            bx, by = X[perm[:B]], Y[perm[:B]]
            rgb, probs, idx, counts = net(bx, 1 / 3, should_calc_counts=False)
            mse, kls, coll = loss_fn(rgb, by, probs.shape[-1], probs, torch.tensor([]), torch.tensor([]))
            loss = 1 * mse + ((1 * kls) + (1e-3 * coll if coll.nelement() != 0 else 1)).sum(0)
            loss.backward()
            out["rgb"], out["idx"], out["mse"], out["kls"], out["loss"] = np32(rgb), np32(idx), np32(mse), np32(kls), np32(loss)
            for k_, p_ in net.named_parameters():
                if p_.grad is not None:
                    out["grad_" + k_.replace(".", "_")] = np32(p_.grad)
            save(f"G17_blend_{tag}", **out)
    finally:
        rh.set_flag(mods + (P_,), "should_softmax_topk_features", True)
        rh.set_flag(mods, "should_use_hash_function", False)


def g12s(F_, U_, M, P_):
    """The reference's OWN numerical spread at the G12 shape: the G12 forward pass (models.py:90-123: four
    Linear layers, Softmax over T = 2^19, nan_to_num, top-K) evaluated by the reference under three configurations of the same
    torch build — 8 intra-op threads (how G12 was written), 1 thread, and 8 threads with the oneDNN (mkldnn) backend off — and the
    largest difference between any two of them in `topk_probs` (absolute and relative), `rgb` and the logit scale.  The GPU test
    of G12 asserts err <= max(1e-5, 2 x spread) on the top-K probabilities: a tolerance above north_star's 1e-5 is only claimed
    as far as the reference itself moves (~20 s and ~12 GiB per configuration).  Also stored: the same forward pass through the
    reference's modules in float64 (`fp64_topk_probs`: the exact values for these weights) and the fp32 reference's distance from
    it (`ref_fp32_vs_fp64_*`)."""
    mods = (F_, U_, M)
    img, X, Y, h, w = load_strawberry()
    L, T, Fd, K, Pn = 16, 2 ** 19, 2, 4, 8
    net = make_net(M, mods, hash_mode=False, T=T, L=L, n_min=16, n_max=512, F=Fd, K=K)
    tabs = seeded_tables(L, T, Fd, SEED + 13)
    W_last, b_last = seeded_hpd_last_layer(T)
    with torch.no_grad():
        for l in range(L):
            net.encoding._hash_tables[l].weight.copy_(tabs[l])
        net.HPD.module_list[3][0].weight.copy_(W_last)
        net.HPD.module_list[3][0].bias.copy_(b_last)
    gen = torch.Generator().manual_seed(SEED + 12)
    sel = torch.randperm(h * w, generator=gen)[:Pn]
    sel[0] = 0
    sel[1] = h * w - 1
    bx = X[sel]
    prev_threads = torch.get_num_threads()
    prev_mkldnn = torch.backends.mkldnn.enabled
    runs, names = [], []
    try:
        for name, threads, mkldnn in (("8 threads", 8, True), ("1 thread", 1, True), ("8 threads, mkldnn off", 8, False)):
            torch.set_num_threads(threads)
            torch.backends.mkldnn.enabled = mkldnn
            with torch.no_grad():
                rgb, probs, idx, _counts = net(bx, 1.0, should_calc_counts=False)
                tp = torch.topk(probs, K, dim=-1)[0]
                # the largest |logit| the softmax saw (what turns one fp32 ulp of a logit into a probability error)
                hid = net.HPD.module_list[2](net.HPD.module_list[1](net.HPD.module_list[0](torch.tensor([[513.0, 340.0]]))))
                zmax = float(net.HPD.module_list[3][0](hid).abs().max())
            runs.append((np32(tp).astype(np.float64), np32(idx), np32(rgb).astype(np.float64)))
            names.append(name)
            print(f"[G12 spread] {name}: done")
            del probs
    finally:
        torch.set_num_threads(prev_threads)
        torch.backends.mkldnn.enabled = prev_mkldnn
    # ... and the reference evaluated in float64 (the same modules, .double()): how far the reference's fp32 run itself is from
    # exact arithmetic on the same weights — what no fp32 implementation with another summation order can be asked to undercut
    net64 = net.double()
    with torch.no_grad():
        rgb64, probs64, idx64, _c = net64(bx.double(), 1.0, should_calc_counts=False)
        tp64 = torch.topk(probs64, K, dim=-1)[0]
    m64 = (np32(idx64) == runs[0][1]).all(-1)
    dev64 = np.abs(np32(tp64) - runs[0][0])[m64]
    own_abs = float(dev64.max())
    own_rel = float((dev64 / np.maximum(np32(tp64)[m64], 1e-300)).max())
    own_rgb = float(np.abs(np32(rgb64) - runs[0][2]).max())
    print(f"[G12 spread] reference fp32 vs the reference in float64: topk_probs max-abs {own_abs:.3e}, max-rel {own_rel:.3e}, rgb {own_rgb:.3e}; "
          f"ordered top-K equal on {m64.mean():.4f} of the (pixel, level, corner) rows")
    exact_tp, exact_idx = np32(tp64), np32(idx64).astype(np.int32)
    del probs64, net64
    sp_abs = sp_rel = sp_rgb = 0.0
    same_idx = True
    for i in range(len(runs)):
        for j in range(i + 1, len(runs)):
            m = (runs[i][1] == runs[j][1]).all(-1)                    # (compare where the two runs agree on the ordered top-K)
            same_idx &= bool(m.all())
            d = np.abs(runs[i][0] - runs[j][0])[m]
            sp_abs = max(sp_abs, float(d.max()))
            sp_rel = max(sp_rel, float((d / np.maximum(runs[i][0][m], 1e-300)).max()))
            sp_rgb = max(sp_rgb, float(np.abs(runs[i][2] - runs[j][2]).max()))
    print(f"[G12 spread] topk_probs max-abs {sp_abs:.3e}, max-rel {sp_rel:.3e}; rgb {sp_rgb:.3e}; ordered top-K identical: {same_idx}; |z| ~ {zmax:.1f}")
    save("G12_spread", topk_probs_spread_abs=np.float64(sp_abs), topk_probs_spread_rel=np.float64(sp_rel), rgb_spread_abs=np.float64(sp_rgb),
         ordered_topk_identical=np.bool_(same_idx), logit_scale=np.float64(zmax), configs=np.array(names),
         ref_fp32_vs_fp64_topk_probs_abs=np.float64(own_abs), ref_fp32_vs_fp64_topk_probs_rel=np.float64(own_rel),
         ref_fp32_vs_fp64_rgb_abs=np.float64(own_rgb), fp64_topk_probs=exact_tp, fp64_topk_idx=exact_idx)


GROUPS = {"G12s": g12s, "G17": g17, "G15": g15, "G16": g16, "G3b": g3b, "G11": g11, "G12": g12, "G13": g13, "G14": g14, "G10": g10, "G1": g1, "G2": g2_g3, "G4": g4, "G5": g5, "G6": g6, "G7": g7, "G9": g9}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--with-train-curve", action="store_true")
    a = ap.parse_args()
    assert rh.available(), "reference not present: goldens can only be regenerated in the build container"
    torch.set_num_threads(8)
    mods4 = rh.load_reference()
    sel = [s for s in a.only.split(",") if s] or list(GROUPS)
    for name in sel:
        if name == "G8":
            continue
        GROUPS[name](*mods4)
    if a.with_train_curve or "G8" in sel:
        g8(*mods4)
