"""Pins the CPU oracle (oracle/gngf_oracle.py) against golden vectors captured from the Python reference
(oracle/make_goldens.py).  CPU only.  Tolerances are written per test."""
import numpy as np
import pytest

from oracle import gngf_oracle as orc


def close(a, b, rtol, atol, msg=""):
    """assert_allclose + a row in the achieved-error report (tests/conftest.py: ParityRecorder)"""
    import inspect
    from conftest import parity_close
    if not msg:
        ctx = inspect.stack()[1].code_context
        msg = (ctx[0].strip() if ctx else "")[:100]
    parity_close(a, b, rtol, atol, msg)


def test_g1_level_resolutions(golden):
    g = golden("G1_level_resolutions")
    i = 0
    while f"case{i}" in g:
        a, b, L = (int(v) for v in g[f"case{i}"])
        assert np.array_equal(orc.level_resolutions(a, b, L), g[f"n_ls{i}"]), (a, b, L)
        i += 1
    assert i >= 5
    assert np.array_equal(orc.corner_offsets(2), g["hypercube"].reshape(2, 4))


@pytest.mark.parametrize("tag,cfg", [("cfg1", (8, 32, 4)), ("cfg2", (16, 512, 16)), ("cfg4", (16, 4096, 16))])
def test_g2_corners_g3_hash(golden, tag, cfg):
    g = golden("G2G3_corners_hash")
    n_ls = orc.level_resolutions(*cfg)
    scaled, grid = orc.scale_to_grid(g["x"], n_ls)
    assert np.array_equal(scaled, g[f"{tag}_scaled"])      # bit-exact fp32
    assert np.array_equal(grid, g[f"{tag}_grid"])
    for T in (2 ** 8, 2 ** 19, 1000):
        h = orc.spatial_hash(grid.astype(np.int32), T)
        assert h.dtype == np.int64
        assert np.array_equal(h, g[f"{tag}_hash_T{T}"]), T  # bit-exact, incl. the non-power-of-two table


@pytest.mark.parametrize("tag", ["small", "mid", "f4k3"])
def test_g4_encoding(golden, tag):
    g = golden("G4_encoding")
    tables = g[f"{tag}_tables"]
    out = orc.encoding_forward(tables, g[f"{tag}_hash_idx"])
    assert np.array_equal(out, g[f"{tag}_hash_out"])        # pure gather: exact
    dt, _ = orc.encoding_backward(tables, g[f"{tag}_hash_idx"], None, True, g[f"{tag}_hash_gout"])
    close(dt, g[f"{tag}_hash_dtables"], 1e-5, 1e-6)
    for vname, flag in (("softmax", True), ("raw", None), ("norm", False)):
        out = orc.encoding_forward(tables, g[f"{tag}_gngf_idx"], g[f"{tag}_gngf_probs"], flag)
        close(out, g[f"{tag}_gngf_{vname}_out"], 1e-5, 1e-9)
        dt, dp = orc.encoding_backward(tables, g[f"{tag}_gngf_idx"], g[f"{tag}_gngf_probs"], flag, g[f"{tag}_gngf_gout"])
        close(dt, g[f"{tag}_gngf_{vname}_dtables"], 1e-4, 1e-6)
        close(dp, g[f"{tag}_gngf_{vname}_dprobs"], 1e-4, 2e-7 if flag is not False else 2e-5)


@pytest.mark.parametrize("tag", ["cfg1", "cfg2", "f4"])
def test_g5_bilinear(golden, tag):
    g = golden("G5_bilinear")
    a, b, L, F = (int(v) for v in g[f"{tag}_cfg"])
    n_ls = orc.level_resolutions(a, b, L)
    out = orc.bilinear_forward(g[f"{tag}_x"], n_ls, g[f"{tag}_feats"])
    close(out, g[f"{tag}_out"], 1e-6, 1e-6)
    df = orc.bilinear_backward(g[f"{tag}_x"], n_ls, g[f"{tag}_gout"], F)
    close(df, g[f"{tag}_dfeats"], 1e-6, 1e-7)


def _mlp_params(g, prefix, n):
    return ([g[f"{prefix}{i}_0_weight"] for i in range(n)], [g[f"{prefix}{i}_0_bias"] for i in range(n)])


@pytest.mark.parametrize("tag", ["cfg1", "cfg2", "bw_leaky"])
def test_g9_decoder(golden, tag):
    g = golden("G9_decoder")
    leaky = bool(g[f"{tag}_cfg"][3])
    W, B = _mlp_params(g, f"{tag}_w_", 3)
    y = orc.decoder_forward(g[f"{tag}_x"], W, B, leaky)
    close(y, g[f"{tag}_y"], 1e-5, 1e-6)
    dx, dW, dB = orc.decoder_backward(g[f"{tag}_x"], W, B, g[f"{tag}_gy"], leaky)
    close(dx, g[f"{tag}_dx"], 1e-4, 1e-6)
    for i in range(3):
        close(dW[i], g[f"{tag}_g_{i}_0_weight"], 1e-4, 1e-5)
        close(dB[i], g[f"{tag}_g_{i}_0_bias"], 1e-4, 1e-5)


def topk_sets_match(ti, tp, gi, gp, probs, tol):
    """top-K membership equal up to ties: every index the oracle picked that the golden did not must have a
    probability within `tol` (relative) of the golden's K-th value."""
    kth = gp[..., -1:]
    bad = 0
    for r in range(ti.shape[0]):
        extra = set(ti[r].tolist()) - set(gi[r].tolist())
        for e in extra:
            if abs(probs[r, e] - kth[r, 0]) > tol * abs(kth[r, 0]):
                bad += 1
    return bad


@pytest.mark.parametrize("T", [256, 2048])
def test_g6_hpd(golden, T):
    g = golden("G6_hpd")
    W, B = _mlp_params(g, f"T{T}_hpd_module_list_", 4)
    verts = g[f"T{T}_verts"]
    probs, _, _ = orc.hpd_forward(verts, W, B, 1)
    close(probs, g[f"T{T}_probs"], 2e-5, 1e-9)
    close(probs.sum(-1), np.ones(len(verts)), 1e-5, 0)
    for K in (1, 4, 20):
        _, tp, ti = orc.hpd_forward(verts, W, B, K)
        tag = f"T{T}_K{K}"
        gi, gp = g[f"{tag}_topk_idx"].reshape(len(verts), -1), g[f"{tag}_topk_probs"].reshape(len(verts), -1)
        close(tp.reshape(gp.shape), gp, 2e-5, 1e-9)                 # sorted values agree even where ties reorder indices
        assert topk_sets_match(ti.reshape(gi.shape), tp, gi, gp, g[f"T{T}_probs"], 1e-5) == 0
        dq = g[f"{tag}_dq_in"].reshape(tp.shape)
        # torch.topk's tie order is unspecified (many exactly-tied tail probabilities here): the backward is
        # compared with d_topk routed to the slots the reference picked.
        dW, dB = orc.hpd_backward(verts, W, B, K, dq, g[f"T{T}_dprobs_in"], topk_idx=gi)
        for i in range(4):
            gw = g[f"{tag}_grad_module_list_{i}_0_weight"]
            scale = np.abs(gw).max()
            close(dW[i], gw, 2e-3, 2e-4 * scale)
            close(dB[i], g[f"{tag}_grad_module_list_{i}_0_bias"], 2e-3, 2e-4 * scale)


def _state(g, prefix):
    dec = _mlp_params(g, prefix + "mlp_", 3)
    tables = np.stack([g[f"{prefix}encoding__hash_tables_{l}_weight"] for l in range(4)])
    return tables, dec


@pytest.mark.parametrize("mode", ["hash", "gngf"])
def test_g7_end_to_end_forward(golden, mode):
    g = golden(f"G7_end_to_end_{mode}")
    img = golden("strawberry_rgb")["img"]
    h, w = (int(v) for v in g["hw"])
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    X = (np.stack([rows, cols], -1).reshape(-1, 2).astype(np.float32) / np.float32(max(w, h) - 1)).astype(np.float32)
    Y = (img.reshape(-1, 3) / 255).astype(np.float32)
    sl = g["perm"][:4096]
    tables, (dw, db) = _state(g, "init_")
    n_ls = orc.level_resolutions(8, 32, 4)
    kw = {}
    if mode == "gngf":
        kw["hpd_w"], kw["hpd_b"] = _mlp_params(g, "init_HPD_module_list_", 4)
    out = orc.gngf_forward(X[sl], n_ls, tables, dw, db, hash_mode=(mode == "hash"), K=4, **kw)
    close(out["rgb"], g["s0_rgb"], 1e-5, 1e-6)
    if mode == "hash":
        assert np.array_equal(out["idx"], g["s0_idx"])
    else:
        close(out["topk_probs"], g["s0_topk_probs"], 1e-5, 1e-9)
        assert (out["idx"] == g["s0_idx"]).mean() > 0.999
    mse, kls = orc.loss_forward(out["rgb"], Y[sl], out["probs"], gamma=-2, epsilon=1)
    close(mse, g["s0_mse"], 1e-5, 0)
    if mode == "gngf":
        close(kls, g["s0_kls"], 1e-4, 1e-9)
        close(mse + (kls + 1).sum(), g["s0_loss"], 1e-5, 0)       # functions.py:245 "+1 per level" quirk


def test_adam_matches_reference_step(golden):
    g = golden("G7_end_to_end_hash")
    for name, lr, wd in (("encoding__hash_tables_0_weight", 1e-4, 0.0), ("mlp_0_0_weight", 1e-3, 1e-6)):
        p0 = g["init_" + name]
        p1, m, v = orc.adam_step(p0, g["s0_grad_" + name], np.zeros_like(p0), np.zeros_like(p0), 1, lr, wd)
        close(p1, g["s0_param_" + name], 1e-5, 1e-9)
        p2, m, v = orc.adam_step(p1, g["s1_grad_" + name], m, v, 2, lr, wd)
        close(p2, g["s1_param_" + name], 1e-5, 1e-9)


def _strawberry_xy(golden, g, raw=False):
    img = golden("strawberry_rgb")["img"]
    h, w = (int(v) for v in g["hw"])
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    X = np.stack([rows, cols], -1).reshape(-1, 2).astype(np.float32)
    if not raw:
        X = (X / np.float32(max(w, h) - 1)).astype(np.float32)
    Y = (img.reshape(-1, 3) / 255).astype(np.float32)
    return X, Y


def test_g15_keep_topk_only(golden):
    """should_keep_topk_only=True (reference models.py:478-484, functions.py:226-232): probs IS the (P,L,4,K) top-K tensor
    and the distribution term of the loss runs with N = K."""
    g = golden("G15_keep_topk_only")
    X, Y = _strawberry_xy(golden, g)
    sl = g["perm"][:4096]
    tables, (dw, db) = _state(g, "init_")
    hw, hb = _mlp_params(g, "init_HPD_module_list_", 4)
    out = orc.gngf_forward(X[sl], orc.level_resolutions(8, 32, 4), tables, dw, db, hash_mode=False, K=4, hpd_w=hw, hpd_b=hb)
    close(out["rgb"], g["s0_rgb"], 1e-5, 1e-6)
    assert g["s0_probs"].shape == (4096, 4, 4, 4)
    same = (out["idx"] == g["s0_idx"]).all(-1)
    assert same.mean() > 0.999
    close(out["topk_probs"][same], g["s0_probs"][same], 1e-5, 1e-9)
    mse, kls = orc.loss_forward(out["rgb"], Y[sl], out["topk_probs"], gamma=-2, epsilon=1)      # N = K
    close(mse, g["s0_mse"], 1e-5, 0)
    close(kls, g["s0_kls"], 1e-4, 1e-9)
    close(mse + (kls + 1).sum(), g["s0_loss"], 1e-5, 0)


@pytest.mark.parametrize("mode", ["hash", "gngf"])
def test_g16_batchnorm_coordinates(golden, mode):
    """should_batchnorm_data=True (reference models.py:394-397 with main.py:50 feeding raw pixel coordinates): BatchNorm1d in
    training mode centres the coordinates on 0, so the grid has NEGATIVE vertices (hashed / fed to the HPD as they are)."""
    g = golden(f"G16_batchnorm_{mode}")
    X, Y = _strawberry_xy(golden, g, raw=True)
    sl = g["perm"]
    x = orc.batch_norm_train(X[sl], g["init__batch_norm_weight"], g["init__batch_norm_bias"])
    assert float(np.floor(x * 8).min()) == float(g["idx_min_vertex"]) < 0
    tables, (dw, db) = _state(g, "init_")
    kw = {}
    if mode == "gngf":
        kw["hpd_w"], kw["hpd_b"] = _mlp_params(g, "init_HPD_module_list_", 4)
    out = orc.gngf_forward(x, orc.level_resolutions(8, 32, 4), tables, dw, db, hash_mode=(mode == "hash"), K=4, **kw)
    close(out["rgb"], g["rgb"], 1e-5, 1e-6)
    if mode == "hash":
        assert np.array_equal(out["idx"], g["idx"])
    else:
        assert (out["idx"] == g["idx"]).mean() > 0.999
        close(out["topk_probs"], g["topk_probs"], 1e-4, 1e-9)
    mse, kls = orc.loss_forward(out["rgb"], Y[sl], out["probs"], gamma=-2, epsilon=1)
    close(mse, g["mse"], 1e-5, 0)
    if mode == "gngf":
        close(kls, g["kls"], 1e-4, 1e-9)


@pytest.mark.parametrize("tag,blend", [("raw", None), ("norm", False)])
def test_g17_blend_variants_at_model_level(golden, tag, blend):
    """should_softmax_topk_features in {None, False} (reference models.py:212-217, params.py:14) through the whole model: the
    oracle's blend restatement against the reference's own rgb / loss terms."""
    g = golden(f"G17_blend_{tag}")
    X, Y = _strawberry_xy(golden, g)
    sl = g["perm"]
    tables, (dw, db) = _state(g, "init_")
    hw, hb = _mlp_params(g, "init_HPD_module_list_", 4)
    out = orc.gngf_forward(X[sl], orc.level_resolutions(8, 32, 4), tables, dw, db, hash_mode=False, K=4, hpd_w=hw, hpd_b=hb, blend=blend)
    close(out["rgb"], g["rgb"], 1e-5, 1e-6)
    assert (out["idx"] == g["idx"]).mean() > 0.999
    mse, kls = orc.loss_forward(out["rgb"], Y[sl], out["probs"], gamma=-2, epsilon=1)
    close(mse, g["mse"], 1e-5, 0)
    close(kls, g["kls"], 1e-4, 1e-9)
    # the variants are not vacuous: the softmax blend gives a different image from the same weights
    soft = orc.gngf_forward(X[sl], orc.level_resolutions(8, 32, 4), tables, dw, db, hash_mode=False, K=4, hpd_w=hw, hpd_b=hb, blend=True)
    assert float(np.abs(soft["rgb"] - g["rgb"]).max()) > 1e-4         # (norm: 2.6e-4, raw: 6e-3; the comparison above holds 1e-5)
