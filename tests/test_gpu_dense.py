"""GPU parity: MFMA linear kernels, softmax/top-K, HPD module vs numpy / oracle / reference goldens."""
import numpy as np
import pytest
import torch

from conftest import parity_close
from oracle import gngf_oracle as orc

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


@pytest.fixture(scope="module")
def ops():
    from collision_handling_in_instantngp_amd import ops as o
    return o


@pytest.mark.parametrize("M,N,K,act", [(1000, 64, 32, 1), (257, 3, 64, 3), (65, 128, 2, 1), (4096, 300, 128, 0), (1, 1, 1, 2),
                                        (130, 70, 33, 2), (300, 257, 130, 1), (129, 128, 2, 3)])
def test_linear_fwd_bwd_vs_numpy(ops, M, N, K, act):
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) * 0.1
    z = x.astype(np.float64) @ w.T.astype(np.float64) + b
    y = {0: z, 1: np.maximum(z, 0), 2: np.where(z > 0, z, 0.01 * z), 3: 1 / (1 + np.exp(-z))}[act]
    yg = ops.linear_fwd(t(x), t(w), t(b), act)
    close(yg, y, 1e-5, 1e-5)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    dact = {0: np.ones_like(z), 1: (z > 0) * 1.0, 2: np.where(z > 0, 1.0, 0.01), 3: y * (1 - y)}[act]
    dz = dy * dact
    close(ops.linear_bwd_input(t(dy), yg, t(w), act), dz @ w.astype(np.float64), 1e-4, 1e-4)
    dw, db = torch.zeros((N, K), device=DEV), torch.zeros((N,), device=DEV)
    ops.linear_bwd_weight(t(dy), yg, t(x), dw, db, act)
    scale = np.abs(dz.T @ x).max() + 1e-6
    close(dw, dz.T @ x.astype(np.float64), 1e-4, 1e-5 * scale)
    close(db, dz.sum(0), 1e-4, 1e-5 * scale)


@pytest.mark.parametrize("shape", [(16, 700, 5000), (256, 390, 3001), (130, 128, 70)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_acc_all_layouts(ops, ta, tb, shape):
    rng = np.random.default_rng(3)
    M, N, Kc = shape
    a = rng.standard_normal((Kc, M) if ta else (M, Kc)).astype(np.float32)
    b = rng.standard_normal((N, Kc) if tb else (Kc, N)).astype(np.float32)
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    c = torch.zeros((M, N), device=DEV)
    ops.gemm_acc(t(a), t(b), c, M, N, Kc, ta, tb)
    close(c, want, 1e-4, 1e-3)


@pytest.mark.parametrize("U,T,K", [(50, 256, 4), (7, 2048, 20), (3, 100000, 32), (130, 33, 1), (5, 64, 32)])
def test_softmax_topk_vs_oracle(ops, U, T, K):
    rng = np.random.default_rng(U * 7 + K)
    z = (rng.standard_normal((U, T)) * 3).astype(np.float32)
    z[0, : T // 2] = z[0, 0]                       # a large tie group: ties must resolve to the lower index
    if U > 2:
        z[2, 5] = np.nan                             # NaN poisons the row; nan_to_num -> zeros (models.py:111)
    zm = z - z.max(-1, keepdims=True)
    e = np.exp(zm.astype(np.float64))
    sm = np.nan_to_num((e / e.sum(-1, keepdims=True)).astype(np.float32))
    probs, tv, ti = ops.SoftmaxTopkFunction.apply(t(z), K)
    close(probs, sm, 2e-5, 1e-12)
    p_gpu = probs.cpu().numpy()
    wv, wi = orc.topk_desc(p_gpu, K)               # selection is checked on the GPU's own fp32 probabilities: exact
    assert np.array_equal(tv.cpu().numpy(), wv)
    assert np.array_equal(ti.cpu().numpy().astype(np.int64), wi)


def test_softmax_bwd_dense_and_topk_gradients(ops):
    rng = np.random.default_rng(11)
    U, T, K = 40, 500, 6
    z = (rng.standard_normal((U, T)) * 2).astype(np.float32)
    zt = t(z).requires_grad_()
    probs, tv, ti = ops.SoftmaxTopkFunction.apply(zt, K)
    gp = rng.standard_normal((U, T)).astype(np.float32)
    gq = rng.standard_normal((U, K)).astype(np.float32)
    (probs * t(gp)).sum().add((tv * t(gq)).sum()).backward()
    p = probs.detach().cpu().numpy().astype(np.float64)
    g = gp.astype(np.float64).copy()
    np.put_along_axis(g, ti.cpu().numpy().astype(np.int64), np.take_along_axis(g, ti.cpu().numpy().astype(np.int64), -1) + gq, -1)
    want = p * (g - (g * p).sum(-1, keepdims=True))
    close(zt.grad, want, 1e-4, 1e-7)


def _mlp_params(g, prefix, n):
    return ([g[f"{prefix}{i}_0_weight"] for i in range(n)], [g[f"{prefix}{i}_0_bias"] for i in range(n)])


@pytest.mark.parametrize("T", [256, 2048])
@pytest.mark.parametrize("K", [1, 4, 20])
def test_hpd_module_vs_reference_golden(golden, T, K):
    """HashProbDistribution.forward/backward (dense per-row formulation) vs the reference's own outputs."""
    from collision_handling_in_instantngp_amd import models
    g = golden("G6_hpd")
    hpd = models.HashProbDistribution([32, 64, 128], in_features=2, out_features=T, k=K).to(DEV)
    sd = {}
    for i in range(4):
        sd[f"module_list.{i}.0.weight"] = t(g[f"T{T}_hpd_module_list_{i}_0_weight"])
        sd[f"module_list.{i}.0.bias"] = t(g[f"T{T}_hpd_module_list_{i}_0_bias"])
    hpd.load_state_dict(sd)
    verts = t(g[f"T{T}_verts"])
    probs, tp, ti = hpd(verts)
    assert ti.dtype == torch.int64
    tag = f"T{T}_K{K}"
    close(probs, g[f"T{T}_probs"], 5e-5, 1e-9)
    gp = g[f"{tag}_topk_probs"].reshape(tp.shape)
    gi = g[f"{tag}_topk_idx"].reshape(ti.shape)
    close(tp, gp, 5e-5, 1e-9)
    # membership equal up to ties at the K-th value
    p_ref = g[f"T{T}_probs"]
    ti_np = ti.cpu().numpy()
    for r in range(ti_np.shape[0]):
        for e in set(ti_np[r].tolist()) - set(gi[r].tolist()):
            assert abs(p_ref[r, e] - gp[r, -1]) <= 1e-4 * abs(gp[r, -1]) + 1e-12
    same_rows = np.all(ti_np == gi, axis=1)
    # backward on the rows where the (unspecified) tie order agrees with the reference's
    dq = t(g[f"{tag}_dq_in"].reshape(tp.shape)) * t(same_rows.astype(np.float32))[:, None]
    (tp * dq).sum().add((probs * t(g[f"T{T}_dprobs_in"])).sum()).backward()
    if same_rows.all():
        for i in range(4):
            gw = g[f"{tag}_grad_module_list_{i}_0_weight"]
            scale = np.abs(gw).max()
            close(hpd.module_list[i][0].weight.grad, gw, 2e-3, 2e-4 * scale)
            close(hpd.module_list[i][0].bias.grad, g[f"{tag}_grad_module_list_{i}_0_bias"], 2e-3, 2e-4 * scale)
    else:
        W, B = _mlp_params(g, f"T{T}_hpd_module_list_", 4)
        dq_np = dq.cpu().numpy()
        dW, dB = orc.hpd_backward(g[f"T{T}_verts"], W, B, K, dq_np, g[f"T{T}_dprobs_in"], topk_idx=ti_np)
        for i in range(4):
            scale = np.abs(dW[i]).max()
            close(hpd.module_list[i][0].weight.grad, dW[i], 2e-3, 2e-4 * scale)
            close(hpd.module_list[i][0].bias.grad, dB[i], 2e-3, 2e-4 * scale)


def test_differentiable_topk_standalone():
    from collision_handling_in_instantngp_amd import models
    rng = np.random.default_rng(5)
    x = rng.standard_normal((6, 5, 300)).astype(np.float32)
    xt = t(x).requires_grad_()
    v, i = models.DifferentiableTopk.apply(xt, 7, -1)
    wv, wi = orc.topk_desc(x, 7)
    assert np.array_equal(v.detach().cpu().numpy(), wv) and np.array_equal(i.cpu().numpy(), wi)
    g = rng.standard_normal(wv.shape).astype(np.float32)
    v.backward(t(g))
    want = np.zeros_like(x)
    np.put_along_axis(want, wi, g, -1)
    assert np.array_equal(xt.grad.cpu().numpy(), want)


@pytest.mark.parametrize("tag", ["cfg1", "cfg2", "bw_leaky"])
@pytest.mark.parametrize("fused", [True, False])
def test_decoder_vs_reference_golden(ops, golden, tag, fused):
    """decoder MLP (fused MFMA kernel and generic chain) vs the reference's own forward / backward (G9)."""
    g = golden("G9_decoder")
    L, Fd, bw, leaky = (int(v) for v in g[f"{tag}_cfg"])
    params = []
    for i in range(3):
        params += [t(g[f"{tag}_w_{i}_0_weight"]).requires_grad_(), t(g[f"{tag}_w_{i}_0_bias"]).requires_grad_()]
    hid = ops.ACT_LEAKY if leaky else ops.ACT_RELU
    x = t(g[f"{tag}_x"]).requires_grad_()
    y = ops.decoder_apply(x, (hid, hid, ops.ACT_SIGMOID), params, fused=fused)
    close(y, g[f"{tag}_y"], 1e-5, 1e-6)
    y.backward(t(g[f"{tag}_gy"]))
    close(x.grad, g[f"{tag}_dx"], 1e-4, 1e-6)
    for i in range(3):
        close(params[2 * i].grad, g[f"{tag}_g_{i}_0_weight"], 1e-4, 1e-5)
        close(params[2 * i + 1].grad, g[f"{tag}_g_{i}_0_bias"], 1e-4, 1e-5)


@pytest.mark.parametrize("P,in_dim,out_dim,leaky", [(1, 32, 3, 0), (127, 32, 3, 0), (129, 8, 3, 0), (40000, 32, 3, 0),
                                                     (5000, 64, 1, 1), (333, 24, 4, 0), (70000, 16, 3, 1)])
def test_fused_decoder_vs_numpy(ops, P, in_dim, out_dim, leaky):
    rng = np.random.default_rng(P + in_dim)
    dims = [in_dim, 64, 64, out_dim]
    W = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    B = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(3)]
    x = rng.standard_normal((P, in_dim)).astype(np.float32)
    gy = rng.standard_normal((P, out_dim)).astype(np.float32)
    y = orc.decoder_forward(x, W, B, bool(leaky))
    dx, dW, dB = orc.decoder_backward(x, W, B, gy, bool(leaky))
    params = []
    for i in range(3):
        params += [t(W[i]).requires_grad_(), t(B[i]).requires_grad_()]
    xt = t(x).requires_grad_()
    hid = ops.ACT_LEAKY if leaky else ops.ACT_RELU
    yg = ops.decoder_apply(xt, (hid, hid, ops.ACT_SIGMOID), params, fused=True)
    close(yg, y, 1e-5, 1e-6)
    yg.backward(t(gy))
    close(xt.grad, dx, 1e-4, 2e-6)
    for i in range(3):
        scale = max(1.0, float(np.abs(dW[i]).max()))
        close(params[2 * i].grad, dW[i], 2e-4, 2e-5 * scale)
        close(params[2 * i + 1].grad, dB[i], 2e-4, 2e-5 * scale)


@pytest.mark.parametrize("U,T,K,Lv", [(100, 700, 4, 5), (64, 8192 + 300, 3, 16), (130, 257, 0, 4), (70, 1000, 6, 0),
                                      # T % 32 == 0: the MFMA form of the two passes (ragged last row block, 1..32 levels)
                                      (200, 4096 + 64, 4, 16), (130, 8192, 3, 5), (33, 96, 2, 3), (70, 1024, 6, 0),
                                      (129, 640, 0, 29)])
@pytest.mark.parametrize("saved_p", [False, True])
def test_softmax_bwd_lowrank_from_logits_vs_numpy(ops, U, T, K, Lv, saved_p):
    """streamed softmax / top-K / batch-mean backward from recomputed logits (chunked per-vertex path).
    saved_p: the probabilities at the top-K slots come from the forward's topk_val (what ops.HpdVertexFunction passes since
    round 4) instead of a second, random read of the logits."""
    from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr
    rng = np.random.default_rng(U + T)
    z = (rng.standard_normal((U, T)) * 2).astype(np.float32)
    z[1, 7] = np.nan                                         # NaN row: probabilities are nan_to_num'ed to 0 -> zero gradient
    probs, tv, ti = ops.SoftmaxTopkFunction.apply(t(z), max(K, 1))
    zt = t(z)
    tvb, tib = torch.empty((U, max(K, 1)), device=DEV), torch.empty((U, max(K, 1)), dtype=torch.int32, device=DEV)
    rowstat = torch.empty((U, 2), device=DEV)
    call("gngf_softmax_topk", ptr(zt), ptr(tvb), ptr(tib), ptr(rowstat), U, T, max(K, 1), stream_ptr())
    p = zt.cpu().numpy().astype(np.float64)                  # probabilities (in place)
    mw = rng.random((U, max(Lv, 1))).astype(np.float32)
    G = rng.standard_normal((max(Lv, 1), T)).astype(np.float32)
    dq = rng.standard_normal((U, max(K, 1))).astype(np.float32)
    g = (mw.astype(np.float64) @ G.astype(np.float64)) if Lv else np.zeros((U, T))
    tin = tib.cpu().numpy().astype(np.int64)
    if K:
        np.put_along_axis(g, tin[:, :K], np.take_along_axis(g, tin[:, :K], -1) + dq[:, :K], -1)
    want = p * (g - (g * p).sum(-1, keepdims=True))
    logits = t(z)                                            # fresh logits, as the backward recomputes them
    db = torch.zeros((T,), device=DEV)
    scratch = torch.empty((U * (1 + max(K, 1)),), device=DEV)
    dq_t = t(dq[:, :K].copy()) if K else None                # keep every device tensor alive across the launch
    ti_t = tib[:, :K].contiguous() if K else None
    mw_t, G_t = (t(mw), t(G)) if Lv else (None, None)
    tp_t = tvb[:, :K].contiguous() if (K and saved_p) else None
    call("gngf_softmax_bwd_lowrank", ptr(logits), ptr(rowstat), ptr(dq_t), ptr(ti_t), ptr(mw_t), ptr(G_t), Lv, ptr(db),
         ptr(scratch), ptr(tp_t), U, T, K, stream_ptr())
    scale = np.abs(want).max()
    close(logits, want, 2e-4, 2e-6 * scale)
    close(db, want.sum(0), 2e-4, 2e-5 * scale)


@pytest.mark.parametrize("U,T,K,Lv", [(70, 500, 4, 5), (33, 9000, 8, 16), (40, 257, 1, 0), (5, 100000, 32, 3),
                                      # T % 32 == 0 and more than 4 levels: the MFMA form of the p-bar accumulation
                                      (300, 4096 + 96, 4, 16), (129, 8192, 2, 5), (7, 160, 3, 32), (257, 1024, 4, 11)])
def test_streaming_logits_topk_pbar_vs_numpy(ops, U, T, K, Lv):
    """one-read online statistics + top-K on the logits, and the p-bar accumulation from logits"""
    from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr
    rng = np.random.default_rng(U * 3 + K)
    z = (rng.standard_normal((U, T)) * 3).astype(np.float32)
    z[0, : T // 3] = z[0, 0]                                  # ties: lower index wins
    if U > 2:
        z[2, 9] = np.nan                                       # NaN row -> probabilities all zero, slots 0..K-1
    zt = t(z)
    tv = torch.empty((U, K), device=DEV)
    ti = torch.empty((U, K), dtype=torch.int32, device=DEV)
    rowstat = torch.empty((U, 2), device=DEV)
    mw_t = t(rng.random((U, max(Lv, 1))).astype(np.float32))
    pbar = torch.zeros((max(Lv, 1), T), device=DEV)
    call("gngf_logits_topk_pbar", ptr(zt), ptr(tv), ptr(ti), ptr(rowstat), ptr(mw_t if Lv else None), Lv, ptr(pbar if Lv else None),
         U, T, K, stream_ptr())
    assert torch.equal(zt, t(z)) or np.isnan(z).any()           # logits are not modified
    zz = z.astype(np.float64)
    zm = zz - np.nanmax(zz, -1, keepdims=True)
    e = np.exp(zm)
    sm = e / e.sum(-1, keepdims=True)
    nan_rows = np.isnan(z).any(-1)
    sm[nan_rows] = 0.0
    order = np.argsort(-np.where(nan_rows[:, None], -np.arange(T)[None, :].astype(np.float64), zz), axis=-1, kind="stable")[:, :K]
    order[nan_rows] = np.arange(K)[None, :]
    assert np.array_equal(ti.cpu().numpy().astype(np.int64), order)
    close(tv, np.take_along_axis(sm, order, -1), 3e-5, 1e-12)
    if Lv:
        want = mw_t.cpu().numpy().astype(np.float64).T @ sm
        close(pbar, want, 1e-4, 1e-9)


@pytest.mark.parametrize("P,in_dim,leaky", [(1000, 32, False), (4500, 16, True), (777, 64, False), (300, 24, True), (128 * 257 + 5, 32, False)])
def test_decoder_saved_hidden_equals_recompute(ops, P, in_dim, leaky):
    """The two backward variants of the fused decoder (hidden layers read back from the forward kernel's buffer /
    recomputed) are the same arithmetic: bit-identical d enc and parameter gradients."""
    rng = np.random.default_rng(P)
    x = rng.standard_normal((P, in_dim)).astype(np.float32)
    dims = [in_dim, 64, 64, 3]
    ws = []
    for i in range(3):
        ws += [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32), (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)]
    dy = rng.standard_normal((P, 3)).astype(np.float32)
    acts = (ops.ACT_LEAKY if leaky else ops.ACT_RELU,) * 2 + (ops.ACT_SIGMOID,)
    res = []
    old = ops.DECODER_SAVE_HIDDEN
    from collision_handling_in_instantngp_amd import _lib
    prev_hybrid = _lib.query("gngf_set_decoder_bwd_hybrid", 0)      # all-fp32 kernels on both sides: the same arithmetic
    try:
        for save in (True, False):
            ops.DECODER_SAVE_HIDDEN = save
            xt = t(x).requires_grad_()
            params = [t(w).requires_grad_() for w in ws]
            y = ops.decoder_apply(xt, acts, params, fused=True)
            y.backward(t(dy))
            res.append([y.detach().cpu().numpy(), xt.grad.cpu().numpy()] + [p.grad.cpu().numpy() for p in params])
    finally:
        ops.DECODER_SAVE_HIDDEN = old
        _lib.query("gngf_set_decoder_bwd_hybrid", prev_hybrid)
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("variant", ["hybrid", "split"])
@pytest.mark.parametrize("P,leaky", [(1000, False), (128 * 300 + 77, True)])
def test_bf16_split_decoder_kernels_are_as_accurate_as_the_fp32_ones(ops, P, leaky, variant):
    """The decoder products moved to v_mfma_f32_32x32x16_bf16 with every operand split exactly into three bf16 terms:
    "hybrid" (the default backward at 32 input features: dh1 and d enc on the bf16 pipe, weight gradients on the fp32 pipe,
    csrc/decoder.hip) and "split" (gngf_set_decoder_split_bf16(1): everything, csrc/decoder_split.inc).  Against a float64
    evaluation the error must not exceed twice that of the all-fp32 kernels (+ 1e-7 absolute on rgb); a hidden unit whose
    pre-activation is within rounding of 0 may switch sides, which changes that pixel's gradient — hence the 99.9 % quantile
    for d enc and the looser bound on the summed gradients."""
    from collision_handling_in_instantngp_amd import _lib
    if variant == "split":
        prev = _lib.query("gngf_set_decoder_split_bf16", 0)
        if prev < 0:
            pytest.skip("library built without the all-bf16 decoder kernels (make SPLIT=1 builds them)")
    rng = np.random.default_rng(P)
    x = (0.5 * rng.standard_normal((P, 32))).astype(np.float32)
    dims = [32, 64, 64, 3]
    ws = []
    for i in range(3):
        ws += [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32), (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)]
    dy = (1e-3 * rng.standard_normal((P, 3))).astype(np.float32)
    acts = (ops.ACT_LEAKY if leaky else ops.ACT_RELU,) * 2 + (ops.ACT_SIGMOID,)
    x64 = torch.tensor(x, dtype=torch.float64, device=DEV, requires_grad=True)
    p64 = [torch.tensor(w, dtype=torch.float64, device=DEV, requires_grad=True) for w in ws]
    actf = (lambda v: torch.nn.functional.leaky_relu(v, 0.01)) if leaky else torch.relu
    y64 = torch.sigmoid(actf(actf(x64 @ p64[0].T + p64[1]) @ p64[2].T + p64[3]) @ p64[4].T + p64[5])
    y64.backward(torch.tensor(dy, dtype=torch.float64, device=DEV))
    want = [y64.detach(), x64.grad] + [p.grad for p in p64]
    errs = []
    for which in ("fp32", variant):
        prev = _lib.query("gngf_set_decoder_split_bf16", 1 if which == "split" else 0) if variant == "split" else 0
        prev_h = _lib.query("gngf_set_decoder_bwd_hybrid", 1 if which == "hybrid" else 0)
        try:
            xt = t(x).requires_grad_()
            params = [t(w).requires_grad_() for w in ws]
            y = ops.decoder_apply(xt, acts, params, fused=True)
            y.backward(t(dy))
            torch.cuda.synchronize()
        finally:
            if variant == "split":
                _lib.query("gngf_set_decoder_split_bf16", prev)
            _lib.query("gngf_set_decoder_bwd_hybrid", prev_h)
        got = [y.detach(), xt.grad] + [p.grad for p in params]
        e = [float((got[0].double() - want[0]).abs().max())]
        rel = (got[1].double() - want[1]).abs().max(1).values / want[1].abs().max()
        e.append(float(torch.quantile(rel, 0.999)))
        e += [float((g.double() - w).abs().max() / w.abs().max()) for g, w in zip(got[2:], want[2:])]
        errs.append(e)
    fp32, split = errs
    assert split[0] <= 2 * fp32[0] + 1e-7, (fp32, split)
    assert split[1] <= 2 * fp32[1] + 1e-7, (fp32, split)
    for a, b in zip(fp32[2:], split[2:]):
        assert b <= 2 * a + 2e-4, (fp32, split)


@pytest.mark.parametrize("P,leaky,gl,out_dim", [(1000, False, 1.0, 3), (128 * 300 + 77, True, 0.5, 3), (128 * 257 + 5, False, 1.0, 3),
                                               (3000, False, 1.0, 1)])
def test_fused_decoder_training_kernel_equals_forward_plus_backward(ops, P, leaky, gl, out_dim):
    """gngf_decoder_train (forward + MSE gradient + backward of the decoder in one launch, hidden layers in registers) against
    the two-kernel path on the same inputs: rgb, the loss value, d enc and the six parameter gradients.  The forward layers of
    the fused kernel run on the bf16 pipe (exact three-way split) instead of the fp32 pipe, so equality is to fp32 rounding;
    a hidden unit within rounding of 0 may switch sides in a few pixels (99.9 % quantile on d enc)."""
    from collision_handling_in_instantngp_amd import _lib
    rng = np.random.default_rng(P)
    x = (0.5 * rng.standard_normal((P, 32))).astype(np.float32)
    tgt = t(rng.random((P, out_dim)).astype(np.float32))          # out_dim 1: the reference's should_bw
    dims = [32, 64, 64, out_dim]
    ws = []
    for i in range(3):
        ws += [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32), (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)]
    acts = (ops.ACT_LEAKY if leaky else ops.ACT_RELU,) * 2 + (ops.ACT_SIGMOID,)
    res = {}
    old = (ops.DECODER_TRAIN_FUSION,)
    try:
        for fusion in (False, True):
            ops.DECODER_TRAIN_FUSION = fusion
            xt = t(x).requires_grad_()
            params = [t(w).requires_grad_() for w in ws]
            _lib.PROFILE = {}
            y = ops.decoder_apply(xt, acts, params, fused=True, mse_target=tgt, mse_gloss=gl)
            loss = ops.mse_loss(y, tgt)
            (loss if gl == 1.0 else gl * loss).backward()
            torch.cuda.synchronize()
            names = set(_lib.PROFILE)
            _lib.PROFILE = None
            assert ("gngf_decoder_train" in names) == fusion and ("gngf_decoder_bwd" in names) == (not fusion), names
            res[fusion] = [y.detach(), loss.detach(), xt.grad] + [p.grad for p in params]
    finally:
        _lib.PROFILE = None
        ops.DECODER_TRAIN_FUSION = old[0]
    a, b = res[False], res[True]
    close(b[0], a[0].cpu().numpy(), 0, 5e-7, "fused training kernel: rgb vs decoder_fwd")
    close(b[1], a[1].cpu().numpy(), 1e-6, 0, "fused training kernel: loss value")
    rel = (a[2] - b[2]).abs().max(1).values / a[2].abs().max()
    assert float(torch.quantile(rel, 0.999)) <= 2e-6, float(torch.quantile(rel, 0.999))
    # (one switched hidden unit moves a summed gradient by one pixel's share of it: ~1e-4 .. 1e-3 of its maximum at these sizes)
    for ga, gb, nm in zip(a[3:], b[3:], ("dW0", "db0", "dW1", "db1", "dW2", "db2")):
        assert float((ga - gb).abs().max() / ga.abs().max()) <= 1e-3, nm
        assert float((ga - gb).abs().median() / ga.abs().max()) <= 2e-6, nm        # ... and only the rows / columns of that unit
    # a different gradient than the promised one is noticed ON THE DEVICE (always on, no synchronisation) and poisons the
    # gradients with NaN rather than handing over gradients computed for the promised value
    xt = t(x).requires_grad_()
    params = [t(w).requires_grad_() for w in ws]
    y = ops.decoder_apply(xt, acts, params, fused=True, mse_target=tgt, mse_gloss=gl)
    (3.0 * gl * ops.mse_loss(y, tgt)).backward()
    torch.cuda.synchronize()
    for p_ in params:
        assert bool(torch.isnan(p_.grad).all()), "a broken gloss promise must poison the decoder gradients"


@pytest.mark.parametrize("n", [(1, 3), (7, 3), (1000, 3), (4099, 4), (2 ** 18 + 5, 3)])
def test_mse_kernels_vs_numpy(ops, n):
    """csrc/loss.hip: value and gradient of torch.nn.MSELoss (reference utils.py:99), incl. a non-unit upstream gradient,
    repeated launches (the kernel's self-resetting counter) and sizes that are not multiples of the vector width."""
    rng = np.random.default_rng(n[0])
    a, b = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
    want = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    for rep in range(3):
        x = t(a).requires_grad_()
        loss = ops.mse_loss(x, t(b))
        (loss * 3.0).backward()
        close(loss.detach().cpu().numpy(), want, 2e-6, 0)
        close(x.grad.cpu().numpy(), 3.0 * 2.0 * (a.astype(np.float64) - b) / a.size, 1e-6, 1e-12)


def _adam_params(seed, dev, sizes=((3, 5), (1000,), (2, 2049), (1 << 16, 2), (7,))):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in sizes]


def test_fused_adam_matches_torch_adam_over_steps(ops):
    """csrc/optim.hip vs torch.optim.Adam with get_optimizer's hyper-parameters (functions.py:96-127): three groups with
    their own lr / weight decay, odd sizes (tails, unaligned segments), 12 steps; then the state dicts are swapped."""
    from collision_handling_in_instantngp_amd.train import FusedAdam
    pa, pb = _adam_params(0, DEV), _adam_params(0, DEV)

    def groups(ps):
        return [{"params": ps[:2], "lr": 1e-2, "weight_decay": 1e-6}, {"params": ps[2:4], "lr": 1e-3, "weight_decay": 0.0},
                {"params": ps[4:], "lr": 5e-3, "weight_decay": 1e-2}]
    oa = FusedAdam(groups(pa), betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam(groups(pb), betas=(0.9, 0.99), eps=1e-15)
    gen = torch.Generator().manual_seed(5)
    for step in range(12):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=gen).to(DEV) * (10.0 ** (step % 4 - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        close(a.detach(), b.detach().cpu().numpy(), 2e-6, 1e-6)     # 12 roundings of a parameter of magnitude ~1
        for k in ("exp_avg", "exp_avg_sq"):                                # gradients span 1e-2 .. 10: absolute error follows the largest
            want = ob.state[b][k].cpu().numpy()
            close(oa.state[a][k], want, 2e-6, 2e-6 * float(np.abs(want).max()))
        assert float(oa.state[a]["step"]) == 12.0
    # a torch.optim.Adam state dict (the reference's whole_opt.pt) continues in the kernel, and the other way round
    oa2 = FusedAdam(groups(pa), betas=(0.9, 0.99), eps=1e-15)
    oa2.load_state_dict(ob.state_dict())
    ob2 = torch.optim.Adam(groups(pb), betas=(0.9, 0.99), eps=1e-15)
    ob2.load_state_dict(oa.state_dict())
    for a, b in zip(pa, pb):
        gr = torch.randn(a.shape, generator=gen).to(DEV)
        a.grad, b.grad = gr.clone(), gr.clone()
    oa2.step(); ob2.step()
    for a, b in zip(pa, pb):
        close(a.detach(), b.detach().cpu().numpy(), 4e-6, 2e-6)
    assert float(oa2.state[pa[0]]["step"]) == 13.0


def test_fused_adam_step_replays_from_a_graph(ops):
    """the device-side step counter makes the update capturable: 4 replays == 4 eager torch steps on the same gradient"""
    from collision_handling_in_instantngp_amd.train import FusedAdam
    pa, pb = _adam_params(3, DEV), _adam_params(3, DEV)
    oa = FusedAdam(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    for a, b in zip(pa, pb):
        a.grad = torch.randn_like(a); b.grad = a.grad.clone()
    oa.step(); ob.step()                                  # eager first step: state exists, pointers are settled
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    static = [a.grad.clone() for a in pa]
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            for a, g_ in zip(pa, static):
                a.grad = g_ * 1.0                          # gradients allocated inside the capture: new addresses, so the
            oa.step()                                      # segment table is uploaded by a copy node of the graph
    torch.cuda.current_stream().wait_stream(side)
    ob.step()                                             # capture does not execute: replay below is step 2
    graph.replay()
    for _ in range(3):
        graph.replay(); ob.step()
    torch.cuda.synchronize()
    assert float(oa.state[pa[0]]["step"]) == 5.0
    for a, b in zip(pa, pb):
        close(a.detach(), b.detach().cpu().numpy(), 3e-6, 1e-6)
    # an eager step between replays uses its own table; the next replay restores the captured one
    for a, b, g_ in zip(pa, pb, static):
        a.grad = g_.clone(); b.grad = g_.clone()
    oa.step(); ob.step()
    graph.replay(); ob.step()
    torch.cuda.synchronize()
    assert float(oa.state[pa[0]]["step"]) == 7.0
    for a, b in zip(pa, pb):
        close(a.detach(), b.detach().cpu().numpy(), 4e-6, 2e-6)


@pytest.mark.parametrize("L,T", [(3, 1000), (16, 1 << 15), (1, 77)])
def test_js_kl_kernels_vs_float64_and_torch_expression(ops, L, T):
    """Loss.js_kl_rows (utils.py:122-174) on the HIP kernels: value and gradient vs a float64 evaluation of the same
    formula, and vs the torch expression the CPU tests pin to the reference's goldens"""
    from collision_handling_in_instantngp_amd.train import Loss
    rng = np.random.default_rng(L * 7 + T)
    z = rng.standard_normal((L, T)) * 2.0
    p = np.exp(z - z.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    p = p.astype(np.float32)
    gamma, eps = 1.0, 0.5
    loss = Loss(gamma=gamma, epsilon=eps)
    pt = t(p).requires_grad_()
    out = loss.js_kl_rows(pt)
    w = rng.standard_normal(L).astype(np.float32)
    (out * t(w)).sum().backward()
    pd = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    want = loss.js_kl_rows_torch(pd)
    (want * torch.tensor(w, dtype=torch.float64)).sum().backward()
    close(out, want.detach().numpy(), 2e-6, 1e-9)
    scale = float(pd.grad.abs().max())
    close(pt.grad, pd.grad.numpy(), 2e-5, 1e-6 * scale)
    # the fp32 torch expression (what ran before) agrees to its own rounding
    pt2 = t(p).requires_grad_()
    ref = loss.js_kl_rows_torch(pt2)
    close(out, ref.detach().cpu().numpy(), 2e-4, 1e-8)


def _decoder_case(P, leaky, out_dim, seed):
    rng = np.random.default_rng(seed)
    x = (0.5 * rng.standard_normal((P, 32))).astype(np.float32)
    tgt = rng.random((P, out_dim)).astype(np.float32)
    dims = [32, 64, 64, out_dim]
    ws = []
    for i in range(3):
        ws += [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32), (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)]
    return x, tgt, ws


def _decoder_float64(x, tgt, ws, leaky, gl):
    """float64 evaluation of decoder + MSELoss + backward seeded with gl (reference models.py:382-392,469-470, utils.py:99)."""
    x64 = torch.tensor(x, dtype=torch.float64, device=DEV, requires_grad=True)
    p64 = [torch.tensor(w, dtype=torch.float64, device=DEV, requires_grad=True) for w in ws]
    actf = (lambda v: torch.nn.functional.leaky_relu(v, 0.01)) if leaky else torch.relu
    y64 = torch.sigmoid(actf(actf(x64 @ p64[0].T + p64[1]) @ p64[2].T + p64[3]) @ p64[4].T + p64[5])
    loss = ((y64 - torch.tensor(tgt, dtype=torch.float64, device=DEV)) ** 2).mean()
    (gl * loss).backward()
    return [y64.detach(), loss.detach(), x64.grad] + [p.grad for p in p64]


def _decoder_errors(got, want):
    """[rgb max abs, loss rel, d enc 99.9 % quantile of the per-pixel error over the largest |d enc|, six weight gradients
    max abs over max abs] — a hidden unit whose pre-activation is within rounding of 0 may switch sides, which changes that
    pixel's d enc and moves a summed gradient by that pixel's share (hence the quantile and the absolute slack below)."""
    e = [float((got[0].double() - want[0]).abs().max()), float(abs(float(got[1]) - float(want[1])) / max(abs(float(want[1])), 1e-30))]
    rel = (got[2].double() - want[2]).abs().max(1).values / want[2].abs().max()
    e.append(float(torch.quantile(rel[:2 ** 20], 0.999)))
    e += [float((g.double() - w).abs().max() / w.abs().max()) for g, w in zip(got[3:], want[3:])]
    return e


@pytest.mark.parametrize("P,leaky,gl,out_dim", [(1, False, 1.0, 3), (127, True, 1.0, 3), (40001, False, 0.37, 3), (2 ** 20 + 5, False, 1.0, 3),
                                               (5000, True, 2.5, 1)])
def test_training_decoder_kernels_against_float64(ops, P, leaky, gl, out_dim):
    """The kernel bench.py times (gngf_decoder_train: forward + MSE gradient + hybrid backward in one launch) and the two
    variants of the default two-kernel path (hybrid backward on saved hidden layers / fp32 backward recomputing them), each
    against a float64 evaluation of decoder + MSELoss + backward: rgb, the loss value, d enc and all six weight gradients.
    Criterion: at most twice the error of the all-fp32 MFMA kernels on the same inputs (+ a small absolute slack for the
    summed gradients, see _decoder_errors)."""
    from collision_handling_in_instantngp_amd import _lib
    x, tgt, ws = _decoder_case(P, leaky, out_dim, seed=P)
    want = _decoder_float64(x, tgt, ws, leaky, gl)
    acts = (ops.ACT_LEAKY if leaky else ops.ACT_RELU,) * 2 + (ops.ACT_SIGMOID,)
    tt = t(tgt)
    errs = {}
    for which in ("fp32", "train", "hybrid_saved", "fp32_recompute"):
        old = (ops.DECODER_TRAIN_FUSION, ops.DECODER_SAVE_HIDDEN)
        prev_h = _lib.query("gngf_set_decoder_bwd_hybrid", 0 if which == "fp32" else 1)
        ops.DECODER_TRAIN_FUSION = which == "train"
        ops.DECODER_SAVE_HIDDEN = which != "fp32_recompute"
        _lib.PROFILE = {}
        try:
            xt = t(x).requires_grad_()
            params = [t(w).requires_grad_() for w in ws]
            y = ops.decoder_apply(xt, acts, params, fused=True, mse_target=tt, mse_gloss=(gl if which == "train" else None))
            loss = ops.mse_loss(y, tt)
            (loss if gl == 1.0 else gl * loss).backward()
            torch.cuda.synchronize()
            names = set(_lib.PROFILE)
        finally:
            _lib.PROFILE = None
            ops.DECODER_TRAIN_FUSION, ops.DECODER_SAVE_HIDDEN = old
            _lib.query("gngf_set_decoder_bwd_hybrid", prev_h)
        assert ("gngf_decoder_train" in names) == (which == "train"), (which, names)
        got = [y.detach(), loss.detach(), xt.grad] + [p.grad for p in params]
        assert all(bool(torch.isfinite(g_).all()) for g_ in got), which
        errs[which] = _decoder_errors(got, want)
    base = errs["fp32"]
    names = ("rgb", "loss", "d enc", "dW0", "db0", "dW1", "db1", "dW2", "db2")
    for which in ("train", "hybrid_saved", "fp32_recompute"):
        for k, nm in enumerate(names):
            slack = 1e-7 if k == 0 else (1e-6 if k == 1 else (1e-7 if k == 2 else 2e-4))
            parity_close(np.array([errs[which][k]]), np.array([0.0]), 0, 2 * base[k] + slack,
                         f"{which} vs float64: {nm} (P={P}; all-fp32 kernels: {base[k]:.2e})")


@pytest.mark.parametrize("fused_train", [False, True])
def test_decoder_gradients_accumulate_over_two_backward_passes(ops, fused_train):
    """Gradient accumulation (two micro-batches without zero_grad, zero_grad(set_to_none=False), a second backward): the six
    weight gradients are filled by a slab reduction that normally rides a LATER launch of the backward pass — with an existing
    .grad autograd adds the returned views right away, so the reduction must run at once.  grad after two passes == 2 x one."""
    from collision_handling_in_instantngp_amd import _lib
    x, tgt, ws = _decoder_case(30001, False, 3, seed=77)
    acts = (ops.ACT_RELU,) * 2 + (ops.ACT_SIGMOID,)
    tt = t(tgt)
    params = [t(w).requires_grad_() for w in ws]
    xt = t(x).requires_grad_()

    def one_pass():
        y = ops.decoder_apply(xt, acts, params, fused=True, mse_target=tt, mse_gloss=(1.0 if fused_train else None))
        ops.mse_loss(y, tt).backward()
    _lib.PROFILE = {}
    try:
        one_pass()
        torch.cuda.synchronize()
        names = set(_lib.PROFILE)
    finally:
        _lib.PROFILE = None
    assert ("gngf_decoder_train" in names) == fused_train
    single = [p.grad.clone() for p in params] + [xt.grad.clone()]
    one_pass()                                   # no zero_grad in between
    torch.cuda.synchronize()
    for g1, p_ in zip(single, params + [xt]):
        assert bool(torch.isfinite(p_.grad).all())
        parity_close(p_.grad, (2.0 * g1).cpu().numpy(), 1e-6, 1e-7 * float(g1.abs().max()), "accumulated decoder gradient == 2 x single pass")
    # zero_grad(set_to_none=False): the gradient tensors stay, so the next pass accumulates into zeros
    for p_ in params + [xt]:
        p_.grad.zero_()
    one_pass()
    torch.cuda.synchronize()
    for g1, p_ in zip(single, params + [xt]):
        parity_close(p_.grad, g1.cpu().numpy(), 1e-6, 1e-7 * float(g1.abs().max()), "decoder gradient after zero_grad(set_to_none=False)")


def test_decoder_slab_reduction_riding_on_the_encoder_backward_gives_the_same_gradients(ops):
    """ops.DECODER_REDUCE_RIDES: gngf_decoder_bwd stops at its slabs and the reduction runs as extra workgroups of the tiled
    encoder backward's launch (or, when none follows, on its own at the end of the backward pass) == the one-call form, bit
    for bit; the per-slab maxima handed to the encoder backward bound |d enc|."""
    from oracle import gngf_oracle as orc
    rng = np.random.default_rng(4)
    P = 40000
    params = [t((rng.standard_normal(s) * 0.3).astype(np.float32)) for s in ((64, 32), (64,), (64, 64), (64,), (3, 64), (3,))]
    gy = t(rng.standard_normal((P, 3)).astype(np.float32))
    acts = (ops.ACT_RELU, ops.ACT_RELU, ops.ACT_SIGMOID)
    n_ls = orc.level_resolutions(16, 256, 16)
    n_host = [int(n) for n in n_ls]
    xy = t(rng.random((P, 2), dtype=np.float32))
    tables = t((rng.random((16, 4096, 2), dtype=np.float32) - 0.5) * 2e-1)
    x = t(rng.standard_normal((P, 32)).astype(np.float32))
    res = {}
    for with_encoder in (False, True):
        for rides in (False, True):
            ops.DECODER_REDUCE_RIDES = rides
            try:
                ps = [p.clone().requires_grad_() for p in params]
                link = ops.StepLink()
                if with_encoder:
                    tt = tables.clone().requires_grad_()
                    xs = ops.encode_apply(xy, t(n_ls, torch.int32), n_host, tt, None, None, 0, path="tiled", link=link)
                else:
                    tt = None
                    xs = x.clone().requires_grad_()
                ops.decoder_apply(xs, acts, ps, fused=True, link=link).backward(gy)
                torch.cuda.synchronize()
                assert link.pending_reduce is None                  # picked up by the encoder backward, or flushed at the end
                lead = tt.grad if with_encoder else xs.grad
                res[(with_encoder, rides)] = [lead.clone()] + [p.grad.clone() for p in ps]
                if not with_encoder and rides:
                    (am, count, stride), _ptr, _ver = link.absmax
                    bound = float(torch.stack([am[i * stride] for i in range(count)]).max())
                    assert bound == float(xs.grad.abs().max())
            finally:
                ops.DECODER_REDUCE_RIDES = True
        for k, (a, b) in enumerate(zip(res[(with_encoder, False)], res[(with_encoder, True)])):
            if with_encoder and k == 0:          # table gradient: float atomics of the hash vertex stage, any order
                assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
            else:
                assert torch.equal(a, b)


def test_fused_pixel_loss_equals_the_separate_loss_kernels(ops):
    """DecoderMseFunction (loss value in the decoder forward's epilogue, loss gradient formed in the backward's prologue) ==
    decoder + ops.MseFunction: every gradient bit for bit (the same d rgb expression), the value to the rounding of a
    different summation order; ragged sizes, a scaled loss, one output channel, and rgb feeding a second consumer."""
    rng = np.random.default_rng(12)
    acts = (ops.ACT_RELU, ops.ACT_RELU, ops.ACT_SIGMOID)
    for P, in_dim, out_dim in ((40001, 32, 3), (130, 32, 3), (5000, 8, 1), (2 ** 16, 64, 4)):
        params = [t((rng.standard_normal(s) * 0.3).astype(np.float32)) for s in ((64, in_dim), (64,), (64, 64), (64,), (out_dim, 64), (out_dim,))]
        x = t(rng.standard_normal((P, in_dim)).astype(np.float32))
        tgt = t(rng.random((P, out_dim), dtype=np.float32))
        res = {}
        for fused in (False, True):
            xs = x.clone().requires_grad_()
            ps = [p.clone().requires_grad_() for p in params]
            rgb = ops.decoder_apply(xs, acts, ps, fused=True, mse_target=(tgt if fused else None))
            assert (getattr(rgb, "_gngf_fused_mse", None) is not None) == fused
            loss = ops.mse_loss(rgb, tgt)
            (loss * 3.0).backward()
            res[fused] = (loss.detach().clone(), rgb.detach().clone(), [xs.grad.clone()] + [p.grad.clone() for p in ps])
        assert torch.equal(res[False][1], res[True][1])
        close(res[True][0], res[False][0].cpu().numpy(), 2e-6, 0, f"fused MSE value vs mse_fwd kernel (P={P})")
        want = ((res[False][1].double() - tgt.double()) ** 2).mean()
        close(res[True][0], want.cpu().numpy(), 2e-6, 0, f"fused MSE value vs float64 (P={P})")
        for a, b in zip(res[False][2], res[True][2]):
            assert torch.equal(a, b)
    # rgb also feeds another consumer: gradients add up
    xs = x.clone().requires_grad_()
    ps = [p.clone().requires_grad_() for p in params]
    rgb = ops.decoder_apply(xs, acts, ps, fused=True, mse_target=tgt)
    (ops.mse_loss(rgb, tgt) + 0.5 * rgb.sum()).backward()
    xs2 = x.clone().requires_grad_()
    ps2 = [p.clone().requires_grad_() for p in params]
    rgb2 = ops.decoder_apply(xs2, acts, ps2, fused=True)
    (ops.mse_loss(rgb2, tgt) + 0.5 * rgb2.sum()).backward()
    for a, b in zip([xs.grad] + [p.grad for p in ps], [xs2.grad] + [p.grad for p in ps2]):
        scale = float(b.abs().max())
        close(a, b.cpu().numpy(), 1e-5, 1e-6 * scale)
    # a different label tensor does not pick the fused value up
    other = tgt.clone()
    rgb3 = ops.decoder_apply(x, acts, params, fused=True, mse_target=tgt)
    assert ops.mse_loss(rgb3, other) is not rgb3._gngf_fused_mse[1]


def test_decoder_bwd_reports_its_device_clock_span(ops):
    """gngf_decoder_bwd_last_span_ns: first workgroup start -> last workgroup end of the most recent backward kernel"""
    import ctypes
    from collision_handling_in_instantngp_amd._lib import call
    rng = np.random.default_rng(9)
    P = 1 << 16
    x = t(rng.standard_normal((P, 32)).astype(np.float32)).requires_grad_()
    params = [t((rng.standard_normal(s) * 0.3).astype(np.float32)).requires_grad_()
              for s in ((64, 32), (64,), (64, 64), (64,), (3, 64), (3,))]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y = ops.decoder_apply(x, (ops.ACT_RELU, ops.ACT_RELU, ops.ACT_SIGMOID), params, fused=True)
    g = torch.ones_like(y)
    torch.cuda.synchronize()
    e0.record(); y.backward(g); e1.record()
    torch.cuda.synchronize()
    ns = ctypes.c_double(-1.0)
    call("gngf_decoder_bwd_last_span_ns", ctypes.byref(ns))
    assert 1e3 < ns.value < e0.elapsed_time(e1) * 1e6          # longer than a microsecond, inside the bracketing events


def test_fused_adam_fp16_parameters_keep_fp32_master_and_moments(ops):
    """fp16 level tables (BASELINE config 5): the update is applied to an fp32 master copy with fp32 moments and the fp16
    parameter is its rounding — compared with torch.optim.Adam on fp32 clones fed the same (un-scaled) gradients, 10 steps,
    mixed with fp32 parameters in the same launch; loss scaling through `grad_scale`; state dict round trip keeps fp32."""
    from collision_handling_in_instantngp_amd.train import FusedAdam
    g = torch.Generator().manual_seed(21)
    shapes = [(1 << 15, 4), (999,), (3, 7)]
    p16 = [torch.nn.Parameter((torch.randn(s, generator=g) * 1e-2).half().to(DEV)) for s in shapes]
    p32 = [torch.nn.Parameter(torch.randn((64, 32), generator=g).to(DEV))]
    ref = [torch.nn.Parameter(p.detach().float().clone()) for p in p16] + [torch.nn.Parameter(p32[0].detach().clone())]
    scale = 1024.0
    oa = FusedAdam([{"params": p16, "lr": 1e-3, "weight_decay": 0.0}, {"params": p32, "lr": 1e-2, "weight_decay": 1e-6}],
                   betas=(0.9, 0.99), eps=1e-15)
    oa.grad_scale = scale
    ob = torch.optim.Adam([{"params": ref[:3], "lr": 1e-3, "weight_decay": 0.0}, {"params": ref[3:], "lr": 1e-2, "weight_decay": 1e-6}],
                          betas=(0.9, 0.99), eps=1e-15)
    for step in range(10):
        for a, b in zip(p16 + p32, ref):
            gr = torch.randn(a.shape, generator=g).to(DEV) * 1e-3
            a.grad = (gr * scale).to(a.dtype)                       # what a scaled loss hands the optimizer
            b.grad = a.grad.float() / scale                         # the same values, un-scaled, in fp32
        oa.step(); ob.step()
    for a, b in zip(p16, ref[:3]):
        st = oa.state[a]
        assert st["master"].dtype == torch.float32 and st["exp_avg"].dtype == torch.float32
        close(st["master"], b.detach().cpu().numpy(), 2e-6, 1e-8)                          # fp32 trajectory
        assert torch.equal(a.detach(), st["master"].half())                                 # parameter = round(master)
        close(st["exp_avg_sq"], ob.state[b]["exp_avg_sq"].cpu().numpy(), 1e-5, 1e-14)
    close(p32[0].detach(), ref[3].detach().cpu().numpy(), 2e-6, 1e-6)
    sd = oa.state_dict()
    ob2 = FusedAdam([{"params": p16, "lr": 1e-3, "weight_decay": 0.0}, {"params": p32, "lr": 1e-2, "weight_decay": 1e-6}],
                    betas=(0.9, 0.99), eps=1e-15)
    ob2.load_state_dict(sd)
    assert ob2.state[p16[0]]["master"].dtype == torch.float32 and ob2.state[p16[0]]["exp_avg_sq"].dtype == torch.float32
    assert torch.equal(ob2.state[p16[0]]["master"], oa.state[p16[0]]["master"])


@pytest.mark.parametrize("M,T,K", [(128, 1024, 4), (256, 8192, 4), (128, 4096, 1), (384, 2048, 8), (128, 128, 2)])
def test_logits_gemm_with_epilogue_statistics_equals_the_separate_pass(ops, M, T, K):
    """Round 4: gngf_linear_fwd_rowstats leaves (max, sum exp) per row and 64-column block in the GEMM's epilogue and
    gngf_rowstats_topk merges them into the row statistics and reads only the K blocks with the largest maxima for the top-K —
    against gngf_linear_fwd (same split-bf16 kernel) + gngf_logits_topk_pbar, which reads every logit (reference models.py:85,
    105-116): logits bit-identical, top-K indices identical (ties -> lower index, also across blocks), probabilities and row
    statistics to fp32 rounding; NaN rows (nan_to_num: all zero, slots 0..K-1); p-bar through gngf_pbar_accumulate."""
    from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr, query
    rng = np.random.default_rng(M + T + K)
    Kc = 128
    x = t((rng.standard_normal((M, Kc)) * 0.7).astype(np.float32))
    W = (rng.standard_normal((T, Kc)) * 0.5).astype(np.float32)
    b = (rng.standard_normal(T) * 0.3).astype(np.float32)
    # ties: make a few columns identical (same weights and bias) in different 64-column blocks, and dominant
    if T >= 256:
        W[5] = W[T - 3] = W[70] = 0.0
        b[5] = b[T - 3] = b[70] = 50.0
    Wt, bt = t(W), t(b)
    xx = x.clone()
    if M >= 3:
        xx[2, 7] = float("nan")                                  # a NaN input row -> NaN logits in that row
    z0 = torch.empty((M, T), device=DEV)
    z1 = torch.empty((M, T), device=DEV)
    parts = torch.empty((M, T // 64, 2), device=DEV)
    prev = query("gngf_set_gemm_split_bf16", 1)
    try:
        call("gngf_linear_fwd", ptr(xx), ptr(Wt), ptr(bt), ptr(z0), M, T, Kc, 0, stream_ptr())
    finally:
        query("gngf_set_gemm_split_bf16", prev)
    call("gngf_linear_fwd_rowstats", ptr(xx), ptr(Wt), ptr(bt), ptr(z1), ptr(parts), M, T, Kc, stream_ptr())
    assert torch.equal(torch.nan_to_num(z0, nan=-7.0), torch.nan_to_num(z1, nan=-7.0))
    Lv = 5
    mw = t(rng.random((M, Lv)).astype(np.float32))
    tv0, ti0, rs0 = torch.empty((M, K), device=DEV), torch.empty((M, K), dtype=torch.int32, device=DEV), torch.empty((M, 2), device=DEV)
    tv1, ti1, rs1 = torch.empty_like(tv0), torch.empty_like(ti0), torch.empty_like(rs0)
    pb0, pb1 = torch.zeros((Lv, T), device=DEV), torch.zeros((Lv, T), device=DEV)
    call("gngf_logits_topk_pbar", ptr(z0), ptr(tv0), ptr(ti0), ptr(rs0), ptr(mw), Lv, ptr(pb0), M, T, K, stream_ptr())
    call("gngf_rowstats_topk", ptr(z1), ptr(parts), ptr(tv1), ptr(ti1), ptr(rs1), M, T, K, stream_ptr())
    call("gngf_pbar_accumulate", ptr(z1), ptr(rs1), ptr(mw), Lv, ptr(pb1), M, T, stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(ti0, ti1), (ti0[:4], ti1[:4])
    if T >= 256 and K >= 3:
        good = [r for r in range(M) if r != 2]
        assert set(ti1[good[0]].tolist()[:3]) == {5, 70, T - 3} and ti1[good[0]].tolist()[:3] == [5, 70, T - 3]     # ties -> lower index
    ok = torch.ones(M, dtype=torch.bool, device=DEV)
    if M >= 3:
        ok[2] = False
        assert float(tv1[2].abs().max()) == 0.0 and ti1[2].tolist() == list(range(K)) and bool(torch.isnan(rs1[2, 1]))
    close(rs1[ok][:, 0], rs0[ok][:, 0], 0, 0, "row maxima from the epilogue partials")
    close(rs1[ok][:, 1], rs0[ok][:, 1], 1e-5, 0, "row sums exp(z - max) from the epilogue partials")
    close(tv1, tv0, 1e-5, 1e-12, "top-K probabilities from the epilogue partials")
    close(pb1, pb0, 1e-5, 1e-9, "p-bar from the merged statistics")
    # shapes the epilogue form does not take are REJECTED (the caller falls back), never computed wrongly
    import pytest as _pt
    with _pt.raises(RuntimeError):
        call("gngf_linear_fwd_rowstats", ptr(xx[:100].contiguous()), ptr(Wt), ptr(bt), ptr(z1), ptr(parts), 100, T, Kc, stream_ptr())


@pytest.mark.parametrize("mode", [1, 2, 17])
def test_split_gemms_with_bf16_planes_in_lds(ops, mode):
    """Round 5: the three T-wide GEMMs of the HashProbDistribution's last layer (reference models.py:84-85 and its autograd
    backward) with every operand value split ONCE into bf16 planes on the way into LDS (mode 1: three planes, six products;
    mode 2: two planes, three products, accumulating GEMMs only) against float64 and against the per-wave split kernel (mode 17):
    the three-plane kernel computes the same six terms in the same order — logits bit-identical; error over sum |a b| at the
    fp32 level for modes 1 / 17 and <= 3 * 2^-18 (its a-priori bound) for mode 2."""
    from collision_handling_in_instantngp_amd._lib import query
    rng = np.random.default_rng(5)
    n, T, H = 512, 2048, 128
    h = t((rng.standard_normal((n, H)) * 50).astype(np.float32))
    W = t(((rng.random((T, H)) * 2 - 1) / H ** 0.5).astype(np.float32))
    b = t(((rng.random(T) * 2 - 1) / H ** 0.5).astype(np.float32))
    # gradients with 30 decades between the columns, as softmax gradients have
    dz = t((rng.standard_normal((n, T)) * np.exp(rng.uniform(-60, 0, size=(1, T)))).astype(np.float32))

    def run(md):
        prev = query("gngf_set_gemm_split_bf16", md)
        try:
            z = ops.linear_fwd(h, W, b, ops.ACT_NONE)
            dW = torch.zeros_like(W)
            ops.linear_bwd_weight(dz, None, h, dW, None, ops.ACT_NONE)
            dh = torch.zeros((n, H), device=DEV)
            ops.gemm_acc(dz, W, dh, n, H, T, ta=False, tb=False)
            torch.cuda.synchronize()
        finally:
            query("gngf_set_gemm_split_bf16", prev)
        return z, dW, dh

    z, dW, dh = run(mode)
    z17, dW17, dh17 = run(17)
    hd, Wd, dzd = h.double(), W.double(), dz.double()
    zr, sz = hd @ Wd.T + b.double(), hd.abs() @ Wd.abs().T
    dWr, sW = dzd.T @ hd, dzd.abs().T @ hd.abs()
    dhr, sH = dzd @ Wd, dzd.abs() @ Wd.abs()
    ez = float(((z.double() - zr).abs() / sz).max())
    eW = float(((dW.double() - dWr).abs() / sW.clamp_min(1e-300)).max())
    eH = float(((dh.double() - dhr).abs() / sH).max())
    print(f"mode {mode}: |err| / sum|a b|  logits {ez:.2e}  dW {eW:.2e}  dh {eH:.2e}")
    assert ez <= 6e-7                                              # the logits keep the exact three-way split in every mode
    if mode != 17:
        assert torch.equal(z, z17)
    tol = 3 * 2.0 ** -18 if mode == 2 else 6e-7
    assert eW <= tol and eH <= tol, (eW, eH, tol)
    # every column of dW keeps its own leading digits (Adam normalises per element: a column 1e-20 of the largest must not come out as zero)
    small = dWr.abs().amax(dim=1) > 0
    rel_col = ((dW.double() - dWr).abs().amax(dim=1) / dWr.abs().amax(dim=1).clamp_min(1e-300))[small]
    assert float(rel_col.max()) <= (2e-4 if mode == 2 else 2e-5), float(rel_col.max())


@pytest.mark.parametrize("planes", [3, 2])
@pytest.mark.parametrize("U,T,L,K", [(256, 2048, 16, 4), (128, 1024, 5, 1), (384, 4096, 16, 0), (128, 512, 0, 3), (1024, 131072, 16, 4)])
def test_softmax_backward_formed_in_the_gemm_loaders(ops, U, T, L, K, planes):
    """Round 5: gngf_hpd_bwd_dot + gngf_hpd_bwd_fused (dz = p (mw G - dot) + top-K terms formed inside the dW / dh GEMMs, which read
    the logits) against float64 autograd-equivalent algebra (reference models.py:84-85,105-116; utils.py:138,159) and against the
    three separate entry points they replace (gngf_softmax_bwd_lowrank + gngf_linear_bwd_weight + gngf_gemm_acc)."""
    from collision_handling_in_instantngp_amd._lib import call, ptr, stream_ptr, query
    rng = np.random.default_rng(U + T + L + K)
    Hd = 128
    z = t((rng.standard_normal((U, T)) * 4).astype(np.float32))
    h = t(np.maximum(rng.standard_normal((U, Hd)) * 30, 0).astype(np.float32))
    W = t(((rng.random((T, Hd)) * 2 - 1) / Hd ** 0.5).astype(np.float32))
    mw = t((rng.random((U, max(L, 1))) / (4 * U)).astype(np.float32)) if L else None
    G = t((rng.standard_normal((max(L, 1), T)) * 3).astype(np.float32)) if L else None
    zd = z.double()
    m = zd.max(dim=1, keepdim=True).values
    ssum = torch.exp(zd - m).sum(dim=1, keepdim=True)
    p = torch.exp(zd - m) / ssum
    rowstat = torch.cat([m, ssum], dim=1).float().contiguous()
    g = (mw.double() @ G.double()) if L else torch.zeros_like(zd)
    if K:
        tp, ti = torch.topk(p, K, dim=1)
        ti32 = ti.to(torch.int32).contiguous()
        pk = tp.float().contiguous()
        dq = t(rng.standard_normal((U, K)).astype(np.float32) * 1e-3)
        g = g.scatter_add(1, ti, dq.double())
    else:
        ti32 = pk = dq = None
    dot = (p * g).sum(dim=1, keepdim=True)
    dz = p * (g - dot)
    dWr, dbr, dHr = dz.T @ h.double(), dz.sum(dim=0), dz @ W.double()

    dotv = torch.empty((U,), device=DEV)
    dW, db, dH = torch.zeros((T, Hd), device=DEV), torch.zeros((T,), device=DEV), torch.zeros((U, Hd), device=DEV)
    assert query("gngf_hpd_bwd_fused_applies", U, T, L, K, Hd) == 1
    call("gngf_hpd_bwd_dot", ptr(z), ptr(rowstat), ptr(dq), ptr(pk), ptr(mw), ptr(G), L, ptr(dotv), U, T, K, stream_ptr())
    # the small operands split once: prepared for a LARGER row set (the step's vertices), this chunk starting at row u0 of it
    u0, rows_total = 128, U + 256
    h_all = torch.cat([torch.full((u0, Hd), 7.0, device=DEV), h, torch.full((rows_total - U - u0, Hd), -3.0, device=DEV)])
    mw_all = torch.cat([torch.ones((u0, max(L, 1)), device=DEV), mw if L else torch.zeros((U, 1), device=DEV),
                        torch.ones((rows_total - U - u0, max(L, 1)), device=DEV)]) if L else None
    prep = ops._HpdBwdPlanes(h_all, mw_all, W, G, L, planes)
    if L:
        back = prep.mwp.view(torch.bfloat16).float().double().sum(dim=0)[u0:u0 + U, :L]
        assert torch.equal(back.float(), mw) and float(prep.mwp[:, :, L:].abs().max() if L < 16 else 0) == 0
        assert torch.equal(prep.Gtp.view(torch.bfloat16).float().double().sum(dim=0)[:, :L].float(), G.T.contiguous())
    prep.fused(z, rowstat, dotv, dq, pk, ti32, h, W, dW, db, dH, u0, U, T, K)
    # the three entry points it replaces
    dz1 = z.clone()
    scratch = torch.empty((U * (1 + max(K, 1)),), device=DEV)
    dW1, db1, dH1 = torch.zeros_like(dW), torch.zeros_like(db), torch.zeros_like(dH)
    call("gngf_softmax_bwd_lowrank", ptr(dz1), ptr(rowstat), ptr(dq), ptr(ti32), ptr(mw), ptr(G), L, ptr(db1), ptr(scratch), ptr(pk),
         U, T, K, stream_ptr())
    ops.linear_bwd_weight(dz1, None, h, dW1, None, ops.ACT_NONE)
    ops.gemm_acc(dz1, W, dH1, U, Hd, T, ta=False, tb=False)
    torch.cuda.synchronize()

    def rel(a, ref):
        return float((a.double() - ref).abs().max() / ref.abs().max())
    close(dotv, dot[:, 0].float(), 2e-5, 1e-9, "row dots")
    e = {"dW": rel(dW, dWr), "db": rel(db, dbr), "dH": rel(dH, dHr)}
    e1 = {"dW": rel(dW1, dWr), "db": rel(db1, dbr), "dH": rel(dH1, dHr)}
    print(f"U={U} T={T} L={L} K={K} planes={planes}: fused |err|/max {e}   separate {e1}")
    for k in e:
        assert e[k] <= max(2e-5, 4 * e1[k]), (k, e, e1)
    # shapes the fused form does not take are rejected, never computed wrongly
    assert query("gngf_hpd_bwd_fused_applies", U + 3, T, L, K, Hd) == 0 and query("gngf_hpd_bwd_fused_applies", U, T, 17, K, Hd) == 0
    with pytest.raises(RuntimeError):
        prep.fused(z, rowstat, dotv, dq, pk, ti32, h, W, dW, db, dH, u0, U + 3, T, K)
