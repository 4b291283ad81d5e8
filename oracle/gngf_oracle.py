"""CPU ORACLE (numpy, fp32) — a restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (collision_handling_in_instantngp_amd/) never does and has no CPU fallback.

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py against
golden vectors captured by running the Python reference itself on CPU (oracle/make_goldens.py,
fixtures in tests/golden/).  The reference has no tests/fixtures of its own.

Each function cites the reference lines it restates (paths relative to the reference root).
The restatement is deliberately literal (per-instance, dense (P,L,4,T) distributions, no
de-duplication) so that the product's restructured algorithm is checked against the original
formulation, at sizes where that formulation is feasible.

Layouts: x (P,2) fp32 = (row, col) in [0,1];  tables (L,T,F) fp32;  idx (P,L,4) or (P,L,4,K) int64;
corner order v = dx + 2*dy with dx on input dim 0 (models.py:322-331).
"""
import numpy as np

f32 = np.float32

PRIMES = (1, 2654435761, 805459861)  # models.py:346


# --------------------------------------------------------------------------- init-time helpers
def level_resolutions(n_min: int, n_max: int, num_levels: int) -> np.ndarray:
    """models.py:305-317 — b = exp((ln n_max - ln n_min)/(L-1)); N_l = floor(n_min * b**l) in float64 -> int32."""
    b = np.exp((np.log(n_max) - np.log(n_min)) / (num_levels - 1))
    return np.array([np.floor(n_min * b ** l) for l in range(num_levels)]).astype(np.int32)


def corner_offsets(input_dim: int = 2) -> np.ndarray:
    """models.py:322-331 — hypercube (input_dim, 2**input_dim): dim0 = [0,1,0,1], dim1 = [0,0,1,1]."""
    h = np.empty((input_dim, 2 ** input_dim), dtype=np.int32)
    for i in range(input_dim):
        h[i] = np.array(([0] * (2 ** i) + [1] * (2 ** i)) * (2 ** (input_dim - i - 1)), dtype=np.int32)
    return h


# --------------------------------------------------------------------------- a5 / a6
def scale_to_grid(x: np.ndarray, n_ls: np.ndarray):
    """models.py:486-502 — scaled = x * N_l (fp32; int32 N_l promoted), grid = floor(scaled) + hypercube.
    Returns scaled (P,2,L,1) fp32, grid (P,2,L,4) fp32 exactly as the reference shapes them."""
    x = np.asarray(x, dtype=f32)
    scaled = (x[:, :, None, None] * n_ls.astype(f32)[None, None, :, None]).astype(f32)
    grid = (np.floor(scaled) + corner_offsets(2).astype(f32)[None, :, None, :]).astype(f32)
    return scaled, grid


def spatial_hash(grid_int: np.ndarray, table_size: int) -> np.ndarray:
    """models.py:504-528.  NOTE (measured on the reference, torch 2.10): `grid[:, i] * self._prime_numbers[i]`
    multiplies an int32 tensor by a 0-dim int64 tensor; type promotion keeps int32, so the product WRAPS in
    32 bits (2654435761 itself wraps to -1640531535).  The XOR with the int64 zeros sign-extends, and
    torch.remainder is Python-style (non-negative).  grid_int: (P,2,L,4) int32 -> (P,L,4) int64."""
    g = grid_int.astype(np.int32)
    tmp = np.zeros((g.shape[0], g.shape[2], g.shape[3]), dtype=np.int64)
    for i in range(2):
        prod = (g[:, i].astype(np.int64) * np.int64(PRIMES[i])).astype(np.int32)  # int32 wrap-around
        tmp = np.bitwise_xor(prod.astype(np.int64), tmp)
    return np.mod(tmp, np.int64(table_size))


# --------------------------------------------------------------------------- a9-a11 encoding
def blend_weights(probs: np.ndarray, blend):
    """models.py:212-217 — blend = True: softmax over K; None: raw probs; False: probs / sum_K probs."""
    p = probs.astype(f32)
    if blend is None:
        return p
    if blend:
        e = np.exp(p - p.max(-1, keepdims=True)).astype(f32)
        return (e / e.sum(-1, keepdims=True, dtype=f32)).astype(f32)
    return (p / p.sum(-1, keepdims=True, dtype=f32)).astype(f32)


def encoding_forward(tables, idx, probs=None, blend=True):
    """MultiResHashEncoding.forward, models.py:173-229.  -> (P,F,L,4) fp32.
    hash branch (idx (P,L,4)): E_l[idx] permuted (models.py:181-191);
    GNGF branch (idx (P,L,4,K)): sum_k w_k * E_l[idx_k] (models.py:193-222)."""
    L = tables.shape[0]
    lvl = np.arange(L)
    if idx.ndim == 3:
        feats = tables[lvl[None, :, None], idx]                      # (P,L,4,F)
        return np.ascontiguousarray(feats.transpose(0, 3, 1, 2)).astype(f32)
    feats = tables[lvl[None, :, None, None], idx]                    # (P,L,4,K,F)
    w = blend_weights(probs, blend)                                  # (P,L,4,K)
    out = (feats * w[..., None]).astype(f32).sum(3, dtype=f32)       # (P,L,4,F)
    return np.ascontiguousarray(out.transpose(0, 3, 1, 2)).astype(f32)


def encoding_backward(tables, idx, probs, blend, gout):
    """Autograd of the above: table scatter-add (embedding_dense_backward) and d probs.
    gout (P,F,L,4).  Returns dtables (L,T,F) [accumulated in float64, rounded to fp32], dprobs or None."""
    L, T, F = tables.shape
    g = gout.transpose(0, 2, 3, 1).astype(np.float64)               # (P,L,4,F)
    dt = np.zeros((L, T, F), dtype=np.float64)
    lvl = np.arange(L)
    if idx.ndim == 3:
        np.add.at(dt, (np.broadcast_to(lvl[None, :, None], idx.shape), idx), g)
        return dt.astype(f32), None
    w = blend_weights(probs, blend).astype(np.float64)               # (P,L,4,K)
    np.add.at(dt, (np.broadcast_to(lvl[None, :, None, None], idx.shape), idx), g[:, :, :, None, :] * w[..., None])
    feats = tables[lvl[None, :, None, None], idx].astype(np.float64)  # (P,L,4,K,F)
    d = (feats * g[:, :, :, None, :]).sum(-1)                         # dL/dw_k  (P,L,4,K)
    p = probs.astype(np.float64)
    if blend is None:
        dp = d
    elif blend:
        dp = w * (d - (w * d).sum(-1, keepdims=True))
    else:
        s = p.sum(-1, keepdims=True)
        dp = d / s - (d * p).sum(-1, keepdims=True) / (s * s)
    return dt.astype(f32), dp.astype(f32)


# --------------------------------------------------------------------------- a12 bilinear
def bilinear_coeffs(scaled, grid):
    """models.py:626-637 — a = grid[...,0] (floor), d = grid[...,-1] (floor+1);
    c0=(xd-x)(yd-y) c1=(x-xa)(yd-y) c2=(xd-x)(y-ya) c3=(x-xa)(y-ya).  -> (P,L,4) fp32."""
    a = grid[:, :, :, 0]
    d = grid[:, :, :, -1]
    s = scaled[:, :, :, 0]
    c = np.stack([(d[:, 0] - s[:, 0]) * (d[:, 1] - s[:, 1]),
                  (s[:, 0] - a[:, 0]) * (d[:, 1] - s[:, 1]),
                  (d[:, 0] - s[:, 0]) * (s[:, 1] - a[:, 1]),
                  (s[:, 0] - a[:, 0]) * (s[:, 1] - a[:, 1])], axis=-1)
    return c.astype(f32)


def bilinear_forward(x, n_ls, feats):
    """_bilinear_interpolate, models.py:621-655.  feats (P,F,L,4) -> (P, L*F), level-major / feature-minor."""
    scaled, grid = scale_to_grid(x, n_ls)
    c = bilinear_coeffs(scaled, grid)                                # (P,L,4)
    ws = (feats * c[:, None]).astype(f32).sum(-1, dtype=f32)         # (P,F,L)
    return np.ascontiguousarray(ws.transpose(0, 2, 1)).reshape(feats.shape[0], -1)


def bilinear_backward(x, n_ls, gout, F):
    """d feats = c_v * g (coords carry no grad: _scale_to_grid is no_grad, models.py:486).  gout (P,L*F) -> (P,F,L,4)."""
    scaled, grid = scale_to_grid(x, n_ls)
    c = bilinear_coeffs(scaled, grid)
    P, L = c.shape[:2]
    g = gout.reshape(P, L, F).transpose(0, 2, 1)                     # (P,F,L)
    return (g[..., None] * c[:, None]).astype(f32)


# --------------------------------------------------------------------------- a13 decoder / generic MLP
def _act(z, kind, leaky_slope=0.01):
    if kind == "relu":
        return np.maximum(z, 0)
    if kind == "leaky":
        return np.where(z > 0, z, z * f32(leaky_slope))
    if kind == "sigmoid":
        return (1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(f32)
    raise ValueError(kind)


def decoder_forward(x, weights, biases, leaky=False, keep=False):
    """models.py:382-392,469-470 — Linear+ReLU|LeakyReLU … Linear+Sigmoid.  weights[i] is (out,in) like nn.Linear."""
    acts = [x.astype(f32)]
    pre = []
    n = len(weights)
    for i, (W, b) in enumerate(zip(weights, biases)):
        z = (acts[-1] @ W.T.astype(f32) + b.astype(f32)).astype(f32)
        pre.append(z)
        acts.append(_act(z, ("leaky" if leaky else "relu") if i < n - 1 else "sigmoid").astype(f32))
    return (acts[-1], acts, pre) if keep else acts[-1]


def decoder_backward(x, weights, biases, gy, leaky=False):
    """Standard Linear/ReLU/Sigmoid backward.  Returns dx, [dW], [db] (float64 accumulation -> fp32)."""
    y, acts, pre = decoder_forward(x, weights, biases, leaky, keep=True)
    n = len(weights)
    g = gy.astype(np.float64) * (y.astype(np.float64) * (1 - y.astype(np.float64)))
    dWs, dbs = [None] * n, [None] * n
    for i in range(n - 1, -1, -1):
        dWs[i] = (g.T @ acts[i].astype(np.float64)).astype(f32)
        dbs[i] = g.sum(0).astype(f32)
        g = g @ weights[i].astype(np.float64)
        if i > 0:
            z = pre[i - 1]
            g = g * (np.where(z > 0, 1.0, 0.01) if leaky else (z > 0))
    return g.astype(f32), dWs, dbs


# --------------------------------------------------------------------------- a7 / a8 HPD + top-K
def topk_desc(probs, k):
    """DifferentiableTopk.forward, models.py:11 — torch.topk(largest, sorted).  Tie order is unspecified in torch;
    the oracle (and the product) break ties towards the LOWER index."""
    order = np.argsort(-probs, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(probs, order, -1), order.astype(np.int64)


def hpd_forward(verts, weights, biases, k, keep=False):
    """HashProbDistribution.forward, models.py:90-123 — Linear+ReLU ×(n-1), Linear+Softmax(dim=-1), nan_to_num, top-K.
    verts (...,2) fp32 raw integer vertex coordinates.  Returns probs (...,T), topk_probs, topk_idx."""
    h = verts.astype(f32)
    acts = [h]
    n = len(weights)
    for i, (W, b) in enumerate(zip(weights, biases)):
        z = (h @ W.T.astype(f32) + b.astype(f32)).astype(f32)
        if i < n - 1:
            h = np.maximum(z, 0).astype(f32)
            acts.append(h)
    z = z - z.max(-1, keepdims=True)
    e = np.exp(z).astype(f32)
    sm = (e / e.sum(-1, keepdims=True, dtype=f32)).astype(f32)
    probs = np.nan_to_num(sm)                                         # models.py:111
    tp, ti = topk_desc(probs, k)
    if keep:
        return probs, tp, ti, acts, sm
    return probs, tp, ti


def hpd_backward(verts, weights, biases, k, d_topk, d_probs=None, topk_idx=None):
    """Autograd of hpd_forward: scatter d_topk into zeros at topk_idx (models.py:27-35), add the direct gradient
    on probs, nan_to_num backward (pass-through where finite), softmax backward, Linear/ReLU backward.
    `topk_idx` overrides the oracle's own selection (torch.topk's tie order is unspecified, so a golden
    comparison must route d_topk to the slots the reference picked).  Returns [dW], [db]."""
    probs, tp, ti, acts, sm = hpd_forward(verts, weights, biases, k, keep=True)
    if topk_idx is not None:
        ti = topk_idx.reshape(ti.shape)
    g = np.zeros(probs.shape, dtype=np.float64)
    np.put_along_axis(g, ti, d_topk.astype(np.float64), -1)
    if d_probs is not None:
        g = g + d_probs
    g = g * np.isfinite(sm)
    p = sm.astype(np.float64)
    g = p * (g - (g * p).sum(-1, keepdims=True))                      # softmax backward -> d logits
    n = len(weights)
    g = g.reshape(-1, g.shape[-1])
    dWs, dbs = [None] * n, [None] * n
    for i in range(n - 1, -1, -1):
        a = acts[i].reshape(-1, acts[i].shape[-1]).astype(np.float64)
        dWs[i] = (g.T @ a).astype(f32)
        dbs[i] = g.sum(0).astype(f32)
        if i > 0:
            g = (g @ weights[i].astype(np.float64)) * (a > 0)
    return dWs, dbs


# --------------------------------------------------------------------------- Loss
def _kldiv_batchmean(log_input, target):
    """torch.nn.KLDivLoss(reduction='batchmean') on 1-D input: sum(target*(log target - input)) / input.shape[0]
    with the 0*log0 = 0 convention (xlogy)."""
    t = target.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        term = np.where(t > 0, t * (np.log(t) - log_input.astype(np.float64)), 0.0)
    return term.sum() / log_input.shape[0]


def js_kl_terms(pbar, gamma, epsilon):
    """utils.py:122-174 for ONE level: pbar (N,) mean distribution; q uniform.
    kl = KLDiv(log pbar, q); js = (KLDiv(log pbar, m) + KLDiv(log q, m))/2 with m=(pbar+q)/2;
    returns -(gamma+epsilon)*js + epsilon*kl (utils.py:127)."""
    N = pbar.shape[0]
    q = np.full(N, 1.0 / N)
    with np.errstate(divide="ignore"):
        lp = np.log(pbar.astype(np.float64))
    kl = _kldiv_batchmean(lp, q)
    m = (pbar.astype(np.float64) + q) / 2
    js = (_kldiv_batchmean(lp, m) + _kldiv_batchmean(np.log(q), m)) / 2
    return -(gamma + epsilon) * js + epsilon * kl


def loss_forward(pred, labels, probs, gamma, epsilon):
    """Loss.forward, utils.py:91-120 (collision term handled by the caller: functions.py:243-245).
    probs (P,L,4,N) or None.  Returns mse, kls (L,) or None."""
    mse = np.mean((pred.astype(np.float64) - labels.astype(np.float64)) ** 2)
    if probs is None:
        return mse, None
    P, L, V, N = probs.shape
    kls = np.array([js_kl_terms(probs[:, l].astype(np.float64).sum(0).sum(0) / (P * V), gamma, epsilon)
                    for l in range(L)])
    return mse, kls


def djs_kl_dpbar(pbar, gamma, epsilon):
    """Analytic gradient of js_kl_terms w.r.t. pbar (N,).  With c = 1/N (batchmean on 1-D):
    d kl/dp = -c*q/p ;  d js/dp = c/2 * [ 0.5*(log m + 1 - log p) - m/p  + 0.5*(log m + 1 - log q) ]."""
    N = pbar.shape[0]
    p = pbar.astype(np.float64)
    q = 1.0 / N
    m = (p + q) / 2
    c = 1.0 / N
    dkl = -c * q / p
    djs = c / 2 * (0.5 * (np.log(m) + 1 - np.log(p)) - m / p + 0.5 * (np.log(m) + 1 - np.log(q)))
    return -(gamma + epsilon) * djs + epsilon * dkl


# --------------------------------------------------------------------------- Adam (functions.py:96-127)
def adam_step(p, g, m, v, step, lr, wd, betas=(0.9, 0.99), eps=1e-15):
    """torch.optim.Adam (non-amsgrad, L2 weight decay added to the gradient), single tensor, float32 state."""
    p = p.astype(f32); g = g.astype(f32)
    if wd != 0:
        g = (g + f32(wd) * p).astype(f32)
    m = (m + (g - m) * f32(1 - betas[0])).astype(f32)                 # lerp form used by torch
    v = (v * f32(betas[1]) + (g * g) * f32(1 - betas[1])).astype(f32)
    bc1 = 1 - betas[0] ** step
    bc2 = 1 - betas[1] ** step
    denom = (np.sqrt(v) / f32(np.sqrt(bc2)) + f32(eps)).astype(f32)
    p = (p - f32(lr / bc1) * (m / denom)).astype(f32)
    return p, m, v


def batch_norm_train(x, weight, bias, eps=1e-5):
    """nn.BatchNorm1d in training mode on (P,C) coordinates (models.py:340,394-397): batch mean, BIASED batch variance,
    fp32 arithmetic as ATen's CPU kernel (mean and variance accumulated in double, normalisation in float)."""
    x = x.astype(f32)
    mean = x.astype(np.float64).mean(0)
    var = x.astype(np.float64).var(0)
    invstd = (1.0 / np.sqrt(var + eps)).astype(f32)
    return ((x - mean.astype(f32)) * invstd * weight.astype(f32) + bias.astype(f32)).astype(f32)


# --------------------------------------------------------------------------- a14 end-to-end (literal)
def gngf_forward(x, n_ls, tables, dec_w, dec_b, *, hash_mode, hpd_w=None, hpd_b=None, K=4, blend=True, leaky=False):
    """GeneralNeuralGaugeFields.forward, models.py:394-484, literal per-instance formulation.
    Returns dict(rgb, enc (P,L*F), idx, probs, topk_probs)."""
    T = tables.shape[1]
    scaled, grid = scale_to_grid(x, n_ls)
    out = {}
    if hash_mode:
        idx = spatial_hash(grid.astype(np.int32), T)
        feats = encoding_forward(tables, idx)
        out.update(idx=idx, probs=None, topk_probs=None)
    else:
        rg = np.ascontiguousarray(grid.transpose(0, 2, 3, 1))          # "p xy l v -> p l v xy", models.py:416
        probs, tp, ti = hpd_forward(rg, hpd_w, hpd_b, K)
        feats = encoding_forward(tables, ti, tp, blend)
        out.update(idx=ti, probs=probs, topk_probs=tp)
    enc = bilinear_forward(x, n_ls, feats)
    out["enc"] = enc
    out["rgb"] = decoder_forward(enc, dec_w, dec_b, leaky)
    return out
