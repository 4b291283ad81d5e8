"""Drop-in module classes for the reference's hot path (reference models.py), backed by gfx950 HIP kernels.

Same class names, constructor signatures, forward signatures, return contracts and state-dict keys as the
reference:
    DifferentiableTopk         models.py:5-42
    HashProbDistribution       models.py:45-123
    MultiResHashEncoding       models.py:126-236
    GeneralNeuralGaugeFields   models.py:239-655
Behaviour switches are module globals read at call time, as in the reference (see params.py).

What is different inside: GeneralNeuralGaugeFields.forward never builds per-instance tensors.
HashProbDistribution is evaluated once per DISTINCT grid vertex (its input is the integer vertex only,
models.py:416-418), the top-K of each vertex becomes a per-vertex table, and one fused kernel does
corners + lookup + blend + bilinear interpolation.  The reference-shaped outputs (probs, indices) are
expanded from the per-vertex tables on request.  There is no CPU path: forward on CPU tensors raises.
"""
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .params import *  # noqa: F401,F403  (module-global behaviour switches, as in the reference)

device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")

# Largest reference-shaped dense tensor the module materialises on request (the (P,L,4,T) distribution).
DENSE_OUTPUT_LIMIT_BYTES = 4 << 30
# Row-chunk budget of the per-vertex (rows, T) distribution scratch: sized for 288 GB of HBM (4096 rows at T = 2^19; without
# kept logits the learning step is 1.36 / 1.29 / 1.27 / 1.265 s at 2 / 4 / 8 / 16 GiB — fewer, larger launches; 8 GiB pieces
# pack the kept-logits budget of ops.HPD_Z_CACHE_BYTES best: tools/ab_hpd_chunk.py).
HPD_CHUNK_BYTES = 8 << 30


def level_resolutions(n_min: int, n_max: int, num_levels: int) -> np.ndarray:
    """reference models.py:305-317 — identical float64 numpy expression (4096 -> 4095 at the last level)."""
    b = np.exp((np.log(n_max) - np.log(n_min)) / (num_levels - 1))
    return np.array([np.floor(n_min * b ** l) for l in range(num_levels)]).astype(np.int32)


class DifferentiableTopk(torch.autograd.Function):
    """reference models.py:5-42: torch.topk forward; backward scatters the K gradients into a zero tensor.
    Kept for API compatibility on small (rows, T) inputs; the hot path folds this into softmax backward."""

    @staticmethod
    def forward(ctx, input: torch.Tensor, k: int, dim: int):
        if dim not in (-1, input.dim() - 1):
            raise ValueError("DifferentiableTopk: only the last dimension is supported")
        x = input.reshape(-1, input.shape[-1]).contiguous()
        U, T = x.shape
        vals = torch.empty((U, k), dtype=torch.float32, device=x.device)
        idx = torch.empty((U, k), dtype=torch.int32, device=x.device)
        ops.call("gngf_topk", ops.ptr(x, torch.float32, "input"), ops.ptr(vals), ops.ptr(idx), U, T, k, ops.stream_ptr())
        idx64 = idx.to(torch.int64).reshape(*input.shape[:-1], k)
        ctx.save_for_backward(idx64)
        ctx.in_shape = input.shape
        ctx.mark_non_differentiable(idx64)
        return vals.reshape(*input.shape[:-1], k), idx64

    @staticmethod
    def backward(ctx, grad_values, grad_indices):
        (idx,) = ctx.saved_tensors
        g = torch.zeros(ctx.in_shape, dtype=grad_values.dtype, device=grad_values.device)
        g.scatter_(-1, idx, grad_values)
        return g, None, None


class HashProbDistribution(nn.Module):
    """reference models.py:45-123.  Same parameters / state-dict keys (module_list.{i}.0.{weight,bias})."""

    def __init__(self, hidden_layers_widths: list, in_features: int = 2, out_features: int = 2 ** 14, k: int = 1,
                 topk_dim: int = -1, should_log: bool = False):
        super().__init__()
        self.out_features = out_features
        self._k = k
        self._topk_dim = topk_dim
        self._should_log = should_log
        widths = [in_features, *hidden_layers_widths, out_features]
        self.module_list = nn.ModuleList([
            nn.Sequential(nn.Linear(widths[i], widths[i + 1]),
                          nn.ReLU() if i < len(widths) - 2 else nn.Softmax(dim=-1))
            for i in range(len(widths) - 1)
        ])

    def flat_params(self):
        out = []
        for seq in self.module_list:
            out += [seq[0].weight, seq[0].bias]
        return out

    def forward(self, x: torch.Tensor) -> Tuple:
        """x (..., in_features) -> (probs (..., T), topk_probs (..., K), topk_indices (..., K) int64) — the
        per-row formulation of the reference (dense distribution), on the HIP GEMM / softmax / top-K kernels."""
        lead = x.shape[:-1]
        rows = x.reshape(-1, x.shape[-1])
        n = len(self.module_list)
        acts = tuple([ops.ACT_RELU] * (n - 1) + [ops.ACT_NONE])
        logits = ops.MlpFunction.apply(rows, acts, *self.flat_params())
        probs, tp, ti = ops.SoftmaxTopkFunction.apply(logits, self._k)
        T = probs.shape[-1]
        return probs.reshape(*lead, T), tp.reshape(*lead, self._k), ti.to(torch.int64).reshape(*lead, self._k)


class MultiResHashEncoding(nn.Module):
    """reference models.py:126-236.  L tables (T,F) addressable as `_hash_tables[l].weight`
    (state-dict keys `_hash_tables.{l}.weight`), stored as views of ONE contiguous (L,T,F) buffer so that the
    kernels see a single base pointer."""

    def __init__(self, hash_table_size: int, num_levels: int, feature_dim: int = 2, topk_k: int = 4,
                 should_log: bool = False, table_dtype: torch.dtype = torch.float32) -> None:
        """table_dtype (extension, not in the reference): torch.float16 stores the level tables in half precision
        (BASELINE.json config 5); interpolation arithmetic and gradient accumulation stay fp32, the gradient handed to
        the fp16 parameters is rounded once at the end (use loss scaling as in any fp16 training)."""
        super().__init__()
        self._hash_table_size = hash_table_size
        self._num_levels = num_levels
        self._feature_dim = feature_dim
        self._topk_k = topk_k
        self._should_log = should_log
        base = torch.empty((num_levels, hash_table_size, feature_dim), dtype=torch.float32, device=device)
        base.uniform_(-10.0 ** (-4), 10.0 ** (-4))                       # models.py:169
        base = base.to(table_dtype)
        self._hash_tables = nn.ModuleList([nn.Embedding(hash_table_size, feature_dim, _weight=base[l])
                                           for l in range(num_levels)])
        self._base = base
        self._grad_base = None          # (L,T,F) gradient buffer behind the L `.grad` views (set by backward)
        self._grad_base_fp32 = None     # fp16 storage with the fp32 hand-over: the fp32 accumulation buffer itself
        self.grad_fp32_handover = None  # fp16 storage: hand the fp32 gradient buffer over as param.grad_fp32 (None: ops.FP16_TABLE_GRAD_FP32)

    def _apply_init(self, init_func, *args):
        for i in range(self._num_levels):
            init_func(self._hash_tables[i].weight, *args)

    def zero_grad(self, set_to_none: bool = True) -> None:
        """nn.Module.zero_grad, plus the fp32 hand-over gradients of fp16 tables (`param.grad_fp32`, ops._grad_out): a loop that
        clears gradients through the module (net.zero_grad()) must not let them pile up across steps."""
        for m in self._hash_tables:
            if getattr(m.weight, "grad_fp32", None) is not None:
                m.weight.grad_fp32 = None
        self._grad_base_fp32 = None
        super().zero_grad(set_to_none=set_to_none)

    def packed_tables(self) -> torch.Tensor:
        """The contiguous (L,T,F) storage behind the L parameters; re-packs (keeping the Parameter objects,
        hence optimizer state) if something — .to(), .cuda(), load with assign — split the storage."""
        ws = [m.weight for m in self._hash_tables]
        L, T, F = self._num_levels, self._hash_table_size, self._feature_dim
        base = self._base
        stride = T * F * base.element_size()
        ok = (base.device == ws[0].device and base.is_contiguous()
              and all(w.data_ptr() == base.data_ptr() + l * stride and w.is_contiguous() for l, w in enumerate(ws)))
        if not ok:
            base = torch.stack([w.detach() for w in ws]).contiguous()
            for l, w in enumerate(ws):
                w.data = base[l]
            self._base = base
        return base

    def forward(self, hashed_indices: torch.Tensor, hashed_probs_topk: torch.Tensor, should_calc_counts: bool = False):
        """hashed_indices (P,L,4) [hash] or (P,L,4,K) [GNGF] int64; hashed_probs_topk (P,L,4,K) | None -> (P,F,L,4)."""
        tables, sink = ops.table_view(self)
        if should_use_hash_function:
            if hashed_indices.dim() != 3:
                raise ValueError("hash mode expects indices of shape (P, L, 4)")
            return ops.MrheFunction.apply(tables, hashed_indices, None, 0, sink)
        if hashed_indices.dim() != 4:
            raise ValueError("GNGF mode expects indices of shape (P, L, 4, K)")
        return ops.MrheFunction.apply(tables, hashed_indices, hashed_probs_topk, ops.BLEND_CODES[should_softmax_topk_features], sink)


class VertexDistribution:
    """Compact stand-in for the reference's dense (P,L,4,T) `probs` when it would not fit in memory:
    per-vertex rows are never expanded; `pbar` (L,T) is the batch-mean distribution the loss needs
    (reference utils.py:138,159).  `shape` mirrors the dense tensor's shape."""

    def __init__(self, shape, pbar, topk_probs):
        self.shape = torch.Size(shape)
        self.pbar = pbar
        self.topk_probs = topk_probs

    def clone(self):
        return self


class GeneralNeuralGaugeFields(nn.Module):
    """reference models.py:239-655 — same constructor, forward(x, batch_percentage, should_calc_counts) and
    4-tuple return.  Extra (keyword-only, optional) attributes control what the reference-shaped outputs cost:
        return_indices   (default True)  materialise the (P,L,4[,K]) int64 index tensor of the return contract
        dense_probs      (default 'auto') materialise the (P,L,4,T) distribution when it fits DENSE_OUTPUT_LIMIT_BYTES
        coord_bounds     (default None)  (max_row, max_col) of the coordinates if known: skips one device sync/step
        compute_pbar     (default True)  when probs is returned compact, also reduce the batch-mean distribution
    """

    def __init__(self, input_dim, hash_table_size: int, num_levels: int, n_min: int, n_max: int,
                 MLP_hidden_layers_widths: list, HPD_hidden_layers_widths: list, HPD_out_features: int = 1,
                 feature_dim: int = 2, topk_k: int = 4, should_keep_topk_only: bool = False, should_bw: bool = False,
                 should_log: bool = False, HPD_weights_path: str = None, encoding_weights_path: str = None,
                 table_dtype: torch.dtype = torch.float32):
        super().__init__()
        if input_dim != 2:
            raise ValueError("the gfx950 path implements the reference's 2-D image case (input_dim == 2)")
        self._hash_table_size = hash_table_size
        self._num_levels = num_levels
        self._n_min = n_min
        self._n_max = n_max
        self._feature_dim = feature_dim
        self._input_dim = input_dim
        self._topk_k = topk_k
        self._should_log = should_log
        self._should_keep_topk_only = should_keep_topk_only

        b = np.exp((np.log(n_max) - np.log(n_min)) / (num_levels - 1))
        if b > 2 or b <= 1:                                                # models.py:306-309
            print(f"The between level scale is recommended to be <= 2 and needs to be > 1 but was {b:.4f}.")
        self._n_ls_host = [int(n) for n in level_resolutions(n_min, n_max, num_levels)]
        self._n_ls = torch.from_numpy(level_resolutions(n_min, n_max, num_levels)).reshape(1, 1, -1, 1).to(device).int()
        cube = np.array([[0, 1, 0, 1], [0, 0, 1, 1]], dtype=np.int32)       # models.py:322-331
        self._voxels_helper_hypercube = torch.from_numpy(cube).unsqueeze(0).unsqueeze(2).to(device).int()

        self._batch_norm = nn.BatchNorm1d(input_dim)
        if should_use_hash_function:
            self._prime_numbers = nn.Parameter(torch.from_numpy(np.array([1, 2654435761, 805459861])).to(device), False)
        else:
            self.HPD = HashProbDistribution(HPD_hidden_layers_widths, in_features=input_dim,
                                            out_features=HPD_out_features, k=topk_k, topk_dim=-1)
            if HPD_weights_path is not None:                               # models.py:364-371
                self.HPD.load_state_dict(torch.load(HPD_weights_path))
                for _name, param in self.HPD.named_parameters():
                    param.requires_grad = False
        self.encoding = MultiResHashEncoding(hash_table_size, num_levels, feature_dim, topk_k, should_log, table_dtype)
        widths = [num_levels * feature_dim, *MLP_hidden_layers_widths, (3 if not should_bw else 1)]
        self._MLP_hidden_layers_widths = widths
        self.mlp = nn.ModuleList([
            nn.Sequential(nn.Linear(widths[i], widths[i + 1]),
                          (nn.LeakyReLU() if should_leaky_relu else nn.ReLU()) if i < len(widths) - 2 else nn.Sigmoid())
            for i in range(len(widths) - 1)
        ])
        self._hash_mode = bool(should_use_hash_function)
        self._leaky = bool(should_leaky_relu)
        self.return_indices = True
        self.dense_probs = "auto"
        self.coord_bounds = None
        self.compute_pbar = True           # batch-mean distribution for the loss when probs is returned compact
        self._frozen_table = None          # cached per-vertex (idx, w, q) when the HPD is frozen
        self._fused_mse_target = None      # see fused_mse()
        self._fused_mse_gloss = None
        self.dp = ops.DataParallel()       # data-parallel state of THIS model (parallel.enable_vertex_grid_exchange(net, ...))
        self.dp.level_params = tuple(m.weight for m in self.encoding._hash_tables)
        self.hpd_stats = {}                # shape of the last chunked HPD evaluation (rows, chunks, chunks kept)
        self.loss_value_aside = False      # the caller joins the fused loss value itself (train.GraphedStep)
        self._last_link = None             # ops.StepLink of the most recent forward pass
        self.track_collisions = False      # every forward pass also marks the table slots its batch uses (start_collision_tracking)
        self._slot_maps = None             # (slot bit maps (K|1, L, T bits), touched-vertex workspace)
        self.to(device)

    # ------------------------------------------------------------------ helpers
    def _n_ls_flat(self, dev):
        if self._n_ls.device != dev:
            self._n_ls = self._n_ls.to(dev)
        return self._n_ls.reshape(-1)

    def _decoder_params(self):
        out = []
        for seq in self.mlp:
            out += [seq[0].weight, seq[0].bias]
        return out

    def _decode(self, enc, link=None):
        n = len(self.mlp)
        hidden = ops.ACT_LEAKY if self._leaky else ops.ACT_RELU
        return ops.decoder_apply(enc, tuple([hidden] * (n - 1) + [ops.ACT_SIGMOID]), self._decoder_params(),
                                 mse_target=self._fused_mse_target, mse_gloss=self._fused_mse_gloss, link=link)

    def join_loss_value(self):
        """Launches the fused loss value of the last forward pass if it was deferred (loss_value_aside) and found no launch
        to ride on."""
        if self._last_link is not None:
            self._last_link.join_loss_value()

    def fused_mse(self, target, gloss=None):
        """Context manager: forward passes inside it also evaluate torch.nn.MSELoss()(rgb, target) in the decoder kernels and
        attach the value to the returned rgb; train.Loss (ops.mse_loss) picks it up when it is given that rgb and this very
        `target` tensor, instead of launching the loss kernels.  Without `gloss` the results are those of the separate
        kernels (gradients bit for bit).  The training loops of train.py use it; it is never required.
        gloss (number): a PROMISE that the gradient arriving at that loss value in backward() will be exactly this (the loss
        weight l_mse when the total is l_mse * mse + ... and backward() is seeded with 1): at 32 encoder features the decoder
        then runs its forward and backward in ONE launch (ops.DECODER_TRAIN_FUSION); its forward layers run on the bf16 pipe
        with an exact three-way split, so results equal the two-kernel path to fp32 ROUNDING (not bit for bit; both are
        tested against float64).  The promise is checked on the device when the gradient arrives: if it differs (relative
        1e-6), every gradient of that backward pass is NaN — loud, never silently computed for the wrong value."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev = (self._fused_mse_target, self._fused_mse_gloss)
            self._fused_mse_target, self._fused_mse_gloss = target, gloss
            try:
                yield self
            finally:
                self._fused_mse_target, self._fused_mse_gloss = prev
        return scope()

    def _vertex_extent(self, x):
        """(vstride, NV) of the dense per-vertex table covering every corner of every level for this batch."""
        if self.coord_bounds is not None:
            mx, my = self.coord_bounds
        else:
            if x.shape[0] > 0:
                b = torch.cat([x.amax(0), -x.amin(0)])
            else:
                b = torch.zeros((4,), dtype=x.dtype, device=x.device)       # empty batch
            if self.dp.max_ is not None:
                self.dp.max_(b)                                             # every rank must build the same vertex table
            mx, my, nx, ny = b.tolist()                                     # one host sync
            if nx > 0 or ny > 0:
                raise ValueError("GNGF indexing builds its per-vertex table over [0, max] x [0, max]: negative coordinates "
                                 f"(min {-nx:g}, {-ny:g}) are outside it")
        gx_hi = int(np.floor(np.float32(mx) * np.float32(self._n_max))) + 1
        gy_hi = int(np.floor(np.float32(my) * np.float32(self._n_max))) + 1
        return gx_hi + 1, (gx_hi + 1) * (gy_hi + 1)

    def hpd_is_frozen(self):
        return all(not p.requires_grad for p in self.HPD.parameters())

    @torch.no_grad()
    def _frozen_vertex_table(self, blend_code):
        params = self.HPD.flat_params()
        key = (tuple((p.data_ptr(), p._version) for p in params), blend_code, self._topk_k)
        if self._frozen_table is None or self._frozen_table[0] != key:
            vstride = self._n_max + 2
            NV = vstride * vstride
            tv, ti, _, _ = ops.HpdVertexFunction.apply(NV, vstride, self._topk_k, None, False, HPD_CHUNK_BYTES, None, *params)
            w = ops.BlendFunction.apply(tv, blend_code)
            self._frozen_table = (key, tv, ti, w, vstride, NV, ops.slot_order(ti, self._n_ls_host, vstride))
        return self._frozen_table[1:]

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, batch_percentage: float = 1.0, should_calc_counts: bool = False):
        if should_batchnorm_data:
            x = self._batch_norm(x)
        x = x.contiguous()
        dev = x.device
        n_ls = self._n_ls_flat(dev)
        tables, sink = ops.table_view(self.encoding)
        P, L, T, K = x.shape[0], self._num_levels, self._hash_table_size, self._topk_k
        # (the fused training decoder — 32 encoder features, a promised loss gradient — clears the table-gradient buffer itself)
        train_fused = (self._fused_mse_gloss is not None and self._fused_mse_target is not None and ops.DECODER_TRAIN_FUSION
                       and ops.DECODER_REDUCE_RIDES and L * self._feature_dim == 32 and P > 0
                       and ops.decoder_fused_ok((ops.ACT_RELU, ops.ACT_RELU, ops.ACT_SIGMOID), self._decoder_params()))
        # (no training kernel — e.g. 64 encoder features — but the fused two-kernel decoder with the loss inside: its backward
        # kernel clears the buffer)
        bwd_clears = (not train_fused and ops.DECODER_BWD_CLEARS and self._fused_mse_target is not None and P > 0
                      and torch.is_grad_enabled()
                      and ops.decoder_fused_ok((ops.ACT_RELU, ops.ACT_RELU, ops.ACT_SIGMOID), self._decoder_params()))
        link = self._last_link = ops.StepLink(loss_value_aside=self.loss_value_aside, defer_zero=train_fused or bwd_clears,
                                                  zero_hidden=train_fused)
        dp = self.dp

        if self._hash_mode:
            # batch-normalised coordinates (models.py:394-397) leave [0,1]^2: the direct form hashes any integer vertex
            enc = ops.encode_apply(x, n_ls, self._n_ls_host, tables, None, None, 0, path=("direct" if should_batchnorm_data else None),
                                   dp=dp, link=link, sink=sink)
            rgb = self._decode(enc, link)
            if self.track_collisions and not should_batchnorm_data:
                vs = self._n_max + 2
                self._mark_batch_slots(x.detach(), n_ls, None, vs, vs * vs)
            idx = ops.hash_indices(x.detach(), n_ls, T) if (self.return_indices or should_calc_counts) else None
            counts = self._calc_counts_per_level(idx, x.detach(), n_ls) if should_calc_counts else []
            return rgb, None, idx, counts

        keep_topk = self._should_keep_topk_only
        if should_batchnorm_data:
            return self._forward_per_instance(x, n_ls, link, should_calc_counts)
        dense_bytes = P * L * 4 * T * 4
        want_dense = (not keep_topk) and (self.dense_probs is True or (self.dense_probs == "auto" and dense_bytes <= DENSE_OUTPUT_LIMIT_BYTES))
        extent = None
        if want_dense and self.dense_probs == "auto":
            # the dense tensor is gathered from per-vertex rows (NV, T): a handful of pixels spread over a fine grid at a
            # large T has a small (P,L,4,T) but a huge bounding rectangle of vertices
            extent = self._vertex_extent(x.detach())
            want_dense = extent[1] * T * 4 <= DENSE_OUTPUT_LIMIT_BYTES
        need_pbar = (not keep_topk) and (not want_dense) and self.compute_pbar
        blend_code = ops.BLEND_CODES[should_softmax_topk_features]
        if self.hpd_is_frozen() and not want_dense and not need_pbar:
            # frozen HPD (-hwp mode, models.py:364-371): the per-vertex table is a pure function of the frozen
            # weights, so it is rebuilt only when they change (version counters), over the whole [0,1]^2 domain.
            tv, ti, w, vstride, NV, order = self._frozen_vertex_table(blend_code)
            pbar = probs_u = None
        else:
            vstride, NV = extent if extent is not None else self._vertex_extent(x.detach())
            mw = ops.vertex_multiplicity_weights(x.detach(), n_ls, vstride, NV) if need_pbar else None
            tv, ti, pbar, probs_u = ops.HpdVertexFunction.apply(NV, vstride, K, mw, want_dense, HPD_CHUNK_BYTES,
                                                                ops.HpdAux(dp.mean, self.hpd_stats), *self.HPD.flat_params())
            w = ops.BlendFunction.apply(tv, blend_code)
            order = None
        enc = ops.encode_apply(x, n_ls, self._n_ls_host, tables, ti, w, vstride, order=order, dp=dp, link=link, sink=sink)
        rgb = self._decode(enc, link)
        if self.track_collisions:
            self._mark_batch_slots(x.detach(), n_ls, ti, vstride, NV)

        need_vid = want_dense or keep_topk or should_calc_counts
        need_idx = self.return_indices or should_calc_counts
        vid = idx64 = None
        if need_vid or need_idx:
            vid, idx64, _ = ops.expand_vertex_table(x.detach(), n_ls, vstride, NV, src_idx=ti if need_idx else None,
                                                    want_vid=need_vid)
        if keep_topk:
            to_return_probs = tv[vid]                                       # (P,L,4,K), differentiable gather
        elif want_dense:
            to_return_probs = probs_u[vid]                                  # (P,L,4,T), differentiable gather
        else:
            to_return_probs = VertexDistribution((P, L, 4, T), pbar, tv)
        counts = self._calc_counts_per_level(idx64[..., 0], x.detach(), n_ls) if should_calc_counts else []
        return rgb, to_return_probs, idx64, counts

    def _forward_per_instance(self, x, n_ls, link, should_calc_counts):
        """GNGF indexing on coordinates outside [0,1]^2 (should_batchnorm_data, models.py:394-397: BatchNorm1d output is
        centred on 0, so grid vertices are negative and the dense per-vertex table of the fast path does not apply): the
        reference's own per-instance formulation on the module-boundary kernels — HPD once per DISTINCT vertex of the batch
        (rows found with torch.unique; identical to evaluating every instance), MultiResHashEncoding.forward
        (gngf_mrhe_fwd), _bilinear_interpolate (gngf_bilinear_fwd), decoder.  Dense (U,T) distributions: small T only."""
        P, L, T, K = x.shape[0], self._num_levels, self._hash_table_size, self._topk_k
        xd = x.detach()
        scaled = xd[:, :, None] * n_ls.to(torch.float32)[None, None, :]                         # (P,2,L)   models.py:492-495
        cube = self._voxels_helper_hypercube.to(x.device).to(torch.float32).reshape(1, 2, 1, 4)
        grid = torch.floor(scaled)[..., None] + cube                                             # (P,2,L,4)
        verts = grid.permute(0, 2, 3, 1).reshape(-1, 2)                                          # "p xy l v -> (p l v) xy"
        uniq, inv = torch.unique(verts, dim=0, return_inverse=True)
        if uniq.shape[0] * T * 4 > DENSE_OUTPUT_LIMIT_BYTES:
            raise ValueError(f"per-instance GNGF path: {uniq.shape[0]} distinct vertices x T = {T} exceeds DENSE_OUTPUT_LIMIT_BYTES")
        probs_u, tp_u, ti_u = self.HPD(uniq.contiguous())
        tp = tp_u[inv].reshape(P, L, 4, K)
        ti = ti_u[inv].reshape(P, L, 4, K)
        feats = self.encoding(ti, tp)
        enc = ops.BilinearFunction.apply(xd, n_ls, feats)
        rgb = self._decode(enc, link)
        probs = tp if self._should_keep_topk_only else probs_u[inv].reshape(P, L, 4, T)
        counts = self._calc_counts_per_level(ti[..., 0], xd, n_ls) if should_calc_counts else []
        return rgb, probs, ti, counts

    # ------------------------------------------------------------------ diagnostics (no-grad statistics, not kernels)
    @torch.no_grad()
    def _calc_counts_per_level(self, hash_idx, x, n_ls):
        """reference models.py:530-566, restated literally (a no-grad statistic with numpy round trips in the reference
        too): per level, the pixels with distinct corner sets are found with np.unique(axis=0, return_index=True) and a
        Counter is taken of `rearrange(hash, "p l v -> l (p v)")[level][unique_indices]` — including the reference's
        indexing of the flattened (p v) axis with PIXEL indices."""
        from collections import Counter
        L = self._num_levels
        P = x.shape[0]
        scaled = x[:, :, None] * n_ls.to(torch.float32)[None, None, :]                       # (P,2,L)  models.py:492-495
        cube = self._voxels_helper_hypercube.to(x.device).to(torch.float32).reshape(1, 2, 1, 4)
        grid = torch.floor(scaled)[..., None] + cube                                           # (P,2,L,4)
        rearranged = grid.permute(2, 0, 3, 1).reshape(L, P, 8).cpu().numpy()                   # "p xy l v -> l p (v xy)"
        per_level = hash_idx.permute(1, 0, 2).reshape(L, P * 4).cpu().numpy()
        out = []
        for l in range(L):
            _, first = np.unique(rearranged[l], axis=0, return_index=True)
            out.append(dict(Counter(per_level[l][first].tolist())))
        return out

    @staticmethod
    def _distinct_slot_counts(indices, L, T):
        """(K,L) int32 on the device: distinct slots per (top-K rank, level) of a (P,L,4[,K]) int64 index tensor — one
        bit-map pass of csrc/stats.hip instead of one torch.unique (sort + host synchronisation) per level and rank."""
        from ._lib import call, ptr, query, stream_ptr
        idx = indices.contiguous()
        K = idx.shape[3] if idx.dim() == 4 else 1
        bitmap = torch.empty((query("gngf_slot_bitmap_words", L, K, T),), dtype=torch.int32, device=idx.device)
        counts = torch.empty((K, L), dtype=torch.int32, device=idx.device)
        call("gngf_distinct_slot_counts", ptr(idx, torch.int64, "indices"), idx.shape[0], L, idx.shape[2], K, T, ptr(bitmap),
             ptr(counts), stream_ptr())
        return counts

    def _level_vertex_counts(self):
        n = self._n_ls.reshape(-1).cpu().numpy().astype(np.int64)
        return 4 + (n + 1 - 2) * 4 + (n + 1 - 2) ** 2                        # = (N_l + 1)^2   (models.py:577-582)

    def _collisions_from_used(self, used, dev):
        """used: (K|1, L) int32 device tensor of distinct slots per (top-K rank, level) -> the reference's (collisions,
        min_possible_collisions) of models.py:568-619."""
        nverts = self._level_vertex_counts()
        if self._hash_mode:
            coll = torch.from_numpy(nverts - used[0].cpu().numpy().astype(np.int64))
        else:
            coll = torch.from_numpy(nverts.astype(np.float32)).to(dev)[None, :] - used.to(torch.float32)     # (K,L)
            coll = coll.mean(0)
            coll[coll < 0] = 0
        min_possible = torch.tensor(nverts - self._hash_table_size).to(dev)
        min_possible[min_possible < 0] = 0
        return coll, min_possible

    @torch.no_grad()
    def calc_hash_collisions(self, indices: torch.Tensor):
        """reference models.py:568-619: (#vertices of the level) - (#distinct slots used), per level."""
        nverts = self._level_vertex_counts()
        L = self._num_levels
        on_gpu = indices.is_cuda and indices.dtype == torch.int64 and indices.shape[1] == L
        if on_gpu:
            return self._collisions_from_used(self._distinct_slot_counts(indices, L, self._hash_table_size), indices.device)
        if self._hash_mode:
            per_level = indices.permute(1, 0, 2).reshape(L, -1)
            coll = torch.tensor([int(nverts[i]) - int(torch.unique(per_level[i]).numel()) for i in range(L)])
        else:
            Kk = indices.shape[-1]
            coll = torch.empty((Kk, L), device=indices.device)
            for k in range(Kk):
                per_level = indices[..., k].permute(1, 0, 2).reshape(L, -1)
                coll[k] = torch.tensor([float(int(nverts[i]) - int(torch.unique(per_level[i]).numel())) for i in range(L)])
            coll = coll.mean(0)
            coll[coll < 0] = 0
        min_possible = torch.tensor(nverts - self._hash_table_size).to(indices.device)
        min_possible[min_possible < 0] = 0
        return coll, min_possible

    # ---- the same statistic without the index tensor: the forward passes mark the slots their batches use
    def start_collision_tracking(self):
        """From now on every forward pass marks, into per-(rank, level) T-bit maps on the device, the table slots its batch uses
        (csrc/stats.hip: gngf_mark_batch_slots — both index sources depend on (level, vertex) only, so these are the slots of
        the batch's TOUCHED vertices: ~0.1 ms per step instead of writing and re-reading the 2 GiB (P,L,4,K) int64 tensor).
        tracked_hash_collisions() then returns what calc_hash_collisions(indices of every batch since) would.  Call it once
        per epoch; the maps accumulate over the batches in between (a trainable HPD changes the per-vertex table from batch to
        batch: each batch marks with its own, as the reference's per-batch indices do, functions.py:211-216, 327)."""
        from ._lib import query
        L, T = self._num_levels, self._hash_table_size
        K = 1 if self._hash_mode else self._topk_k
        dev = self.encoding._hash_tables[0].weight.device
        vs = self._n_max + 2
        words = query("gngf_slot_bitmap_words", L, K, T)
        if self._slot_maps is None or self._slot_maps[0].numel() != words or self._slot_maps[0].device != dev:
            self._slot_maps = (torch.zeros((words,), dtype=torch.int32, device=dev),
                               torch.zeros((L * ((vs * vs + 31) // 32),), dtype=torch.int32, device=dev))
        else:
            self._slot_maps[0].zero_()
        self.track_collisions = True

    def stop_collision_tracking(self):
        self.track_collisions = False

    @torch.no_grad()
    def _mark_batch_slots(self, x, n_ls, vert_idx, vstride, NV):
        from ._lib import call, ptr, stream_ptr
        bitmap, touched = self._slot_maps
        if int(NV) > (self._n_max + 2) ** 2:
            raise ValueError("collision tracking: the per-vertex table exceeds the [0, 1]^2 vertex grid")
        K = 1 if vert_idx is None else vert_idx.shape[1]
        call("gngf_mark_batch_slots", ptr(x, torch.float32, "x"), ptr(n_ls, torch.int32), x.shape[0], self._num_levels,
             ptr(vert_idx, torch.int32), K, self._hash_table_size, int(vstride), int(NV), ptr(touched), ptr(bitmap), stream_ptr())

    @torch.no_grad()
    def tracked_hash_collisions(self):
        """(collisions, min_possible_collisions) over every batch seen since start_collision_tracking() — equal to
        calc_hash_collisions(torch.cat(their index tensors)) (reference models.py:568-619, called at functions.py:327)."""
        from ._lib import call, ptr, stream_ptr
        if self._slot_maps is None:
            raise RuntimeError("tracked_hash_collisions(): call start_collision_tracking() before the epoch's forward passes")
        L, T = self._num_levels, self._hash_table_size
        K = 1 if self._hash_mode else self._topk_k
        bitmap = self._slot_maps[0]
        used = torch.empty((K, L), dtype=torch.int32, device=bitmap.device)
        call("gngf_count_slot_bits", ptr(bitmap), L, K, T, ptr(used), stream_ptr())
        return self._collisions_from_used(used, bitmap.device)
