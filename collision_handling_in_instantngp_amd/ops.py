"""torch.autograd.Function wrappers over the C-ABI (include/gngf.h).  Device memory + stream plumbing only;
all arithmetic happens in the HIP kernels.  CPU tensors raise (no fallback)."""
import ctypes as _ct
import dataclasses as _dc
import math as _math
import os
import sys as _sys
import types as _types

import torch

from . import _lib
from ._lib import call, ptr, query, stream_ptr

BLEND_CODES = {True: 0, None: 1, False: 2}   # should_softmax_topk_features -> GNGF_BLEND_*
MODE_HASH, MODE_VERTEX_TABLE = 0, 1

_f32, _i32, _i64 = torch.float32, torch.int32, torch.int64


# ------------------------------------------------------------------------------------------------ tuning: ONE frozen object
# Every measured threshold and A/B switch of the dispatch lives in ops.TUNING (round 5; they were ~35 module globals).  The object is
# frozen: a change replaces it as a whole — `ops.TUNING = dataclasses.replace(ops.TUNING, dg64=False)` — and, for the tests, tools
# and `bench.py --set NAME=VALUE` written against the old names, `ops.DG64 = False` / `ops.DG64` still work (the module forwards
# upper-case names of fields to the object).  The comments that explain each field stand where the mechanism is implemented
# ("# TUNING.<name> (default ...)" lines below).  Which kernel chain a step takes is decided ONCE per forward pass from this
# object, the plan and the model's state: StepConfig.choose().
@_dc.dataclass(frozen=True)
class Tuning:
    # ---- table gradient
    fp16_table_grad_fp32: bool = False        # fp16 tables: hand the fp32 accumulation buffer over (param.grad_fp32) instead of an fp16 .grad copy
    persistent_table_grad: bool = True        # a model whose loop opted in (dp.persist_ok) keeps ONE gradient buffer from step to step
    persistent_min_bytes: int = 1 << 28       # ... above this size (below, the training decoder's hidden dense clear is cheaper)
    # ---- direct levels (csrc/encode_direct.hip, encode_bucket.hip)
    bucketed_direct_bwd: bool = True          # counting sort by table slice + LDS sums instead of one memory-side atomic per contribution
    bucketed_min_pixels: int = 1 << 16
    bucket_image_bytes: int = 64 * 1024       # LDS image of one bucket (= table slice)
    bucketed_min_density: float = 0.5         # contributions per table row below which the atomics win ...
    bucketed_min_density_fresh: float = 0.2   # ... when the bucketed form WRITES the levels (no clear needed)
    direct_fwd_tile_order: bool = True        # the forward gather walks the pixels in the tiled form's binned order
    bucketed_tile_order: bool = True          # ... and so does the bucketed backward (whole-line item runs)
    # ---- HPD per distinct vertex (learning mode)
    hpd_z_cache_bytes: int = 216 << 30        # logits kept from forward to backward (the rest is recomputed)
    hpd_z_cache_reserve: int = 40 << 30       # device memory that must stay free beside them
    hpd_pipeline: bool = True                 # chunks software-pipelined over two streams (GEMMs beside streaming passes)
    hpd_pipeline_fwd: bool = True             # ... in the forward pass too (False: only the backward pass is pipelined)
    hpd_gemm_split_bf16: bool = True          # the three T-wide GEMMs on the exact three-way bf16 split
    hpd_gemm_kernel: int = 1                  # 1: values split once into bf16 planes in LDS (round 5) | 17: every wave splits what it reads
    hpd_bwd_two_planes: bool = True           # dW and dh (accumulated over >= 4096 terms) on two planes, three products
    hpd_bwd_fused: bool = True                # dz formed in the loaders of the dW / dh GEMMs (no apply pass, no d-logits matrix)
    hpd_epilogue_stats: bool = True           # row statistics in the logits GEMM's epilogue
    # ---- tiled form: plan (EncodePlan)
    encode_path: str = "auto"                 # "auto" | "direct" | "tiled" (tests force a path)
    tiled_chunk: object = None                # max pixels per work item (None: about two average tiles' worth)
    tiled_min_pixels: int = 1 << 14           # below this the binning is not worth it
    tiled_cells_per_pixel: float = 4.0        # a level is staged while N_l^2 <= this * P
    tiled_lds_limit: int = 48 * 1024          # forward image of the generic kernels
    tiled_tile_shift_bias: int = 0
    bin_blocks_max: int = 128
    bin_pixels_per_block: int = 8192
    # ---- tiled form: which kernels (StepConfig.choose reads these)
    two_launch_binning: bool = True           # count -> scatter with the scans riding inside (else four launches)
    fused_vertex_fwd: bool = True             # vertex stage forward inside the interleaved pixel stage's staging loop
    bin_pipeline: bool = True                 # an announced next batch is binned by riders of this step's pixel-stage launches
    dg64: bool = True                         # interleaved backward: 64-bit fixed-point vertex grid fed by integer atomics
    hash_vertex_fusion: bool = True           # hash source, single rank: no vertex-stage launch of its own
    hash_direct_scatter: bool = True          # ... and no vertex grid at all: the pixel stage adds to the hashed table rows (round 5)
    vertex_reads_dg64: bool = True            # slot-ordered vertex backward converts the fixed-point grid on the fly
    use_side_stream: bool = True              # False: helper-stream work in line on the current stream (measurement)
    # ---- decoder (csrc/decoder.hip)
    decoder_save_hidden: bool = True          # two-kernel path: hidden layers travel forward -> backward through HBM instead of being recomputed
    decoder_reduce_rides: bool = True         # the slab reduction rides on the encoder backward's launch
    decoder_train_fusion: bool = True         # forward + loss gradient + backward in ONE launch when the loss gradient is promised
    decoder_bwd_clears: bool = True           # 64-feature backward clears the encoder's gradient buffer on the way

TUNING = Tuning()
_TUNING_FIELDS = {f.name for f in _dc.fields(Tuning)}


class _OpsModule(_types.ModuleType):
    def __getattr__(self, name):                      # (only reached when the module has no such attribute)
        if name.isupper() and name.lower() in _TUNING_FIELDS:
            return getattr(self.TUNING, name.lower())
        raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name.isupper() and name.lower() in _TUNING_FIELDS:
            _types.ModuleType.__setattr__(self, "TUNING", _dc.replace(self.TUNING, **{name.lower(): value}))
        else:
            _types.ModuleType.__setattr__(self, name, value)


_sys.modules[__name__].__class__ = _OpsModule


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_f16 = torch.float16


def _tab(tables):
    """(device pointer, GNGF_FEAT_* code) of the level tables: fp32 (the reference) or fp16 storage."""
    if tables.dtype == _f32:
        return ptr(tables, _f32, "tables"), 0
    if tables.dtype == _f16:
        return ptr(tables, _f16, "tables"), 1
    raise TypeError(f"hash tables must be float32 or float16, got {tables.dtype}")


def _grad_buffer(tables):
    """table gradients are always accumulated in fp32 (fp16 atomics would round on every add)"""
    return torch.zeros(tables.shape, dtype=_f32, device=tables.device)


# fp16 level tables (BASELINE config 5): the table gradient is accumulated in an fp32 (L,T,F) buffer.  torch wants `.grad` in
# the parameter's type, so by default that buffer is cast to fp16 (6 GiB of traffic at T = 2^24, F = 4: a quarter of the
# step).  With FP16_TABLE_GRAD_FP32 the buffer itself is handed over instead: level l's parameter gets `grad_fp32` (a view of
# it) and no `.grad`; train.FusedAdam — which updates an fp32 master copy anyway — consumes it directly.
# TUNING.fp16_table_grad_fp32 (default False)     — default of MultiResHashEncoding.grad_fp32_handover (None there = this switch)


def table_view(owner):
    """(tables, sink) of an encoding module: its L level parameters as ONE (L,T,F) tensor inside the autograd graph
    (TableViewFunction), and — fp16 storage with the fp32 hand-over on — the (owner, level parameters) pair the backward hands
    the fp32 gradient buffer to.  The sink travels with the call (encode_apply(..., sink=) / MrheFunction) and ends up on that
    call's autograd ctx: several models in one process do not share it."""
    base = owner.packed_tables()
    ws = tuple(m.weight for m in owner._hash_tables)
    tables = TableViewFunction.apply(base, owner, *ws)
    handover = getattr(owner, "grad_fp32_handover", None)
    handover = TUNING.fp16_table_grad_fp32 if handover is None else handover
    sink = (owner, ws) if (base.dtype != _f32 and handover) else None
    return tables, sink


STEP_TRACE = None            # tests: a list that receives one (what, {facts}) record per dispatch decision of a backward pass
SEEN_STEP_CONFIGS = None     # tests: a set that receives StepConfig.chain() of every forward pass (tests/conftest.py records it per test)


def _trace(what, **facts):
    if STEP_TRACE is not None:
        STEP_TRACE.append((what, facts))


def _grad_out(dtables, tables, sink=None):
    if tables.dtype == _f32:
        return dtables
    _trace("grad_out", fp32_handover=sink is not None)
    if sink is None:
        return dtables.to(tables.dtype)
    owner, ws = sink
    owner._grad_base = None
    fresh = all(getattr(w, "grad_fp32", None) is None for w in ws)
    if fresh:
        owner._grad_base_fp32 = dtables
    for l, w in enumerate(ws):
        if w.requires_grad:
            w.grad = None
            if fresh:
                w.grad_fp32 = dtables[l]
            elif getattr(w, "grad_fp32", None) is None:
                w.grad_fp32 = dtables[l].clone()
            else:
                w.grad_fp32.add_(dtables[l])      # a second backward pass without zero_grad: accumulated, as .grad would be
    if not fresh:
        owner._grad_base_fp32 = None              # the levels' gradients are no longer views of one buffer
    return None


def hash_indices(xy, n_ls, T):
    """_scale_to_grid + _fast_hash (reference models.py:486-528) -> (P, L, 4) int64."""
    xy = _c(xy)
    P, L = xy.shape[0], n_ls.numel()
    idx = torch.empty((P, L, 4), dtype=_i64, device=xy.device)
    call("gngf_hash_indices", ptr(xy, _f32, "xy"), ptr(n_ls, _i32, "n_ls"), ptr(idx), P, L, T, stream_ptr())
    return idx


class MrheFunction(torch.autograd.Function):
    """MultiResHashEncoding.forward (reference models.py:173-229) at the module boundary."""

    @staticmethod
    def forward(ctx, tables, idx, probs, blend_code, sink=None):
        ctx.sink = sink
        tables, idx = _c(tables), _c(idx)
        L, T, F = tables.shape
        P = idx.shape[0]
        K = 0 if idx.dim() == 3 else idx.shape[-1]
        if K:
            probs = _c(probs)
        out = torch.empty((P, F, L, 4), dtype=_f32, device=tables.device)
        call("gngf_mrhe_fwd", *_tab(tables), ptr(idx, _i64, "indices"),
             ptr(probs if K else None, _f32, "probs"), ptr(out), P, L, F, T, K, blend_code, stream_ptr())
        ctx.save_for_backward(tables, idx, probs if K else None)
        ctx.cfg = (P, L, F, T, K, blend_code)
        return out

    @staticmethod
    def backward(ctx, gout):
        tables, idx, probs = ctx.saved_tensors
        P, L, F, T, K, blend_code = ctx.cfg
        gout = _c(gout)
        dtables = _grad_buffer(tables)
        dprobs = torch.empty_like(probs) if (K and ctx.needs_input_grad[2]) else None
        call("gngf_mrhe_bwd", *_tab(tables), ptr(idx), ptr(probs), ptr(gout, _f32, "grad"), ptr(dtables), ptr(dprobs),
             P, L, F, T, K, blend_code, stream_ptr())
        return _grad_out(dtables, tables, ctx.sink), None, dprobs, None, None


class BilinearFunction(torch.autograd.Function):
    """_bilinear_interpolate (reference models.py:621-655); coordinates carry no gradient (models.py:486)."""

    @staticmethod
    def forward(ctx, xy, n_ls, feats):
        xy, feats = _c(xy), _c(feats)
        P, F, L, _ = feats.shape
        enc = torch.empty((P, L * F), dtype=_f32, device=feats.device)
        call("gngf_bilinear_fwd", ptr(xy, _f32, "xy"), ptr(n_ls, _i32, "n_ls"), ptr(feats, _f32, "features"), ptr(enc),
             P, L, F, stream_ptr())
        ctx.save_for_backward(xy, n_ls)
        ctx.cfg = (P, L, F)
        return enc

    @staticmethod
    def backward(ctx, genc):
        xy, n_ls = ctx.saved_tensors
        P, L, F = ctx.cfg
        genc = _c(genc)
        dfeats = torch.empty((P, F, L, 4), dtype=_f32, device=genc.device)
        call("gngf_bilinear_bwd", ptr(xy), ptr(n_ls), ptr(genc, _f32, "grad"), ptr(dfeats), P, L, F, stream_ptr())
        return None, None, dfeats


# Backward of the direct levels, hash source: above BUCKETED_MIN_PIXELS the contributions are counting-sorted by table slice and summed
# in LDS images (csrc/encode_bucket.hip) instead of one memory-side atomic each (20.6 G row updates/s on this chip wherever the rows
# lie: tools/micro/atomic_window.cpp).  Bitwise reproducible; the terms are the same fp32 products, summed
# in 64-bit fixed point whose quantum is 2^-50 of the BUCKET's largest |term| (an absolute bound per bucket: a row 15 decimal orders
# below a neighbour in the same bucket is flushed — csrc/encode_bucket.hip's header), rounded to fp32 once.
# TUNING.bucketed_direct_bwd (default True)
# TUNING.bucketed_min_pixels (default 1 << 16)
# TUNING.bucket_image_bytes (default 64 * 1024)
# contributions per table row of a level below which the atomics win: every bucket costs its image's clear and its slice's
# write-out whatever it holds (measured, tools/perf_bucket.py: 1.0 per row at the 4096^2 shape: 407 -> 188 us; 0.25 per row at
# the 8192^2 one: 817 -> 822 us)
# TUNING.bucketed_min_density (default 0.5)
# ... when the bucketed form WRITES the levels (a fresh gradient buffer: those levels then need no clear — a quarter of the 4 GiB
# clear at the 8192^2 shape): step 2.83 -> 2.63 ms there although the kernels themselves only draw with the atomics
# TUNING.bucketed_min_density_fresh (default 0.2)
# TUNING.direct_fwd_tile_order (default True)   — ... and so does the direct levels' forward gather (gngf_encode_fwd(..., pixel_order))
# TUNING.bucketed_tile_order (default True)     — walk the pixels in the tiled form's binned order when a workspace exists (round 5; see bucket_pixel)


def bucketed_plan(P, F, T, nl, fresh=False):
    """(bucket_shift, buckets per level, pixel blocks, matrix ints, base ints, item bytes) or None — the library's own decision"""
    density = TUNING.bucketed_min_density_fresh if fresh else TUNING.bucketed_min_density
    if not TUNING.bucketed_direct_bwd or P < TUNING.bucketed_min_pixels or nl <= 0 or 4.0 * P < density * T:
        return None
    plan = (_ct.c_int64 * 6)()
    if query("gngf_encode_bwd_bucketed_plan", int(P), int(F), int(T), int(nl), int(TUNING.bucket_image_bytes), plan) != 1:
        return None
    return tuple(int(v) for v in plan)


def _direct_bwd(xy, tables, vert_idx, vert_w, n_ls, genc, dtables, dvw, P, L, F, T, K, mode, vstride, NV, l0, l1, fresh=False, order=None):
    """d tables of levels [l0, l1) in the direct form.  fresh: those levels of dtables hold NOTHING yet (not even zeros) — the
    bucketed form writes every row of them, the atomics form clears them first.  order: the batch's binned pixel records (P,4) of
    the tiled form's workspace, if there is one: the bucketed form walks the pixels tile by tile (whole-line item runs)."""
    plan = bucketed_plan(P, F, T, l1 - l0, fresh) if (mode == MODE_HASH and dtables.dtype == _f32) else None
    _trace("direct_bwd", bucketed=plan is not None, write=bool(fresh and plan is not None), levels=(l0, l1))
    if plan is None:
        if fresh:
            dtables[l0:l1].zero_()
        call("gngf_encode_bwd", ptr(xy), *_tab(tables), ptr(vert_idx), ptr(vert_w), ptr(n_ls), ptr(genc, _f32, "grad"),
             ptr(dtables), ptr(dvw), P, L, F, T, K, mode, vstride, NV, l0, l1, stream_ptr())
        return
    dev = dtables.device
    matrix = torch.empty((plan[3],), dtype=_i32, device=dev)
    base = torch.empty((plan[4],), dtype=_i32, device=dev)
    items = torch.empty((plan[5],), dtype=torch.uint8, device=dev)
    call("gngf_encode_bwd_bucketed", ptr(xy), ptr(n_ls), ptr(genc, _f32, "grad"), ptr(dtables), P, L, F, T, l0, l1,
         int(TUNING.bucket_image_bytes), 0 if fresh else 1, ptr(matrix), ptr(base), ptr(items),
         ptr(order if (TUNING.bucketed_tile_order and order is not None and order.shape[0] == P) else None, _f32, "pixel_order"), stream_ptr())


# Hash source, staged levels on the generic pixel-stage kernels (the big shapes): the table gradient lives in ONE buffer per model
# from step to step instead of a freshly cleared allocation per step.  What a backward pass can write is known without looking at
# the batch — the rows hash(gx, gy) of the staged levels' vertices (cleared by gngf_clear_hashed_rows: 74 MB of stores at the
# 8192^2 shape) and every row of the direct levels (WRITTEN by the bucketed backward) — so the dense clear (3 GiB there: 0.49 ms
# inside the decoder backward) goes.  The buffer is only taken when no level parameter's .grad / .grad_fp32 still lives in it
# (gradient accumulation over several backward passes, or a caller that keeps the gradients): then the pass gets a zeroed
# allocation of its own, as before.  `.grad` of consecutive steps aliases, as it does in torch with zero_grad(set_to_none=False).
# OPT-IN PER MODEL: a tensor a caller still holds (g = p.grad kept for logging, clipping, SAM, manual accumulation)
# would be overwritten by the next step, which torch never does — so the buffer is only used when the code that OWNS THE LOOP
# says so (`net.dp.persist_ok = True`: train.GraphedStep, train.train_epoch and bench.py do — they let go of every gradient at
# the top of each step and hand nothing older than the current step to their caller).  A bare `net(x); loss.backward()` gets an
# allocation per step, like torch.  This module-level switch turns the mechanism off for everybody (A/B measurements).
# TUNING.persistent_table_grad (default True)
# TUNING.persistent_min_bytes (default 1 << 28)     — below this a dense clear hidden inside the training decoder (StepLink.zero_hidden) costs next to nothing


def _persistent_peek(dp, tables):
    """the model's step-to-step gradient buffer if it is free right now (no level parameter's gradient lives in it), else None"""
    params = getattr(dp, "level_params", None)
    if not params or not getattr(dp, "persist_ok", False):
        return None
    buf = getattr(dp, "persist_grad", None)
    shape = tuple(tables.shape)
    if buf is None or tuple(buf.shape) != shape or buf.device != tables.device:
        if torch.cuda.is_current_stream_capturing():
            return None                                   # (a buffer born inside a capture belongs to that graph's pool)
        buf = dp.persist_grad = torch.zeros(shape, dtype=_f32, device=tables.device)
        dp.persist_gen = getattr(dp, "persist_gen", 0) + 1
    base = buf.untyped_storage().data_ptr()
    for w in params:
        for g in (w.grad, getattr(w, "grad_fp32", None)):
            if g is not None and g.is_cuda and g.untyped_storage().data_ptr() == base:
                return None                               # the previous gradient has not been let go of: it stays intact
    return buf


def _persistent_grad(dp, tables, plan, n_ls, cleared=None):
    """the buffer for a backward pass, staged levels' rows cleared — or None (in use / cannot be allocated here).
    cleared: the generation at which this pass's FORWARD had the vertex riders clear those rows (TiledWorkspace(clear_rows=)):
    if no backward pass has taken the buffer since, it is still clean and no clear is launched."""
    buf = _persistent_peek(dp, tables)
    if buf is None:
        return None
    if cleared is None or cleared != dp.persist_gen:
        L, T, F = tables.shape
        call("gngf_clear_hashed_rows", ptr(buf), ptr(n_ls), plan.Ls, F, T, plan.vtot, stream_ptr())
    dp.persist_gen += 1                                   # handed out: what forward passes cleared before this point is spent
    return buf


class EncodeDirectFunction(torch.autograd.Function):
    """Fused coords -> (P, L*F) encoder, direct form (include/gngf.h: gngf_encode_fwd / gngf_encode_bwd).

    tables (L,T,F).  Hash mode: vert_idx = vert_w = None.  Vertex-table mode: vert_idx (NV,K) int32,
    vert_w (NV,K) fp32 blend weights; the gradient returned for vert_w is dL/dw (before the blend's backward)."""

    @staticmethod
    def forward(ctx, xy, n_ls, tables, vert_idx, vert_w, vstride):
        xy, tables = _c(xy), _c(tables)
        L, T, F = tables.shape
        P = xy.shape[0]
        mode = MODE_HASH if vert_idx is None else MODE_VERTEX_TABLE
        K = 0 if vert_idx is None else vert_idx.shape[1]
        NV = 0 if vert_idx is None else vert_idx.shape[0]
        enc = torch.empty((P, L * F), dtype=_f32, device=tables.device)
        call("gngf_encode_fwd", ptr(xy, _f32, "xy"), *_tab(tables), ptr(vert_idx, _i32, "vert_idx"),
             ptr(vert_w, _f32, "vert_w"), ptr(n_ls, _i32, "n_ls"), ptr(enc), P, L, F, T, K, mode, vstride, NV, 0, L, ptr(None),
             stream_ptr())
        ctx.save_for_backward(xy, n_ls, tables, vert_idx, vert_w)
        ctx.cfg = (P, L, F, T, K, mode, vstride, NV)
        return enc

    @staticmethod
    def backward(ctx, genc):
        xy, n_ls, tables, vert_idx, vert_w = ctx.saved_tensors
        P, L, F, T, K, mode, vstride, NV = ctx.cfg
        genc = _c(genc)
        dtables = _grad_buffer(tables)
        dvw = torch.zeros_like(vert_w) if (vert_w is not None and ctx.needs_input_grad[4]) else None
        _direct_bwd(xy, tables, vert_idx, vert_w, n_ls, genc, dtables, dvw, P, L, F, T, K, mode, vstride, NV, 0, L)
        return None, None, _grad_out(dtables, tables), None, dvw, None


# ------------------------------------------------------------------------------------------------ dense layers
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3


def linear_fwd(x, w, b, act):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=_f32, device=x.device)
    call("gngf_linear_fwd", ptr(x, _f32, "x"), ptr(w, _f32, "weight"), ptr(b, _f32, "bias"), ptr(y), M, N, K, act, stream_ptr())
    return y


def linear_bwd_input(dy, y, w, act):
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty((M, K), dtype=_f32, device=dy.device)
    call("gngf_linear_bwd_input", ptr(dy, _f32), ptr(y if act else None, _f32), ptr(w, _f32), ptr(dx), M, N, K, act, stream_ptr())
    return dx


def linear_bwd_weight(dy, y, x, dw, db, act):
    """Accumulates into dw (N,K) and db (N,)."""
    M, N = dy.shape
    K = x.shape[1]
    call("gngf_linear_bwd_weight", ptr(dy, _f32), ptr(y if act else None, _f32), ptr(x, _f32), ptr(dw, _f32),
         ptr(db, _f32) if db is not None else ptr(None),
         M, N, K, act, stream_ptr())


def gemm_acc(a, b, c, M, N, Kc, ta, tb):
    call("gngf_gemm_acc", ptr(a, _f32), ptr(b, _f32), ptr(c, _f32), M, N, Kc, int(ta), int(tb), stream_ptr())


class MlpFunction(torch.autograd.Function):
    """Chain of nn.Linear + activation on the generic MFMA GEMM (reference models.py:382-392 / 80-88).
    apply(x, acts, w0, b0, w1, b1, ...) with acts a tuple of ACT_* codes, one per layer."""

    @staticmethod
    def forward(ctx, x, acts, *params):
        x = _c(x)
        hs = [x]
        for i, act in enumerate(acts):
            hs.append(linear_fwd(hs[-1], _c(params[2 * i]), _c(params[2 * i + 1]), act))
        ctx.acts = acts
        ctx.save_for_backward(*hs, *params)
        return hs[-1]

    @staticmethod
    def backward(ctx, gy):
        n = len(ctx.acts)
        hs, params = ctx.saved_tensors[: n + 1], ctx.saved_tensors[n + 1:]
        g = _c(gy)
        grads = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            w = _c(params[2 * i])
            dw, db = torch.zeros_like(w), torch.zeros_like(params[2 * i + 1])
            linear_bwd_weight(g, hs[i + 1], hs[i], dw, db, ctx.acts[i])
            grads[2 * i], grads[2 * i + 1] = dw, db
            if i > 0 or ctx.needs_input_grad[0]:
                g = linear_bwd_input(g, hs[i + 1], w, ctx.acts[i])
        return (g if ctx.needs_input_grad[0] else None, None, *grads)


# ------------------------------------------------------------------------------------------------ HPD per distinct vertex
def vertex_coords(u0, count, vstride, device):
    v = torch.empty((count, 2), dtype=_f32, device=device)
    call("gngf_vertex_coords", ptr(v), u0, count, vstride, stream_ptr())
    return v


def vertex_multiplicity_weights(xy, n_ls, vstride, NV):
    """mw (NV, L) = (#instances of vertex u at level l) / (4 P): weights of the batch-mean distribution."""
    xy = _c(xy)
    P, L = xy.shape[0], n_ls.numel()
    counts = torch.zeros((L, NV), dtype=_i32, device=xy.device)
    call("gngf_vertex_multiplicity", ptr(xy, _f32), ptr(n_ls, _i32), ptr(counts), P, L, vstride, NV, stream_ptr())
    mw = torch.empty((NV, L), dtype=_f32, device=xy.device)
    call("gngf_multiplicity_weights", ptr(counts), ptr(mw), NV, L, float(4 * P), stream_ptr())
    return mw


def expand_vertex_table(xy, n_ls, vstride, NV, src_idx=None, src_val=None, want_vid=False):
    """(P,L,4[,K]) reference-shaped tensors from per-vertex (NV,K) tables (reference return contract models.py:478-484)."""
    xy = _c(xy)
    P, L = xy.shape[0], n_ls.numel()
    K = src_idx.shape[1] if src_idx is not None else (src_val.shape[1] if src_val is not None else 0)
    dev = xy.device
    vid = torch.empty((P, L, 4), dtype=_i64, device=dev) if want_vid else None
    oi = torch.empty((P, L, 4, K), dtype=_i64, device=dev) if src_idx is not None else None
    ov = torch.empty((P, L, 4, K), dtype=_f32, device=dev) if src_val is not None else None
    call("gngf_expand_vertex_table", ptr(xy, _f32), ptr(n_ls, _i32), ptr(src_idx, _i32), ptr(src_val, _f32), ptr(vid), ptr(oi),
         ptr(ov), P, L, K, vstride, NV, stream_ptr())
    return vid, oi, ov


# Logits kept from the forward to the backward of the chunked HPD (see HpdVertexFunction.forward): budget, and the free
# device memory that must remain after keeping a chunk.
# TUNING.hpd_z_cache_bytes (default 216 << 30)
# TUNING.hpd_z_cache_reserve (default 40 << 30)


class HpdAux:
    """Per-call side inputs of HpdVertexFunction that are not tensors: `mean` (data parallel: callable averaging the batch-mean
    distribution over the ranks in place, or None) and `stats` (a dict — typically the model's `hpd_stats` — that receives the
    shape of the chunked evaluation: bench.py prices the step's GEMM FLOP with it)."""
    __slots__ = ("mean", "stats")

    def __init__(self, mean=None, stats=None):
        self.mean, self.stats = mean, stats

# Chunks are software-pipelined over two streams: the T-wide GEMMs (matrix-pipe bound) of one chunk run beside the streaming
# softmax / top-K / batch-mean passes (HBM bound) of its neighbour.  False: everything in line on the current stream.
# TUNING.hpd_pipeline (default True)


# The three T-wide products of the HPD's last layer (logits, dW, dh: 3 x 2 U 128 T FLOP per training step) run on the
# split-bf16 GEMM (csrc/linear.hip: every fp32 value split exactly into three bf16 terms, six cross products accumulated in
# fp32; measured error against float64 equal to or below the exact-fp32 MFMA kernel's, 1.2-1.3x its speed).  False: exact
# fp32 MFMA for these as well.
# TUNING.hpd_gemm_split_bf16 (default True)
# The logits GEMM leaves per-row (max, sum exp) partials per 64-column block in its epilogue and the row statistics + top-K come
# out of a merge over them (1/32 of the logits' bytes) instead of a pass over the logits: one of the step's ~8.4 passes over the
# (U, T) logit matrix less (learning mode is HBM-bound as a whole).  Needs the split-bf16 GEMM and whole 128 x 128 tiles; other
# chunks (the ragged last one) take the separate statistics pass as before.
# TUNING.hpd_epilogue_stats (default True)
# Round 5: TUNING.hpd_gemm_kernel = 1 — the split GEMMs split a value once on its way into LDS (bf16 planes; 17 = the round-4 kernel);
# TUNING.hpd_bwd_two_planes — dW and dh, which accumulate over >= 4096 terms, on two planes / three products (3 * 2^-18 |a b| per
# product; the logits keep the exact split); TUNING.hpd_bwd_fused — the backward forms the d-logits inside the dW / dh GEMMs' loaders
# from the logits (gngf_hpd_bwd_dot + gngf_hpd_bwd_prepare + gngf_hpd_bwd_fused) for every chunk of whole 128-row tiles; the ragged
# last chunk of a step, L > 16 or a hidden width other than 128 take the three separate entry points.


class _HpdBwdPlanes:
    """the small operands of gngf_hpd_bwd_fused split once per backward pass into bf16 planes (csrc/linear.hip: gngf_hpd_bwd_prepare)"""

    def __init__(self, h_all, mw, W, G, L, planes):
        NV, Hd = h_all.shape
        T = W.shape[0]
        dev = W.device
        self.NV, self.Hd, self.planes, self.L = NV, Hd, planes, L
        self.hp = torch.empty((planes, NV, Hd), dtype=torch.int16, device=dev)
        self.mwp = torch.empty((3, NV, 16), dtype=torch.int16, device=dev)
        self.Wp = torch.empty((planes, T, Hd), dtype=torch.int16, device=dev)
        self.Gtp = torch.empty((3, T, 16), dtype=torch.int16, device=dev)
        call("gngf_hpd_bwd_prepare", ptr(h_all, _f32), ptr(mw if L else None), NV, ptr(self.hp), ptr(self.mwp), ptr(W, _f32),
             ptr(G if L else None), T, ptr(self.Wp), ptr(self.Gtp), L, Hd, planes, stream_ptr())

    def fused(self, dz, rowstat, dots, g_tv, tv, ti, h, W, dW, db, dH, u0, n, T, K):
        call("gngf_hpd_bwd_fused", ptr(dz), ptr(rowstat), ptr(dots), ptr(g_tv if K else None), ptr(tv if K else None),
             ptr(ti if K else None), _lib._P(self.hp.data_ptr() + 2 * u0 * self.Hd), _lib._P(self.mwp.data_ptr() + 2 * u0 * 16), self.NV,
             ptr(self.Wp), ptr(self.Gtp), ptr(h), ptr(W), ptr(dW), ptr(db), ptr(dH), n, T, K, self.Hd, self.planes, stream_ptr())


class _split_gemm:
    """scope in which large aligned GEMMs of this process use the split-bf16 kernel (when HPD_GEMM_SPLIT_BF16)"""

    def __init__(self, accumulating=False):
        self.mode = 2 if (accumulating and TUNING.hpd_bwd_two_planes and TUNING.hpd_gemm_kernel == 1) else TUNING.hpd_gemm_kernel

    def __enter__(self):
        self.prev = query("gngf_set_gemm_split_bf16", self.mode) if TUNING.hpd_gemm_split_bf16 else None
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            query("gngf_set_gemm_split_bf16", self.prev)
        return False


class HpdVertexFunction(torch.autograd.Function):
    """HashProbDistribution (reference models.py:45-123) evaluated ONCE PER DISTINCT VERTEX u (vid = gy*vstride+gx,
    u in [0, NV)), in row chunks so that the (rows, T) distribution never exceeds `chunk_bytes`.

    apply(NV, vstride, K, mw, keep_probs, chunk_bytes, aux, *params) ->   (aux: HpdAux or None)
        topk_val (NV,K), topk_idx (NV,K) int32, pbar (L,T) | None, probs (NV,T) | None
    pbar_l = sum_u mw[u,l] * probs[u]  is the batch-mean distribution of utils.py:138,159 (mw = multiplicity/(4P)).
    Backward recomputes each chunk (hidden layers, logits, softmax) and folds the top-K gradient and the
    low-rank p-bar gradient into the softmax backward: the dense (P,L,4,T) zero-filled scatter of
    models.py:27-35 never exists."""

    @staticmethod
    def _hidden(verts, params, n_layers):
        hs = [verts]
        for i in range(n_layers - 1):
            hs.append(linear_fwd(hs[-1], params[2 * i], params[2 * i + 1], ACT_RELU))
        return hs

    @staticmethod
    def forward(ctx, NV, vstride, K, mw, keep_probs, chunk_bytes, aux, *params):
        params = tuple(_c(p) for p in params)
        n_layers = len(params) // 2
        W_last, b_last = params[-2], params[-1]
        T = W_last.shape[0]
        dev = W_last.device
        rows = int(max(64, min(NV, chunk_bytes // (4 * T))))
        tv = torch.empty((NV, K), dtype=_f32, device=dev)
        ti = torch.empty((NV, K), dtype=_i32, device=dev)
        rowstat = torch.empty((NV, 2), dtype=_f32, device=dev)
        probs = torch.empty((NV, T), dtype=_f32, device=dev) if keep_probs else None
        pbar = None
        if mw is not None:
            mw = _c(mw)
            L = mw.shape[1]
            pbar = torch.zeros((L, T), dtype=_f32, device=dev)
        pipelined = (TUNING.hpd_pipeline and TUNING.hpd_pipeline_fwd and not keep_probs and NV > rows
                     and not torch.cuda.is_current_stream_capturing())
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if pipelined else None
        pipelined = pipelined and side is not main and side.stream_id != main.stream_id
        # un-kept chunks alternate between two logits buffers when pipelined (the passes of chunk i read one while the GEMM
        # of chunk i + 1 fills the other)
        scratch = None if keep_probs else [torch.empty((min(rows, NV), T), dtype=_f32, device=dev) for _ in range(2 if pipelined else 1)]
        scratch_free = [None] * (len(scratch) if scratch else 0)      # event after the last pass that read the buffer
        n_scratch = 0
        # Checkpointing policy sized for 288 GB of HBM: the backward needs every chunk's logits again; the chunks that fit
        # HPD_Z_CACHE_BYTES (and leave HPD_Z_CACHE_RESERVE free on the device) keep theirs, the rest are recomputed.
        zcache, cached = {}, 0
        keep_z = (not keep_probs) and TUNING.hpd_z_cache_bytes > 0 and any(ctx.needs_input_grad[7:])
        budget = 0
        # The hidden layers (2 -> 32 -> 64 -> 128 at the reference's widths) are evaluated for ALL vertices in one go, outside the
        # chunk loop, and kept for the backward pass (0.9 KB per vertex): inside the loop they were ~10 launches per chunk of
        # microseconds of work each, and a small launch that has to find free CUs beside a 5 ms streaming pass on the helper
        # stream takes 0.2-0.4 ms — 108 ms of a 890 ms step at 44 chunks (profiles/r03_kernel_stats_gngf_learning.txt).
        hs_all = HpdVertexFunction._hidden(vertex_coords(0, NV, vstride, dev), params, n_layers)
        if keep_z:
            # one query per forward: memory the driver reports free plus what torch's allocator holds in unused cached blocks
            # (the previous step's kept chunks come back from there), minus the reserve for the backward's own buffers.  Taken
            # AFTER the hidden layers of all NV vertices are allocated (they live until the backward pass), and less the backward's
            # dH (NV x last hidden width): ~1.5 GB at a million vertices that the budget used to hand to the logits
            free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            budget = min(TUNING.hpd_z_cache_bytes, free - TUNING.hpd_z_cache_reserve - NV * W_last.shape[1] * 4)
        parts, parts_free, n_parts = None, None, 0          # row partials of the GEMM epilogue (two buffers when pipelined)
        L = mw.shape[1] if mw is not None else 0
        for u0 in range(0, NV, rows):
            n = min(rows, NV - u0)
            hs = [h[u0:u0 + n] for h in hs_all]
            z = probs[u0:u0 + n] if keep_probs else None
            if keep_z:
                need = n * T * 4
                if cached + need <= budget:
                    try:
                        z = zcache[u0] = torch.empty((n, T), dtype=_f32, device=dev)
                        cached += need
                    except torch.OutOfMemoryError:      # another process took the memory meanwhile: recompute from here on
                        budget = 0
            sb = None
            if z is None:
                sb = n_scratch % len(scratch)
                n_scratch += 1
                z = scratch[sb][:n]
                if scratch_free[sb] is not None:
                    main.wait_event(scratch_free[sb])
            # statistics in the GEMM's epilogue when the chunk is whole 128 x 128 tiles on the split-bf16 kernel
            epi = (TUNING.hpd_epilogue_stats and TUNING.hpd_gemm_split_bf16 and not keep_probs and n % 128 == 0 and T % 128 == 0
                   and W_last.shape[1] % 32 == 0 and K * 64 <= T and T < (1 << 22))
            if epi:
                if parts is None:
                    parts = [torch.empty((min(rows, NV), T // 64, 2), dtype=_f32, device=dev) for _ in range(2 if pipelined else 1)]
                    parts_free = [None] * len(parts)
                pb = n_parts % len(parts)
                n_parts += 1
                if parts_free[pb] is not None:
                    main.wait_event(parts_free[pb])
                call("gngf_linear_fwd_rowstats", ptr(hs[-1]), ptr(W_last), ptr(b_last), ptr(z), ptr(parts[pb]), n, T, W_last.shape[1],
                     stream_ptr())
            else:
                with _split_gemm():
                    call("gngf_linear_fwd", ptr(hs[-1]), ptr(W_last), ptr(b_last), ptr(z), n, T, W_last.shape[1], ACT_NONE, stream_ptr())

            def passes():
                if epi:
                    call("gngf_rowstats_topk", ptr(z), ptr(parts[pb]), ptr(tv[u0:u0 + n]), ptr(ti[u0:u0 + n]), ptr(rowstat[u0:u0 + n]),
                         n, T, K, stream_ptr())
                    if pbar is not None:
                        call("gngf_pbar_accumulate", ptr(z), ptr(rowstat[u0:u0 + n]), ptr(mw[u0:u0 + n]), L, ptr(pbar), n, T, stream_ptr())
                else:
                    call("gngf_logits_topk_pbar", ptr(z), ptr(tv[u0:u0 + n]), ptr(ti[u0:u0 + n]), ptr(rowstat[u0:u0 + n]),
                         ptr(mw[u0:u0 + n] if pbar is not None else None), L if pbar is not None else 0, ptr(pbar), n, T, K, stream_ptr())
            if pipelined:
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    passes()
                    if sb is not None:
                        scratch_free[sb] = torch.cuda.Event()
                        scratch_free[sb].record(side)
                    if epi:
                        parts_free[pb] = torch.cuda.Event()
                        parts_free[pb].record(side)
                continue
            if keep_probs:      # dense distribution requested (small shapes): softmax in place, p-bar by GEMM
                call("gngf_softmax_topk", ptr(z), ptr(tv[u0:u0 + n]), ptr(ti[u0:u0 + n]), ptr(rowstat[u0:u0 + n]), n, T, K, stream_ptr())
                if pbar is not None:
                    gemm_acc(mw[u0:u0 + n], z, pbar, L, T, n, ta=True, tb=False)
            else:               # streaming: the logits are only read (stats + top-K in one pass — or from the GEMM's partials —, p-bar in a second)
                passes()
        if pipelined:
            main.wait_stream(side)
        if pbar is not None and aux is not None and aux.mean is not None:
            aux.mean(pbar)     # the loss is a nonlinear function of the batch mean: average BEFORE the log (SURVEY §8e ii)
        if aux is not None and aux.stats is not None:
            aux.stats.update(rows_total=int(NV), T=int(T), rows_per_chunk=int(rows), chunks=-(-NV // rows), chunks_kept=len(zcache))
        ctx.cfg = (NV, vstride, K, rows, n_layers, T, keep_probs)
        ctx.zcache = zcache
        ctx.hidden = hs_all if any(ctx.needs_input_grad[7:]) else None      # activations of every vertex (no recomputation in backward)
        ctx.save_for_backward(ti, mw, probs, rowstat, tv, *params)
        ctx.mark_non_differentiable(ti)
        return tv, ti, pbar, probs

    @staticmethod
    def backward(ctx, g_tv, g_ti, g_pbar, g_probs):
        NV, vstride, K, rows, n_layers, T, keep_probs = ctx.cfg
        ti, mw, probs, rowstat, tv = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        W_last, b_last = params[-2], params[-1]
        dev = W_last.device
        grads = [torch.zeros_like(p) for p in params]
        hs_all, ctx.hidden = getattr(ctx, "hidden", None), None
        if hs_all is None:
            hs_all = HpdVertexFunction._hidden(vertex_coords(0, NV, vstride, dev), params, n_layers)
        dH = torch.zeros((NV, W_last.shape[1]), dtype=_f32, device=dev)       # d loss / d (last hidden layer), every vertex
        g_tv = _c(g_tv) if g_tv is not None else None
        g_pbar = _c(g_pbar) if (g_pbar is not None and mw is not None) else None
        g_probs = _c(g_probs) if g_probs is not None else None
        L = mw.shape[1] if mw is not None else 0
        lowrank = g_probs is None           # no dense gradient on the distribution: stream from (kept or recomputed) logits
        zcache, ctx.zcache = (getattr(ctx, "zcache", None) or {}), None
        if not lowrank:
            zcache = {}
        dz_buf = None
        if lowrank:
            scratch = torch.empty((min(rows, NV) * (1 + K),), dtype=_f32, device=dev)
        pipelined = TUNING.hpd_pipeline and lowrank and NV > rows and not torch.cuda.is_current_stream_capturing()
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if pipelined else None
        pipelined = pipelined and side.stream_id != main.stream_id
        if pipelined:
            # un-kept chunks alternate between TWO logits buffers; without the memory for both (e.g. several processes sharing
            # one device) the chunks run one after the other as before
            n_chunks = -(-NV // rows)
            dz_bufs = []
            try:
                for _ in range(min(2, n_chunks - len(zcache))):
                    dz_bufs.append(torch.empty((min(rows, NV), T), dtype=_f32, device=dev))
            except torch.OutOfMemoryError:
                pipelined = False
                dz_buf = dz_bufs[0] if dz_bufs else None
                dz_bufs = None
        if pipelined:
            HpdVertexFunction._backward_pipelined(ctx.cfg, ti, mw, rowstat, tv, params, grads, g_tv, g_pbar, L, zcache, scratch, main, side,
                                                  dz_bufs, hs_all, dH)
            HpdVertexFunction._hidden_backward(hs_all, dH, params, grads, n_layers)
            return (None, None, None, None, None, None, None, *grads)
        prep_serial = None
        for u0 in range(0, NV, rows):
            n = min(rows, NV - u0)
            hs = [h[u0:u0 + n] for h in hs_all]
            dz = zcache.pop(u0, None)       # this chunk's logits, kept by the forward (freed as soon as the chunk is done)
            have_z = dz is not None
            if not have_z:
                if dz_buf is None:
                    dz_buf = torch.empty((min(rows, NV), T), dtype=_f32, device=dev)
                dz = dz_buf[:n]
            if lowrank:
                # logits (again, unless kept), then softmax / top-K / batch-mean backward in place; db of the last layer is fused in
                if not have_z:
                    with _split_gemm():
                        call("gngf_linear_fwd", ptr(hs[-1]), ptr(W_last), ptr(b_last), ptr(dz), n, T, W_last.shape[1], ACT_NONE, stream_ptr())
                Lq, Kq = (L if g_pbar is not None else 0), (K if g_tv is not None else 0)
                if (TUNING.hpd_bwd_fused and TUNING.hpd_gemm_split_bf16
                        and query("gngf_hpd_bwd_fused_applies", n, T, Lq, Kq, W_last.shape[1]) == 1):
                    if prep_serial is None:
                        prep_serial = _HpdBwdPlanes(hs_all[-1], mw, W_last, g_pbar, Lq, 2 if TUNING.hpd_bwd_two_planes else 3)
                    dotv = scratch[:n]
                    call("gngf_hpd_bwd_dot", ptr(dz), ptr(rowstat[u0:u0 + n]), ptr(g_tv[u0:u0 + n] if Kq else None),
                         ptr(tv[u0:u0 + n] if Kq else None), ptr(mw[u0:u0 + n] if Lq else None), ptr(g_pbar if Lq else None), Lq,
                         ptr(dotv), n, T, Kq, stream_ptr())
                    prep_serial.fused(dz, rowstat[u0:u0 + n], dotv, g_tv[u0:u0 + n] if Kq else None, tv[u0:u0 + n] if Kq else None,
                                      ti[u0:u0 + n] if Kq else None, hs[-1], W_last, grads[-2], grads[-1], dH[u0:u0 + n], u0, n, T, Kq)
                    continue
                call("gngf_softmax_bwd_lowrank", ptr(dz), ptr(rowstat[u0:u0 + n]),
                     ptr(g_tv[u0:u0 + n] if g_tv is not None else None), ptr(ti[u0:u0 + n]),
                     ptr(mw[u0:u0 + n] if g_pbar is not None else None), ptr(g_pbar), L if g_pbar is not None else 0,
                     ptr(grads[-1]), ptr(scratch), ptr(tv[u0:u0 + n] if g_tv is not None else None), n, T,
                     K if g_tv is not None else 0, stream_ptr())
                with _split_gemm(accumulating=True):
                    linear_bwd_weight(dz, None, hs[-1], grads[-2], None, ACT_NONE)
            else:
                if keep_probs:
                    p_chunk = probs[u0:u0 + n]
                else:   # recompute logits + softmax in the scratch (top-K indices are the saved ones)
                    call("gngf_linear_fwd", ptr(hs[-1]), ptr(W_last), ptr(b_last), ptr(dz), n, T, W_last.shape[1], ACT_NONE, stream_ptr())
                    tv_tmp = torch.empty((n, K), dtype=_f32, device=dev)
                    ti_tmp = torch.empty((n, K), dtype=_i32, device=dev)
                    call("gngf_softmax_topk", ptr(dz), ptr(tv_tmp), ptr(ti_tmp), ptr(None), n, T, K, stream_ptr())
                    p_chunk = dz
                call("gngf_softmax_bwd", ptr(p_chunk), ptr(g_tv[u0:u0 + n] if g_tv is not None else None), ptr(ti[u0:u0 + n]),
                     ptr(g_probs[u0:u0 + n]), ptr(mw[u0:u0 + n] if g_pbar is not None else None), ptr(g_pbar), L, ptr(dz), n, T,
                     K if g_tv is not None else 0, stream_ptr())
                linear_bwd_weight(dz, None, hs[-1], grads[-2], grads[-1], ACT_NONE)
            with _split_gemm(accumulating=lowrank):
                gemm_acc(dz, W_last, dH[u0:u0 + n], n, W_last.shape[1], T, ta=False, tb=False)     # dh = dz @ W_last, split over T
        HpdVertexFunction._hidden_backward(hs_all, dH, params, grads, n_layers)
        return (None, None, None, None, None, None, None, *grads)

    @staticmethod
    def _hidden_backward(hs_all, dH, params, grads, n_layers):
        """backward of the hidden layers for ALL vertices at once (behind the chunk loop: eight launches per step, not per chunk)"""
        g = dH
        for i in range(n_layers - 2, -1, -1):
            linear_bwd_weight(g, hs_all[i + 1], hs_all[i], grads[2 * i], grads[2 * i + 1], ACT_RELU)
            if i > 0:
                g = linear_bwd_input(g, hs_all[i + 1], params[2 * i], ACT_RELU)

    @staticmethod
    def _backward_pipelined(cfg, ti, mw, rowstat, tv, params, grads, g_tv, g_pbar, L, zcache, scratch, main, side, dz_bufs, hs_all, dH):
        """The low-rank backward with its chunks software-pipelined over two streams.  Per chunk: A (main) logits again unless
        kept; B (side) softmax / top-K / batch-mean backward in place, HBM bound; C (main) the dW and dh GEMMs and the small
        layers, matrix-pipe bound.  Issue order on main: A_0, A_1, C_0, A_2, C_1, ... so that C_i runs beside B_{i+1}.  Un-kept
        chunks alternate between two logits buffers; a buffer's next A is issued on main behind the C that last read it.
        Every tensor stays referenced until both streams have joined (the caching allocator knows only the allocating stream)."""
        NV, vstride, K, rows, n_layers, T, _keep = cfg
        W_last, b_last = params[-2], params[-1]
        dev = W_last.device
        n_unkept, alive, pending = 0, [], None
        Lq = L if g_pbar is not None else 0
        Kq = K if g_tv is not None else 0
        planes = 2 if TUNING.hpd_bwd_two_planes else 3
        dots = torch.empty((NV,), dtype=_f32, device=dev) if TUNING.hpd_bwd_fused else None    # row dots of every chunk (the GEMMs of chunk i read
        #                                                                                       them while the dot pass of chunk i + 1 writes its own)

        def fused(n):
            return (TUNING.hpd_bwd_fused and TUNING.hpd_gemm_split_bf16
                    and query("gngf_hpd_bwd_fused_applies", n, T, Lq, Kq, W_last.shape[1]) == 1)
        prep = (_HpdBwdPlanes(hs_all[-1], mw, W_last, g_pbar, Lq, planes) if fused(min(rows, NV)) else None)

        def stage_b(dz, n, u0):
            if fused(n):          # one read of the logits: the row dots only
                call("gngf_hpd_bwd_dot", ptr(dz), ptr(rowstat[u0:u0 + n]), ptr(g_tv[u0:u0 + n] if Kq else None),
                     ptr(tv[u0:u0 + n] if Kq else None), ptr(mw[u0:u0 + n] if Lq else None), ptr(g_pbar if Lq else None), Lq,
                     ptr(dots[u0:u0 + n]), n, T, Kq, stream_ptr())
                return
            call("gngf_softmax_bwd_lowrank", ptr(dz), ptr(rowstat[u0:u0 + n]),
                 ptr(g_tv[u0:u0 + n] if g_tv is not None else None), ptr(ti[u0:u0 + n]),
                 ptr(mw[u0:u0 + n] if g_pbar is not None else None), ptr(g_pbar), L if g_pbar is not None else 0,
                 ptr(grads[-1]), ptr(scratch), ptr(tv[u0:u0 + n] if g_tv is not None else None), n, T,
                 K if g_tv is not None else 0, stream_ptr())

        def stage_c(dz, hs, done_b, n, u0):
            main.wait_event(done_b)
            if fused(n):          # dz formed in the loaders of both GEMMs, which read the logits (dz IS the logits here)
                prep.fused(dz, rowstat[u0:u0 + n], dots[u0:u0 + n], g_tv[u0:u0 + n] if Kq else None, tv[u0:u0 + n] if Kq else None,
                           ti[u0:u0 + n] if Kq else None, hs[-1], W_last, grads[-2], grads[-1], dH[u0:u0 + n], u0, n, T, Kq)
                return
            with _split_gemm(accumulating=True):
                linear_bwd_weight(dz, None, hs[-1], grads[-2], None, ACT_NONE)
                gemm_acc(dz, W_last, dH[u0:u0 + n], n, W_last.shape[1], T, ta=False, tb=False)     # dh = dz @ W_last, split over T

        for u0 in range(0, NV, rows):
            n = min(rows, NV - u0)
            hs = [h[u0:u0 + n] for h in hs_all]
            dz = zcache.pop(u0, None)
            if dz is None:                                     # A: not kept by the forward
                dz = dz_bufs[n_unkept % len(dz_bufs)][:n]
                n_unkept += 1
                with _split_gemm():
                    call("gngf_linear_fwd", ptr(hs[-1]), ptr(W_last), ptr(b_last), ptr(dz), n, T, W_last.shape[1], ACT_NONE, stream_ptr())
            alive.extend([dz, hs])
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):                      # B
                side.wait_event(ready)
                stage_b(dz, n, u0)
                done_b = torch.cuda.Event()
                done_b.record(side)
            if pending is not None:                            # C of the previous chunk, beside this chunk's B
                stage_c(*pending)
            pending = (dz, hs, done_b, n, u0)
        if pending is not None:
            stage_c(*pending)
        main.wait_stream(side)
        side.wait_stream(main)       # the helper's next use starts behind everything issued here
        if dots is not None:
            dots.record_stream(side)
        del alive[:]


class BlendFunction(torch.autograd.Function):
    """w = blend(q) over the K top slots of each vertex (reference models.py:212-217)."""

    @staticmethod
    def forward(ctx, q, blend_code):
        q = _c(q)
        w = torch.empty_like(q)
        call("gngf_blend_fwd", ptr(q, _f32), ptr(w), q.shape[0], q.shape[1], blend_code, stream_ptr())
        ctx.save_for_backward(q)
        ctx.blend_code = blend_code
        return w

    @staticmethod
    def backward(ctx, dw):
        (q,) = ctx.saved_tensors
        dq = torch.empty_like(q)
        dw = _c(dw)            # named: a temporary passed straight to ptr() could be freed (and its block reused) before the launch
        call("gngf_blend_bwd", ptr(q), ptr(dw, _f32), ptr(dq), q.shape[0], q.shape[1], ctx.blend_code, stream_ptr())
        return dq, None


class SoftmaxTopkFunction(torch.autograd.Function):
    """Softmax(dim=-1) + nan_to_num + top-K on dense rows (reference models.py:85,111,116) for the standalone
    HashProbDistribution.forward.  apply(logits (U,T), K) -> probs (U,T), topk_probs (U,K), topk_idx (U,K) int32."""

    @staticmethod
    def forward(ctx, logits, K):
        probs = logits.detach().clone().contiguous()
        U, T = probs.shape
        tv = torch.empty((U, K), dtype=_f32, device=probs.device)
        ti = torch.empty((U, K), dtype=_i32, device=probs.device)
        call("gngf_softmax_topk", ptr(probs, _f32, "logits"), ptr(tv), ptr(ti), ptr(None), U, T, K, stream_ptr())
        ctx.save_for_backward(probs, ti)
        ctx.mark_non_differentiable(ti)
        return probs, tv, ti

    @staticmethod
    def backward(ctx, g_probs, g_tv, g_ti):
        probs, ti = ctx.saved_tensors
        U, T = probs.shape
        K = ti.shape[1]
        dz = torch.empty_like(probs)
        g_tv = _c(g_tv) if g_tv is not None else None
        g_probs = _c(g_probs) if g_probs is not None else None
        call("gngf_softmax_bwd", ptr(probs), ptr(g_tv, _f32), ptr(ti), ptr(g_probs, _f32), ptr(None), ptr(None), 0, ptr(dz), U, T,
             K if g_tv is not None else 0, stream_ptr())
        return dz, None


class TableViewFunction(torch.autograd.Function):
    """Presents the L per-level Parameters (`_hash_tables.{l}.weight`, views of one buffer) to the kernels as ONE
    (L,T,F) tensor, and hands the (L,T,F) gradient buffer to the L parameters as views WITHOUT copies:
    `.grad` of level l becomes g[l] directly (autograd's AccumulateGrad would clone each view), and the owner
    module keeps `_grad_base = g` so that data-parallel training all-reduces ONE contiguous buffer."""

    @staticmethod
    def forward(ctx, base, owner, *weights):
        ctx.owner = owner
        ctx.weights = weights
        ctx.set_materialize_grads(False)
        return base.detach()

    @staticmethod
    def backward(ctx, g):
        ws = ctx.weights
        if g is None:                      # handed over as grad_fp32 (see _grad_out)
            return (None, None, *([None] * len(ws)))
        if all(w.grad is None for w in ws):
            g = g.contiguous()
            for l, w in enumerate(ws):
                if w.requires_grad:
                    w.grad = g[l]
            ctx.owner._grad_base = g
        else:                      # gradient accumulation across several backward passes
            for l, w in enumerate(ws):
                if w.requires_grad:
                    if w.grad is None:
                        w.grad = g[l].clone()
                    else:
                        w.grad.add_(g[l])
            ctx.owner._grad_base = None
        return (None, None, *([None] * len(ws)))


# ------------------------------------------------------------------------------------------------ tiled encoder
# TUNING.encode_path (default "auto")        — "auto" | "direct" | "tiled"  (tests force a path; auto = tiled when it pays)


class BinPipeline:
    """Binning of the NEXT batch on the pixel-stage launches of the current step (one per model, `net.dp.pipeline`).  Binning
    depends on the coordinates only, and the batches of an epoch are fixed slices of one permutation, known in advance
    (reference functions.py:186-194): the caller that owns the loop (train.GraphedStep(unroll=k), train.train_epoch) announces
    the coordinates of the next batch before it runs the current one:

        net.dp.pipeline.announce(next_xy)      # (P,2) fp32 device tensor that the NEXT forward pass will be called with
        ... forward + backward of the current batch ...

    The encoder then runs the count half of that batch's binning in extra workgroups of its forward launch and the scatter half
    as tasks in the tail of its backward launch, and the next forward pass — if it is called with that very tensor, unmodified —
    finds its pixels binned.  Anything else (another tensor, an in-place edit, another shape, no backward pass in between) falls
    back to binning at the head of the step: results never depend on the pipeline."""

    def __init__(self):
        self.next_xy = None       # announced coordinates of the next batch
        self.next_ws = None       # ... and (optional) the workspace its binning must land in (train.GraphedStep across replays)
        self.ready = None         # (key, workspace) binned by the previous step's riders
        self.pending = None       # count half launched (forward), scatter half not yet (backward has not run)
        self.hits = self.misses = 0

    @staticmethod
    def _key(xy, plan):
        return (xy.data_ptr(), xy._version, tuple(xy.shape), str(xy.device), plan.geometry())

    def announce(self, xy, workspace=None):
        """workspace (optional, an ops.TiledWorkspace(plan, xy, launch=False)): where the riders put the binned batch — a caller
        that replays captured steps owns it, so that ANOTHER captured graph can be told where its pixels are (seed)."""
        self.next_xy, self.next_ws = xy, (workspace if xy is not None else None)

    def seed(self, xy, workspace):
        """Declares `xy` binned in `workspace` — by the riders of a step that ran (or, inside a replayed graph, WILL have run) before
        the forward pass this is for.  The caller vouches for the CONTENT: the workspace must hold the binning of the coordinates
        the tensor holds when the kernels run (train.GraphedStep(cross_replay=True) does)."""
        self.ready = ((xy.data_ptr(), xy._version, tuple(xy.shape), str(xy.device), workspace.geometry), workspace)

    def reset(self):
        self.next_xy = self.next_ws = self.ready = None

    def take(self, xy, plan):
        """the workspace the previous step's riders filled for exactly this tensor and plan, or None"""
        r, self.ready = self.ready, None
        if r is not None and r[0] == self._key(xy, plan):
            self.hits += 1
            return r[1]
        self.misses += 1
        return None


class DataParallel:
    """Data-parallel state of ONE model (its `dp` attribute; parallel.enable_vertex_grid_exchange / defer_vertex_stage set it
    up).  It travels with every call as an argument and the bookkeeping of a backward pass is written here, not into the
    process: two models — or a training step and an evaluation pass — never see each other's.
      exchange      callable averaging the vertex-grid gradient over the ranks in place (None: single rank)
      mean / max_   callables averaging / max-reducing a small tensor over the ranks in place (p-bar; coordinate bounds)
      defer_vertex  the encoder backward stops after the pixel stage and leaves the vertex stage's arguments in `deferred`;
                    parallel.allreduce_gradients() exchanges dG and runs it.  The backward pass then holds no collective, so
                    forward + backward replay from ONE hipGraph on every rank (eager launch gaps cost ~15 % of a step)
      deferred      arguments of the pending vertex stage (set by the last backward pass of this model, or None)
      tables_reduced  leading levels whose table gradient of the last backward pass came out of an exchanged dG
      comm_stream / comm_done   set by parallel.allreduce_gradients(overlap=True): the exchange runs on its own stream"""

    def __init__(self):
        self.exchange = self.mean = self.max_ = None
        self.defer_vertex = False
        self.deferred = None
        self.tables_reduced = 0
        self.comm_stream = None
        self.comm_done = None
        self.bin_ws = {}                # persistent counters of the two-launch binning (ops._bin_workspace)
        self.level_params = None        # the encoder's level parameters (models.py sets it): whose .grad may live in persist_grad
        self.persist_ok = False         # the loop's owner vouches that no gradient older than the current step is kept (opt-in)
        self.persist_grad = None        # the step-to-step table-gradient buffer (ops.PERSISTENT_TABLE_GRAD)
        self.persist_gen = 0            # ... and how often it has been handed to a backward pass (or re-allocated)
        self.pipeline = BinPipeline()   # the next batch's binning riding on this step's pixel-stage launches
        self.step_config = None         # the StepConfig of the model's most recent forward pass through the fused encoder
        self.zero = None                # parallel.shard_direct_levels: this rank's share of the direct levels' rows
        self.world = 1
        self.group = None

    @property
    def active(self):
        return self.exchange is not None


class StepLink:
    """What the decoder of ONE forward pass hands to the encoder of the same pass when the backward pass runs (created per
    forward by the model, passed to encode_apply and decoder_apply, kept alive by their autograd contexts):
      pending_reduce  the decoder's slab reduction, waiting for the tiled encoder backward's launch to ride on
      absmax          (hint, data_ptr, version) — bound on max |d enc| for the tensor with that address / version
      loss_value      the fused pixel loss's value, deferred so that it rides on the same launch
      promise         (promised, arrived) device scalars of the fused training decoder's loss gradient (checked on the device)
    loss_value_aside: the caller owns the whole step and joins the loss value itself (train.GraphedStep)."""
    __slots__ = ("pending_reduce", "absmax", "loss_value", "promise", "loss_value_aside", "defer_zero", "zero_hidden", "zero_request",
                 "__weakref__")

    def __init__(self, loss_value_aside=False, defer_zero=False, zero_hidden=False):
        self.pending_reduce = self.absmax = self.loss_value = self.promise = None
        self.loss_value_aside = bool(loss_value_aside)
        # defer_zero: the caller expects the fused training decoder to follow the encoder in this forward pass: the encoder then
        # does NOT clear its table-gradient buffer in rider workgroups of the binning launch but leaves it here (zero_request);
        # gngf_decoder_train clears it between its MFMAs.  If nobody took it, the encoder backward clears it itself.
        self.defer_zero = bool(defer_zero)
        # zero_hidden: that decoder is the one-launch training kernel, which clears between its MFMAs at next to no cost (75 MiB in
        # 4.4 us at cfg2; the 64-feature backward kernel does not: 3 GiB in 0.49 ms at cfg5)
        self.zero_hidden = bool(zero_hidden)
        self.zero_request = None

    def take_absmax(self, genc):
        h, self.absmax = self.absmax, None
        # any in-place edit (or another tensor) since invalidates the bound
        return h[0] if (h is not None and h[1] == genc.data_ptr() and h[2] == genc._version) else None

    def take_reduce(self, device):
        r = self.pending_reduce
        if r is not None and r["device"] == device and r["stream"] == torch.cuda.current_stream(device).cuda_stream:
            self.pending_reduce = None
            return r
        return None

    def take_loss_value(self, device):
        r = self.loss_value
        if r is not None and r["device"] == device and r["stream"] == torch.cuda.current_stream(device).cuda_stream:
            self.loss_value = None
            return r
        return None

    def promise_ptrs(self):
        pr = self.promise
        return (ptr(pr[0]), ptr(pr[1])) if pr is not None else (ptr(None), ptr(None))

    def flush_reduce(self):
        """Launches the slab reduction if nobody picked it up (no tiled encoder backward followed the decoder backward)."""
        r, self.pending_reduce = self.pending_reduce, None
        if r is None:
            return
        with torch.cuda.stream(r["stream_obj"]):      # (the Stream object itself: an ExternalStream wrapped around the default
            # stream's raw handle is a different stream to torch — reductions launched through one came out corrupted)
            call("gngf_decoder_reduce", ptr(r["slabs"]), *[_ct.c_void_p(a) for a in r["gptrs"]], ptr(None), *self.promise_ptrs(),
                 r["P"], r["in_dim"], r["out_dim"], stream_ptr())

    def join_loss_value(self):
        """Launches the deferred loss value if it found no launch to ride on (end of the step)."""
        r, self.loss_value = self.loss_value, None
        if r is None:
            return
        with torch.cuda.stream(r["stream_obj"]):
            call("gngf_mse_fwd", ptr(r["pred"]), ptr(r["label"]), ptr(r["loss"]), ptr(r["ws"]), r["pred"].numel(), stream_ptr())
# TUNING.tiled_chunk (default None)          — max pixels per (tile, chunk) work item; None: about two average tiles' worth (see EncodePlan)
# TUNING.tiled_min_pixels (default 1 << 14)  — below this the binning overhead is not worth it
# a level is staged while N_l^2 <= this * P (sparser levels: direct form).  Measured at the cfg4 shape (2^20 px, N -> 4095):
# 8 also stages the level with 7.6 cells per pixel — its direct backward drops 412 -> 210 us but the vertex stage and the dense
# per-item images of that level cost more (step 1.43 -> 1.58 ms)
# TUNING.tiled_cells_per_pixel (default 4.0)
# TUNING.tiled_lds_limit (default 48 * 1024)    — forward image; the backward image (64-bit accumulators) is twice this
# TUNING.tiled_tile_shift_bias (default 0)      — +1: four times as many (smaller) spatial tiles as "<= 16 cells of the finest staged level per tile side"


class EncodePlan:
    """Host-side geometry of the tiled form for one (batch size, level set): which levels are staged, the tile
    grid, LDS budget, workspace sizes.  Pure integer arithmetic (unit-tested on CPU)."""

    def __init__(self, P, n_ls_host, F, path=None):
        path = path or TUNING.encode_path
        self.P, self.F = int(P), int(F)
        self.n_ls_host = [int(n) for n in n_ls_host]
        L = len(self.n_ls_host)
        self.L = L
        Ls = 0
        if path != "direct" and (path == "tiled" or P >= TUNING.tiled_min_pixels):
            for n in self.n_ls_host:          # resolutions ascend: stage the leading levels that are dense enough
                dense_enough = n * n <= TUNING.tiled_cells_per_pixel * max(P, 1)
                small_enough = (n + 2) * (n + 2) * F * 4 <= (64 << 20)
                if dense_enough and small_enough:
                    Ls += 1
                else:
                    break
        self.Ls = Ls
        if Ls == 0:
            return
        nmax = max(self.n_ls_host[:Ls])
        shift = max(0, min(6, int(_math.ceil(_math.log2(max(nmax / 16.0, 1.0)))) + TUNING.tiled_tile_shift_bias))
        while True:
            TS = 1 << shift
            lds = sum((n // TS + 3) ** 2 for n in self.n_ls_host[:Ls]) * F * 4
            if lds <= TUNING.tiled_lds_limit or shift == 6:
                break
            shift += 1
        if lds > TUNING.tiled_lds_limit:      # drop the finest staged levels until the sub-grids fit
            while Ls > 0 and sum((n // TS + 3) ** 2 for n in self.n_ls_host[:Ls]) * F * 4 > TUNING.tiled_lds_limit:
                Ls -= 1
            self.Ls = Ls
            if Ls == 0:
                return
            lds = sum((n // TS + 3) ** 2 for n in self.n_ls_host[:Ls]) * F * 4
        self.tile_shift = shift
        self.ntiles = 1 << (2 * shift)
        self.lds_bytes = int(lds)
        # one work item per tile for the typical tile (an item that is a sliver of a split tile still stages every
        # sub-grid): twice the mean pixels per tile, as a power of two in [1024, 4096].  Measured at 2^20 px / 1024 tiles:
        # chunk 1024 -> 0.70 ms per step, 2048..4096 -> 0.67 ms (tools/ab_chunk.py).
        if TUNING.tiled_chunk is not None:
            self.chunk = int(TUNING.tiled_chunk)
        else:
            want = 2 * max(1, -(-P // self.ntiles))
            self.chunk = min(4096, max(1024, 1 << (want - 1).bit_length()))
        self.NB = max(1, min(TUNING.bin_blocks_max, -(-P // TUNING.bin_pixels_per_block)))        # more binning workgroups do not help (measured: tools/perf_bin.py)
        self.max_items = -(-P // self.chunk) + self.ntiles
        self.vtot = sum((n + 2) ** 2 for n in self.n_ls_host[:Ls])
        self.n_ls_c = (_ct.c_int32 * L)(*self.n_ls_host)

    def geometry(self):
        """what a binned workspace depends on besides the coordinates"""
        return (self.P, tuple(self.n_ls_host[:self.Ls]), self.F, self.Ls, self.tile_shift, self.chunk, self.NB, self.max_items)

    def interleaved(self, backward):
        """True when the launcher will run the level-interleaved pixel-stage kernel for this plan (the library's own decision —
        gngf_tiled_interleaved_applies: F = 2, <= 16 staged levels, the image fits the LDS, the process switch is on).  Only
        the interleaved BACKWARD fills the fixed-point vertex grid, so the buffers of a backward pass are chosen with it."""
        if self.Ls == 0:
            return False
        return bool(query("gngf_tiled_interleaved_applies", self.n_ls_c, self.Ls, self.F, self.tile_shift, self.lds_bytes,
                          int(bool(backward))))


# TUNING.bin_blocks_max (default 128)            — binning workgroups: more do not help (measured: tools/perf_bin.py, tools/perf_overlap.py)
# TUNING.bin_pixels_per_block (default 8192)
# TUNING.two_launch_binning (default True)     — count -> scatter (the scans ride inside the scatter launch) when the launch carries no gradient clear
_BIN_WORKSPACES = {}


def _bin_workspace(dev, ntiles, owner=None, kind="prepare"):
    """Persistent zero-initialised counters of the two-launch binning (tile totals, tile cursors, a ticket).  The kernels put
    them back to zero themselves, so one buffer serves every step — including every replay of a captured step (it is allocated
    at warm-up, outside the capture).  It belongs to ONE sequence of steps: `owner` is the model's DataParallel object (one per
    model: two models stepping concurrently on two streams do not share counters); calls without an owner (the op used on its
    own) share one buffer per device and tiling and must not overlap in time."""
    store = owner.bin_ws if owner is not None else _BIN_WORKSPACES
    # kind: the two binning protocols keep different things in their counters ("prepare": totals / cursors / ticket, left zero;
    # "reserve" — gngf_bin_pixels2 and the riders: running cursors, never reset) and must not share a buffer
    key = (dev.type, dev.index, int(ntiles)) if kind == "prepare" else (dev.type, dev.index, int(ntiles), kind)
    w = store.get(key)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            # first use inside a capture: the four-launch binning (no allocation + memset in the graph) — correct, ~10 us slower
            # per step for the life of that graph; one eager step before capturing (train.GraphedStep warms up) avoids it
            import warnings
            warnings.warn("gngf: the two-launch binning's counters were first needed inside a hipGraph capture; this graph uses "
                          "the four-launch binning (run one eager step of the model before capturing)", RuntimeWarning, stacklevel=3)
            return None
        w = store[key] = torch.zeros((2 * int(ntiles) + 3,), dtype=_i32, device=dev)
    return w


def _drop_bin_workspace(dev, ntiles, owner=None):
    store = owner.bin_ws if owner is not None else _BIN_WORKSPACES
    store.pop((dev.type, dev.index, int(ntiles)), None)


class TiledWorkspace:
    """Device buffers of one forward/backward pair (binning result is shared by both)."""

    def __init__(self, plan, xy, vertex=None, zero_dG=None, zero=None, zero_dG_words=1, owner=None, launch=True, clear_rows=None):
        """vertex = (tables, vert_idx, vert_w, n_ls, vstride, G): also run the vertex stage forward into G (riding on the binning
        launches); zero_dG (same shape as G) and zero (any fp32 buffer, typically the table gradient): cleared on the way.
        launch=False: buffers only (filled by bin2(), or by another step's riders through job())."""
        dev = xy.device
        P = plan.P
        self._job = None
        self.blockhist = torch.empty((plan.ntiles * (plan.NB + 1),), dtype=_i32, device=dev)
        self.tile_off = torch.empty((plan.ntiles + 1,), dtype=_i32, device=dev)
        self.tile_item_base = torch.empty((plan.ntiles + 1,), dtype=_i32, device=dev)
        self.items = torch.empty((plan.max_items, 4), dtype=_i32, device=dev)
        self.n_items = torch.empty((4,), dtype=_i32, device=dev)      # [0] items; [1..3] counters of the persistent pixel-stage kernels
        self.sorted = torch.empty((max(P, 1), 4), dtype=_f32, device=dev)
        self.geometry = plan.geometry()
        if not launch:
            return
        if vertex is None:
            call("gngf_bin_pixels", ptr(xy, _f32, "xy"), P, plan.tile_shift, plan.NB, plan.chunk, ptr(self.blockhist),
                 ptr(self.tile_off), ptr(self.tile_item_base), ptr(self.items), ptr(self.n_items), ptr(self.sorted), stream_ptr())
        else:
            # binning + vertex stage forward + the clears of the backward's gradient buffers: one chain of four launches
            tables, vert_idx, vert_w, n_ls, vstride, G = vertex
            L, T, F = tables.shape
            mode = MODE_HASH if vert_idx is None else MODE_VERTEX_TABLE
            if zero is not None and (zero.numel() % 4 or zero.data_ptr() % 16 or zero.dtype != _f32):
                zero.zero_()
                zero = None
            try:
                call("gngf_encode_tiled_prepare", ptr(xy, _f32, "xy"), P, plan.tile_shift, plan.NB, plan.chunk, ptr(self.blockhist),
                     ptr(self.tile_off), ptr(self.tile_item_base), ptr(self.items), ptr(self.n_items), ptr(self.sorted),
                     *_tab(tables), ptr(vert_idx), ptr(vert_w), ptr(n_ls), plan.n_ls_c, ptr(G), ptr(zero_dG), int(zero_dG_words), plan.Ls, F, T,
                     0 if vert_idx is None else vert_idx.shape[1], mode, vstride, 0 if vert_idx is None else vert_idx.shape[0],
                     ptr(zero), 0 if zero is None else zero.numel(), ptr(_bin_workspace(dev, plan.ntiles, owner) if TUNING.two_launch_binning else None),
                     ptr(clear_rows if vert_idx is None else None, _f32, "clear_rows"), stream_ptr())
            except Exception:
                # the count launch may have run without the scatter launch that puts the persistent counters back to zero: the
                # next step gets a fresh (zeroed) workspace instead of binning with dirty totals
                _drop_bin_workspace(dev, plan.ntiles, owner)
                raise


def _bin_job(ws, plan, xy, pws):
    """gngf_bin_job of binning `xy` into workspace `ws` (kept on the workspace: the struct and the tensors it points to must
    outlive the launches that read it — the struct is read at launch time, the tensors by the kernels)."""
    job = _lib.BinJob(ptr(xy, _f32, "xy"), plan.P, plan.tile_shift, plan.NB, plan.chunk, ptr(ws.blockhist), ptr(pws, _i32, "persistent_ws"),
                      ptr(ws.tile_off), ptr(ws.tile_item_base), ptr(ws.items), ptr(ws.n_items), ptr(ws.sorted))
    ws._job = (job, xy, pws)
    return job


def _vertex_fwd(plan, tables, vert_idx, vert_w, n_ls, vstride, G):
    L, T, F = tables.shape
    mode = MODE_HASH if vert_idx is None else MODE_VERTEX_TABLE
    call("gngf_vertex_grid_fwd", *_tab(tables), ptr(vert_idx), ptr(vert_w), ptr(n_ls), plan.n_ls_c, ptr(G), plan.Ls, F, T,
         0 if vert_idx is None else vert_idx.shape[1], mode, vstride, 0 if vert_idx is None else vert_idx.shape[0], stream_ptr())


_FIRST_LEVEL_KEYS = {}


def _first_level_keys(n_ls_host, vstride, NV, K, device):
    """(NV*K,) int64: (first level the vertex belongs to) << 32, per (vertex, k) entry — geometry only, so it is built once
    per (resolutions, table extent) with scalar comparisons (no host->device copy: usable while a stream captures) and kept."""
    key = (n_ls_host, vstride, NV, K, str(device))
    hit = _FIRST_LEVEL_KEYS.get(key)
    if hit is None:
        if len(_FIRST_LEVEL_KEYS) > 16:
            _FIRST_LEVEL_KEYS.clear()
        vid = torch.arange(NV, device=device, dtype=_i64)
        m = torch.maximum(vid % vstride, vid // vstride)                 # max(gx, gy)
        lmin = torch.zeros_like(m)
        for n in n_ls_host:                                              # first level with max(gx,gy) <= n_l + 1
            lmin += (m > n + 1).to(_i64)
        hit = _FIRST_LEVEL_KEYS[key] = (lmin << 32).repeat_interleave(K)
    return hit


def slot_order(vert_idx, n_ls_host=None, vstride=None):
    """Visiting order of the contention-free vertex backward: (vertex,k) entries sorted by (first level the vertex
    belongs to, slot).  Equal slots stay adjacent inside a level group (wave-level segmented reduction), and a wave
    only walks the levels >= its group's first level (coarse levels contain few vertices).  One device sort per table
    build (frozen HPD: once; learning: once per step, next to the HPD GEMMs).  Returns int32 (NV*K)."""
    NV, K = vert_idx.shape
    flat = vert_idx.reshape(-1).to(_i64)
    if n_ls_host is not None and vstride:
        flat = _first_level_keys(tuple(int(n) for n in n_ls_host), int(vstride), int(NV), int(K), vert_idx.device) | flat
    return torch.sort(flat, stable=True)[1].to(_i32)


_TILE_LEVEL_OFF = {}


def tile_level_offsets(plan, device):
    """(ntiles, Ls) int32: float offset of level l's sub-grid inside tile t's LDS image, -1 where the level does not fit —
    the layout rule of csrc/encode_tiled.hip::setup_tile, evaluated once per (resolutions, tiling, device) so that the
    gather pass of the backward does not re-derive it for every vertex."""
    key = (tuple(plan.n_ls_host[:plan.Ls]), plan.tile_shift, plan.lds_bytes, plan.F, str(device))
    hit = _TILE_LEVEL_OFF.get(key)
    if hit is None:
        import numpy as np
        s, ts = plan.tile_shift, 1 << plan.tile_shift
        t = np.arange(ts * ts, dtype=np.int64)
        tx, ty = t & (ts - 1), t >> s
        lo = np.zeros(ts * ts, dtype=np.int64)
        out = np.full((ts * ts, plan.Ls), -1, dtype=np.int32)
        cap = plan.lds_bytes // 4
        for l, n in enumerate(plan.n_ls_host[:plan.Ls]):
            cx, cy = (tx * n) >> s, (ty * n) >> s
            hx = np.minimum((((tx + 1) * n) >> s) + 1, n + 1)
            hy = np.minimum((((ty + 1) * n) >> s) + 1, n + 1)
            sz = (hx - cx + 1) * (hy - cy + 1) * plan.F
            fits = lo + sz <= cap
            out[fits, l] = lo[fits]
            lo = np.where(fits, lo + sz, lo)
        hit = torch.from_numpy(out).to(device)
        _TILE_LEVEL_OFF[key] = hit
    return hit


# TUNING.hash_vertex_fusion (default True)    — hash indexing, single rank: the vertex stage backward rides on the gather pass of the pixel stage


# The kernel chain of one training step at the headline shape, as a string that changes whenever the chain does: PMC traffic
# figures (profiles/traffic.json) are stamped with it and bench.py reports them only for the chain they were measured on.
STEP_CHAIN_SIGNATURE = "r5: [bin_count_ride+bin_scatter2 | riders of the previous step] > tiled_fwd_il<SRC tables>(+count riders) > decoder_train > tiled_bwd_il(+scatter tasks, reduce, mse)<hash: HDT, table rows added by the store pass> [> vertex_bwd_sorted<FROM64> (vertex-table source)]"
# TUNING.fused_vertex_fwd (default True)      — fp32 tables on the interleaved forward kernel: the vertex stage forward runs inside its staging loop
# TUNING.bin_pipeline (default True)          — ... and an announced next batch (BinPipeline) is binned by riders of this step's pixel-stage launches
# TUNING.dg64 (default True)                  — F = 2, <= 16 staged levels, bounded |genc|: 64-bit fixed-point vertex grid fed by global integer atomics
# Round 5, spatial-hash source on a single rank (no exchange): the pixel-stage backward (level-interleaved AND generic kernels) adds
# its items' exact sums — rounded to fp32 once per item and vertex — straight to the table-gradient rows hash(gx, gy): no vertex grid
# (nothing to clear, nothing to convert), no vertex-stage launch behind the interleaved kernel (vertex_bwd_hash64: 10.7 us of the
# 396 us hash step) and no partial images + gather pass behind the generic one (gather_partials: 121 / 245 us at the 4096^2 / 8192^2
# shapes).  False: round 4's chains (fixed-point grid + vertex_bwd_hash64; partial images + gather_partials<HASHFUSE>).
# TUNING.hash_direct_scatter (default True)
# TUNING.vertex_reads_dg64 (default True)     — ... and the slot-ordered vertex backward converts it on the fly (False: dg64_to_float first)


PIXEL_BWD_TRACE = None       # tests: a list that receives one record per pixel-stage backward launch (which chain ran)


def _pixel_bwd(plan, ws, n_ls, genc, dG, L, F, absmax=None, link=None, hash_fuse=None, dG64=None, next_bin=None):
    """absmax: None or (tensor, count, stride) — `count` floats `stride` apart whose maximum bounds |genc|.
    link: the StepLink of the forward pass (decoder slab reduction / loss value waiting for a launch to ride on, promise)."""
    if PIXEL_BWD_TRACE is not None:
        PIXEL_BWD_TRACE.append({"P": plan.P, "Ls": plan.Ls, "bound": absmax is not None, "dG64": dG64 is not None and absmax is not None,
                                "hash_fuse": hash_fuse is not None, "fp32_grid": dG is not None,
                                "direct_hash": hash_fuse is not None and dG is None and dG64 is None,
                                "interleaved": plan.interleaved(backward=True)})
    partials = torch.empty((plan.max_items * (plan.lds_bytes // 4),), dtype=_f32, device=genc.device)
    am, am_count, am_stride = absmax if absmax is not None else (None, 0, 0)
    ride = link.take_reduce(genc.device) if link is not None else None        # a decoder slab reduction waiting for a launch to ride on
    if ride is None:
        ride_args = [ptr(None)] * 7 + [0, 0, 0]
    else:
        ride_args = [ptr(ride["slabs"])] + [_ct.c_void_p(a) for a in ride["gptrs"]] + [ride["P"], ride["in_dim"], ride["out_dim"]]
    # the loss gradient a fused training decoder was promised vs the one that arrived: compared inside the launch
    ride_args += list(link.promise_ptrs()) if link is not None else [ptr(None), ptr(None)]
    lv = link.take_loss_value(genc.device) if link is not None else None      # ... and a loss value nobody on the critical path reads
    if lv is None:
        ride_args += [ptr(None)] * 4 + [0]
    else:
        ride_args += [ptr(lv["pred"]), ptr(lv["label"]), ptr(lv["loss"]), ptr(lv["ws"]), lv["pred"].numel()]
    # hash_fuse = (dtables (L,T,F) fp32, T): the gather pass scatters into the table gradient itself (no vertex-stage launch)
    ride_args += [ptr(hash_fuse[0], _f32, "dtables"), int(hash_fuse[1])] if hash_fuse is not None else [ptr(None), 0]
    # dG64: zeroed (vtot * F + 2) int64 grid; log2_pixels bounds the number of terms any vertex can receive
    ride_args += [ptr(dG64, _i64, "dG64"), max(1, int(plan.P - 1).bit_length())] if dG64 is not None else [ptr(None), 0]
    # next_bin: gngf_bin_job of the NEXT batch — its scatter half runs as tasks in the tail of this launch (BinPipeline)
    ride_args += [_ct.byref(next_bin) if next_bin is not None else None]
    call("gngf_encode_tiled_bwd", ptr(ws.sorted), ptr(ws.items), ptr(ws.n_items), plan.max_items, ptr(ws.tile_item_base),
         ptr(tile_level_offsets(plan, genc.device)), ptr(n_ls), plan.n_ls_c, ptr(genc, _f32, "grad"), ptr(am),
         int(am_count), int(am_stride), ptr(dG), ptr(partials), L, plan.Ls, F, plan.tile_shift, plan.lds_bytes, plan.chunk,
         *ride_args, stream_ptr())


def run_deferred_vertex_stage(dp, exchanged=False):
    """Second half of a deferred encoder backward (DataParallel.defer_vertex): dG -> table gradient.  The arguments stay
    registered on `dp` (a replayed hipGraph refills the same buffers), so this can be called after every replay.
    exchanged: the caller has already averaged dG over the ranks (parallel.allreduce_gradients does, together with the
    other gradients)."""
    if dp is None or dp.deferred is None:
        return False
    plan, tables, vert_idx, vert_w, n_ls, vstride, dG, dtables, order, gout = dp.deferred
    if dp.exchange is not None and not exchanged:
        dp.exchange(dG)
    _vertex_bwd(plan, tables, vert_idx, vert_w, n_ls, vstride, dG, dtables, None, order)
    if gout is not None and gout is not dtables:
        # fp16 table storage: the gradient autograd received is a rounded COPY of the fp32 accumulation buffer, made before
        # the vertex stage ran.  Only the staged levels are refreshed (the direct levels' slice of `gout` may already hold
        # the all-reduced gradient; their fp32 rows here are still this rank's own).
        gout[:plan.Ls].copy_(dtables[:plan.Ls])
    return True


def _vertex_bwd(plan, tables, vert_idx, vert_w, n_ls, vstride, dG, dtables, dvw, order=None, dG64=None):
    """dG64: the fixed-point vertex grid of _pixel_bwd(..., dG=None, dG64=...) read directly (slot-ordered form only)."""
    L, T, F = tables.shape
    if vert_idx is not None and order is not None:
        call("gngf_vertex_grid_bwd_sorted", *_tab(tables), ptr(vert_idx), ptr(vert_w), ptr(order, _i32, "order"), ptr(n_ls), ptr(dG),
             ptr(dG64, _i64, "dG64"), plan.vtot, ptr(dtables), ptr(dvw), plan.Ls, F, T, vert_idx.shape[1], vstride, vert_idx.shape[0],
             stream_ptr())
        return
    if dG64 is not None:
        raise ValueError("the fixed-point vertex grid is read by the slot-ordered vertex backward only")
    mode = MODE_HASH if vert_idx is None else MODE_VERTEX_TABLE
    call("gngf_vertex_grid_bwd", *_tab(tables), ptr(vert_idx), ptr(vert_w), ptr(n_ls), plan.n_ls_c, ptr(dG), ptr(dtables), ptr(dvw),
         plan.Ls, F, T, 0 if vert_idx is None else vert_idx.shape[1], mode, vstride,
         0 if vert_idx is None else vert_idx.shape[0], stream_ptr())


_SIDE_STREAMS = {}
# TUNING.use_side_stream (default True)      — False: helper-stream work is issued in line on the current stream (measurement)


def _side_stream(device):
    """One helper stream per device: work that does not depend on the binned pixels (vertex stage, zero-filling the
    gradient buffers the backward will need) runs beside the binning kernels; joins are explicit (capturable)."""
    # one helper per (device, main stream): a helper that has exchanged events with the legacy default stream must not
    # later join a hipGraph capture started on another stream (hipStreamEndCapture crashed on exactly that history:
    # tools/dbg_graphed.py), and two main streams must not serialise through one shared helper anyway
    if not TUNING.use_side_stream:
        return torch.cuda.current_stream(device)
    key = (device.type, device.index, torch.cuda.current_stream(device).stream_id)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


@_dc.dataclass(frozen=True)
class StepConfig:
    """The kernel chain ONE forward / backward pair of the fused encoder takes — decided ONCE, at the top of the forward pass
    (StepConfig.choose: a pure function of the plan, the index source, the table dtype, what the decoder of the same pass will do
    with the gradient buffer, the model's data-parallel state and ops.TUNING), recorded on the model (`net.dp.step_config`), in
    ops.STEP_TRACE and in bench.py's line (`config.step_config`).  forward() and backward() read it instead of re-deriving it.
    What only the backward pass can know (did a bound on |d enc| arrive? is this a second backward through the same graph?) can
    DEMOTE the planned chain to the general one (partial images + gather pass, zeroed allocations); that is traced too
    ("pixel_bwd" records of PIXEL_BWD_TRACE).  tests/test_step_config_cpu.py enumerates the reachable configurations and names
    the GPU parity test that runs each."""
    source: str            # "hash" | "vertex_table"
    staged: int            # levels in the tiled form (plan.Ls); 0: direct form only
    direct: int            # levels in the direct form
    binning: str           # "pipeline": two launches of its own or none (the previous step's riders) | "prepare": rides with the vertex stage | "none"
    vertex_fwd: str        # "fused" into the pixel stage's staging loop | "riders" of the binning launches | "none"
    pixel: str             # "interleaved" | "generic" | "none"        (forward and backward kernels of the staged levels)
    grad_sink: str         # where the staged levels' backward puts its sums: "table_rows" | "dG64" | "fp32_grid" | "none" (no gradient)
    table_grad: str        # "block" ([dE | grid]: ONE allocation) | "persist" (step-to-step buffer) | "alloc" | "none"
    clear: str             # who clears it: "decoder" (between its MFMAs) | "riders" (of the binning launch) | "rows" (sparse, persist) | "none"
    direct_bwd: str        # "bucketed_write" | "bucketed_or_atomics" (decided per call on the density) | "none"
    exchange: bool         # a data-parallel exchange of the vertex-grid gradient is set up

    def signature(self):
        return (f"{self.source} L{self.staged}+{self.direct} bin={self.binning} vfwd={self.vertex_fwd} px={self.pixel} sink={self.grad_sink} "
                f"dE={self.table_grad}/{self.clear} direct={self.direct_bwd}" + (" xchg" if self.exchange else ""))

    def chain(self):
        """the signature without the level COUNTS (the same kernels run whether 4 or 16 levels are staged): the key the test
        coverage table of tests/test_step_config_cpu.py is written in"""
        lv = ("tiled" if self.staged else "") + ("+" if self.staged and self.direct else "") + ("direct" if self.direct else "")
        return (f"{self.source} {lv} bin={self.binning} vfwd={self.vertex_fwd} px={self.pixel} sink={self.grad_sink} "
                f"dE={self.table_grad}/{self.clear} direct={self.direct_bwd}" + (" xchg" if self.exchange else ""))

    @staticmethod
    def choose(plan, L, T, F, P, mode, fp32_tables, needs_grad, link_defer_zero, link_zero_hidden, exchange, persist_ok, persist_alloc_ok,
               have_reserve_ws):
        """link_defer_zero / link_zero_hidden: the StepLink of the pass says a fused decoder kernel will clear the buffer it is left
        (zero_hidden: the one-launch training kernel, which clears for free).  persist_alloc_ok: the step-to-step buffer exists or may
        be allocated now (not inside a capture).  have_reserve_ws: the persistent counters of the two-launch binning exist."""
        t = TUNING
        src = "hash" if mode == MODE_HASH else "vertex_table"
        tiled = plan.Ls > 0 and P > 0
        nd = L - plan.Ls
        if not tiled:       # (direct form only: a data-parallel exchange changes nothing in the encoder's own chain)
            return StepConfig(src, 0, L, "none", "none", "none", "none", "alloc" if needs_grad else "none", "none",
                              "bucketed_or_atomics" if (needs_grad and nd > 0) else "none", False)
        il_f, il_b = plan.interleaved(backward=False), plan.interleaved(backward=True)
        fused = bool(t.fused_vertex_fwd and t.two_launch_binning and have_reserve_ws and fp32_tables and F == 2 and il_f)
        if needs_grad and (L * T * F) % 4 != 0:
            fused = False                                    # (the one-block gradient allocation wants whole 16-byte vectors)
        if not needs_grad:
            vf0 = "fused" if (fused or (t.fused_vertex_fwd and mode == MODE_HASH and not il_f)) else "riders"
            return StepConfig(src, plan.Ls, nd, "pipeline" if fused else "prepare", vf0,
                              "interleaved" if il_f else "generic", "none", "none", "none", "none", bool(exchange))
        direct_hash = bool(t.hash_direct_scatter and t.hash_vertex_fusion and mode == MODE_HASH and not exchange)
        use64 = bool(t.dg64 and F == 2 and plan.Ls <= 16 and il_b)
        sink = "table_rows" if direct_hash else ("dG64" if use64 else "fp32_grid")
        if fused:
            return StepConfig(src, plan.Ls, nd, "pipeline", "fused", "interleaved", sink, "block", "decoder" if link_defer_zero else "riders",
                              "bucketed_or_atomics" if nd > 0 else "none", bool(exchange))
        fresh = nd > 0 and mode == MODE_HASH and bucketed_plan(P, F, T, nd, True) is not None
        hidden = link_defer_zero and link_zero_hidden and L * T * F * 4 <= t.persistent_min_bytes
        persist = bool(t.persistent_table_grad and not hidden and mode == MODE_HASH and persist_ok and not exchange and (fresh or nd == 0)
                       and persist_alloc_ok)
        # generic kernels (or the interleaved one on fp16 tables): hash source — the pixel stage gathers from the tables itself too
        # (no vertex grid; the riders of the binning launch only clear), vertex-table source — vertex riders build the grid
        vf = "fused" if (t.fused_vertex_fwd and mode == MODE_HASH and not il_f) else "riders"
        return StepConfig(src, plan.Ls, nd, "prepare", vf, "interleaved" if il_b else "generic", sink,
                          "persist" if persist else "alloc", "rows" if persist else ("decoder" if link_defer_zero else "riders"),
                          "bucketed_write" if fresh else ("bucketed_or_atomics" if nd > 0 else "none"), bool(exchange))


class EncodeFunction(torch.autograd.Function):
    """Fused coords -> (P, L*F) encoder.  Levels [0, plan.Ls) run through the tiled form (vertex stage + binned,
    LDS-privatised pixel stage), levels [plan.Ls, L) through the direct form.  Same inputs / gradients as
    EncodeDirectFunction (reference models.py:486-528, 173-229, 621-655 and their autograd backward)."""

    @staticmethod
    def forward(ctx, xy, n_ls, plan, tables, vert_idx, vert_w, vstride, order=None, dp=None, link=None, sink=None):
        ctx.dp, ctx.link, ctx.sink = dp, link, sink
        xy, tables = _c(xy), _c(tables)
        L, T, F = tables.shape
        P = xy.shape[0]
        mode = MODE_HASH if vert_idx is None else MODE_VERTEX_TABLE
        K = 0 if vert_idx is None else vert_idx.shape[1]
        NV = 0 if vert_idx is None else vert_idx.shape[0]
        enc = torch.empty((P, L * F), dtype=_f32, device=tables.device)
        ws = None
        if vert_idx is not None and order is None and plan.Ls > 0 and P > 0 and ctx.needs_input_grad[3]:
            order = slot_order(vert_idx, plan.n_ls_host[:plan.Ls], vstride)
        pre = None
        ctx.next_bin = None
        ctx.fresh_direct = False
        ctx.direct_hash = False
        ctx.persist = False
        ctx.persist_cleared = None
        clear_now = None
        dev = tables.device
        tiled = plan.Ls > 0 and P > 0
        pws = _bin_workspace(dev, plan.ntiles, dp, kind="reserve") if (tiled and TUNING.two_launch_binning and TUNING.fused_vertex_fwd) else None
        # ONE decision per pass (see StepConfig): which kernels, which buffers, who clears them
        sc = ctx.step_config = StepConfig.choose(
            plan, L, T, F, P, mode, tables.dtype == _f32, bool(ctx.needs_input_grad[3]),
            bool(link is not None and link.defer_zero), bool(link is not None and link.zero_hidden),
            bool(dp is not None and dp.exchange is not None), bool(dp is not None and dp.persist_ok and getattr(dp, "level_params", None)),
            bool(dp is not None and (getattr(dp, "persist_grad", None) is not None or not torch.cuda.is_current_stream_capturing())),
            bool(pws is not None and pws.numel() >= 2 * plan.ntiles + 3))
        if dp is not None:
            dp.step_config = sc
        _trace("step_config", signature=sc.signature())
        if SEEN_STEP_CONFIGS is not None:
            SEEN_STEP_CONFIGS.add(sc.chain())
        if tiled:
            # fp32 tables on the level-interleaved kernel: the vertex stage forward runs INSIDE the pixel stage's staging loop (no
            # vertex grid G, no vertex riders) and the binning is two launches of its own — or none, when the previous step's
            # launches carried it (BinPipeline)
            fused = sc.vertex_fwd == "fused" and sc.binning == "pipeline"      # (the interleaved kernel's own binning: two launches or riders)
            use64 = sc.grad_sink == "dG64"
            big = None
            if ctx.needs_input_grad[3]:
                # grad_sink "dG64" (F = 2, <= 16 staged levels: the level-interleaved kernels): the pixel stage of the backward adds
                # its exact fixed-point sums straight into a 64-bit vertex grid (cleared here; + scale and poison words), no gather
                # pass; "fp32_grid": the generic kernels (e.g. the 4096^2 shape, whose interleaved image exceeds the LDS) accumulate
                # into a ZEROED fp32 grid; "table_rows" (hash source, single rank): no vertex-grid gradient at all — the pixel stage
                # adds to the table gradient itself
                ctx.direct_hash = sc.grad_sink == "table_rows"
                nt = tables.numel()
                if fused:
                    # ONE allocation [table gradient | vertex-grid gradient]: whoever clears the table gradient — the fused
                    # training decoder between its MFMAs, or rider workgroups of the count launch — clears both
                    ng = 0 if ctx.direct_hash else ((plan.vtot * F + 2) * 2 if use64 else plan.vtot * F)
                    big = torch.empty((nt + ((ng + 3) & ~3),), dtype=_f32, device=dev)
                    dgrid = None if ctx.direct_hash else (big[nt:nt + ng].view(_i64) if use64 else big[nt:nt + ng].view(plan.vtot, F))
                    pre = [big[:nt].view(tables.shape), dgrid, big, big]
                else:
                    dgrid = None if ctx.direct_hash else (torch.empty((plan.vtot * F + 2,), dtype=_i64, device=dev) if use64
                                                          else torch.empty((plan.vtot, F), dtype=_f32, device=dev))
                    # direct_bwd "bucketed_write": direct levels whose backward WRITES every row are left out of the clear
                    fresh = sc.direct_bwd == "bucketed_write"
                    ctx.fresh_direct = fresh
                    # table_grad "persist": with a step-to-step buffer there is no dense clear at all (the backward takes the buffer,
                    # or a zeroed allocation when the buffer is in use) — not for small tables where the training decoder clears
                    # the buffer between its MFMAs at next to no cost (at the 4096^2 shape that hidden clear of 448 MB costs the
                    # decoder 36 us — a draw against a sparse-clear LAUNCH of 7.3 M rows, a loss against the same clear riding on
                    # the vertex riders: tools/ab_persist_cfg4.sh)
                    ctx.persist = sc.table_grad == "persist"
                    if ctx.persist:
                        pre = [None, dgrid, None, None]
                        # the rows the staged levels can touch are cleared by the vertex riders of the binning launch, if the
                        # buffer is free now (else the backward pass clears them, or takes an allocation of its own)
                        clear_now = _persistent_peek(dp, tables)
                        ctx.persist_cleared = dp.persist_gen if clear_now is not None else None
                    else:
                        dt_ = torch.empty(tables.shape, dtype=_f32, device=dev)
                        pre = [dt_, dgrid, None, dt_[:plan.Ls] if fresh else dt_]
                tile_level_offsets(plan, dev)        # cached; built here so that no backward (or graph capture) uploads it
            zbuf = None if pre is None else pre[3]
            defer = zbuf is not None and link is not None and link.defer_zero and zbuf.numel() % 4 == 0
            if defer:
                link.zero_request = zbuf
            if fused:
                pipe = dp.pipeline if (dp is not None and TUNING.bin_pipeline) else None
                if pipe is not None and pipe.pending is not None:
                    # a count half whose scatter half never ran (a forward pass without its backward pass): nothing to repair —
                    # the cursors run on from job to job and the abandoned job noted where its successor starts
                    pipe.pending = None
                ws = pipe.take(xy, plan) if pipe is not None else None
                if ws is None:
                    ws = TiledWorkspace(plan, xy, launch=False)
                    zero_now = zbuf if (zbuf is not None and not defer) else None
                    call("gngf_bin_pixels2", _ct.byref(_bin_job(ws, plan, xy, pws)), ptr(zero_now), 0 if zero_now is None else zero_now.numel(),
                         stream_ptr())
                elif zbuf is not None and not defer:
                    zbuf.zero_()
                job_next = None
                nx = nws = None
                if pipe is not None:
                    nx, nws, pipe.next_xy, pipe.next_ws = pipe.next_xy, pipe.next_ws, None, None
                if (nx is not None and ctx.needs_input_grad[3] and plan.interleaved(backward=True) and nx.is_cuda and nx.dtype == _f32
                        and nx.is_contiguous() and tuple(nx.shape) == tuple(xy.shape) and nx.device == xy.device
                        and (nws is None or (nws.geometry == plan.geometry() and nws is not ws))):
                    ws_next = nws if nws is not None else TiledWorkspace(plan, nx, launch=False)
                    job_next = _bin_job(ws_next, plan, nx, pws)
                    pipe.pending = ws_next
                    ctx.next_bin = (ws_next, job_next, nx, pipe)
                call("gngf_encode_tiled_fwd_fused", ptr(ws.sorted), ptr(ws.items), ptr(ws.n_items), plan.max_items, ptr(n_ls), plan.n_ls_c,
                     *_tab(tables), ptr(vert_idx), ptr(vert_w), ptr(enc), L, plan.Ls, F, T, K, mode, vstride, NV, plan.tile_shift,
                     plan.lds_bytes, _ct.byref(job_next) if job_next is not None else None, stream_ptr())
            else:
                # binning, vertex stage and the clears of the backward's gradient buffers: ONE chain of launches (the vertex stage
                # and the clears ride on the binning kernels as extra workgroups: ops.TiledWorkspace)
                gather = sc.vertex_fwd == "fused"             # hash source on the generic kernels: no vertex grid (riders only clear)
                G = None if gather else torch.empty((plan.vtot, F), dtype=_f32, device=dev)
                ws = TiledWorkspace(plan, xy, vertex=(tables, vert_idx, vert_w, n_ls, vstride, G),
                                    zero_dG=(pre[1].view(_f32) if pre[1].dtype == _i64 else pre[1]) if (pre and pre[1] is not None) else None,
                                    zero=(pre[3] if (pre and pre[3] is not None and not defer) else None),
                                    zero_dG_words=(2 if use64 else 1), owner=dp, clear_rows=clear_now)
                if gather:
                    call("gngf_encode_tiled_fwd_fused", ptr(ws.sorted), ptr(ws.items), ptr(ws.n_items), plan.max_items, ptr(n_ls), plan.n_ls_c,
                         *_tab(tables), ptr(None), ptr(None), ptr(enc), L, plan.Ls, F, T, 0, mode, 0, 0, plan.tile_shift, plan.lds_bytes, None,
                         stream_ptr())
                else:
                    call("gngf_encode_tiled_fwd", ptr(ws.sorted), ptr(ws.items), ptr(ws.n_items), plan.max_items, ptr(n_ls), plan.n_ls_c,
                         ptr(G), ptr(enc), L, plan.Ls, F, plan.tile_shift, plan.lds_bytes, stream_ptr())
        if plan.Ls < L:
            call("gngf_encode_fwd", ptr(xy, _f32, "xy"), *_tab(tables), ptr(vert_idx, _i32, "vert_idx"),
                 ptr(vert_w, _f32, "vert_w"), ptr(n_ls, _i32, "n_ls"), ptr(enc), P, L, F, T, K, mode, vstride, NV, plan.Ls, L,
                 ptr(ws.sorted if (TUNING.direct_fwd_tile_order and ws is not None) else None, _f32, "pixel_order"), stream_ptr())
        ctx.save_for_backward(xy, n_ls, tables, vert_idx, vert_w, order)
        ctx.cfg = (P, L, F, T, K, mode, vstride, NV, plan, ws)
        ctx.pre = pre                                           # zero-filled (dtables, dG), consumed by the first backward
        return enc

    @staticmethod
    def backward(ctx, genc):
        xy, n_ls, tables, vert_idx, vert_w, order = ctx.saved_tensors
        P, L, F, T, K, mode, vstride, NV, plan, ws = ctx.cfg
        genc = _c(genc)
        dp, link, sink = ctx.dp, ctx.link, ctx.sink
        absmax = link.take_absmax(genc) if link is not None else None
        # data-parallel bookkeeping describes THIS backward only: a step that takes another path (direct form, no staged
        # levels, no deferral) must not inherit the previous step's staged-level count or deferred vertex stage
        if dp is not None:
            dp.tables_reduced = 0
            dp.deferred = None
        exchange = dp.exchange if dp is not None else None
        NONE = (None,) * 5                                      # (order, dp, link, sink) + vstride: no gradients
        pre, ctx.pre = ctx.pre, None                            # a second backward (retain_graph) allocates fresh buffers
        zbuf = None if not pre else pre[3]
        fresh_direct = bool(pre) and getattr(ctx, "fresh_direct", False)        # (a second backward gets a zeroed buffer: nothing is fresh)
        if zbuf is not None and link is not None and link.zero_request is zbuf:
            link.zero_request = None                            # the decoder that was to clear the buffer did not run: clear it here
            zbuf.zero_()
        next_bin, ctx.next_bin = getattr(ctx, "next_bin", None), None
        dtables = pre[0] if pre else None
        if dtables is None and pre and getattr(ctx, "persist", False):
            # staged levels' rows cleared (by this pass's forward, or now); the direct levels will be written
            dtables = _persistent_grad(dp, tables, plan, n_ls, getattr(ctx, "persist_cleared", None))
        _trace("table_grad", source=("persist" if (dtables is not None and dp is not None and dtables is getattr(dp, "persist_grad", None))
                                     else ("forward_alloc" if dtables is not None else "zeroed_alloc")))
        if dtables is None:
            dtables = _grad_buffer(tables)                      # a zeroed allocation of this pass's own
            fresh_direct = False
        dvw = torch.zeros_like(vert_w) if (vert_w is not None and ctx.needs_input_grad[5]) else None
        if plan.Ls > 0 and P > 0:
            dG64 = pre[1] if (pre and pre[1] is not None and pre[1].dtype == _i64) else None
            if dG64 is not None and absmax is None:            # no bound on |genc| from its producer: the per-item scales need the fp32 path
                dG64 = None
            if dG64 is not None and not plan.interleaved(backward=True):
                dG64 = None                                     # gngf_set_tiled_interleaved changed since the forward pass: fp32 path
            fuse = (dtables, T) if (TUNING.hash_vertex_fusion and vert_idx is None and exchange is None) else None
            # the forward pass planned for it (no vertex-grid gradient was allocated) and the conditions still hold: the pixel stage
            # — level-interleaved or generic kernel, with or without a bound on |d enc| — adds to the table gradient itself.
            # Otherwise (a second backward pass through the same graph): partial images + gather pass with the hash fused in, on a
            # vertex grid allocated here
            direct_hash = bool(getattr(ctx, "direct_hash", False) and pre and fuse is not None)
            # vertex-table source in slot order, single rank, no d w: the vertex stage reads the fixed-point grid itself (no
            # fp32 copy of the vertex-grid gradient, no conversion launch)
            direct64 = (dG64 is not None and TUNING.vertex_reads_dg64 and vert_idx is not None and order is not None and exchange is None
                        and dvw is None)
            dG = None if (direct64 or direct_hash) else (pre[1] if (pre and pre[1] is not None and pre[1].dtype == _f32) else
                                                         (torch.empty if dG64 is not None else torch.zeros)((plan.vtot, F), dtype=_f32, device=tables.device))
            job_next = None
            if next_bin is not None and plan.interleaved(backward=True):
                job_next = next_bin[1]
            _pixel_bwd(plan, ws, n_ls, genc, dG, L, F, absmax, link, fuse, dG64, job_next)
            if job_next is not None:
                # the next batch is binned: its forward pass picks the workspace up if it is called with that very tensor
                ws_next, _job, nx, pipe = next_bin
                pipe.ready = (BinPipeline._key(nx, plan), ws_next)
                if pipe.pending is ws_next:
                    pipe.pending = None
            if fuse is None and exchange is not None and dp.defer_vertex and dvw is None:
                # the caller exchanges dG and runs the vertex stage after backward (parallel.allreduce_gradients)
                if plan.Ls < L:
                    _direct_bwd(xy, tables, vert_idx, vert_w, n_ls, genc, dtables, dvw, P, L, F, T, K, mode, vstride, NV, plan.Ls, L,
                                fresh=fresh_direct, order=(ws.sorted if ws is not None else None))
                gout = _grad_out(dtables, tables, sink)
                dp.deferred = (plan, tables, vert_idx, vert_w, n_ls, vstride, dG, dtables, order, gout)
                dp.tables_reduced = plan.Ls
                return (None, None, None, gout, None, dvw, *NONE)
            if exchange is not None:
                exchange(dG)                        # one small all-reduce instead of the staged levels' table gradient
                dp.tables_reduced = plan.Ls
            if fuse is not None:
                dvw_t = None                        # (hash source: the vertex stage rode on the gather pass)
            elif order is not None and dvw is not None and plan.Ls < L:
                # the sorted kernel WRITES dvert_w; the direct levels below accumulate into the same buffer
                dvw_t = torch.empty_like(dvw)
                _vertex_bwd(plan, tables, vert_idx, vert_w, n_ls, vstride, dG, dtables, dvw_t, order)
            else:
                dvw_t = None
                _vertex_bwd(plan, tables, vert_idx, vert_w, n_ls, vstride, dG, dtables, dvw, order, dG64 if direct64 else None)
        else:
            dvw_t = None
        if plan.Ls < L:
            _direct_bwd(xy, tables, vert_idx, vert_w, n_ls, genc, dtables, dvw, P, L, F, T, K, mode, vstride, NV, plan.Ls, L,
                        fresh=fresh_direct, order=(ws.sorted if ws is not None else None))
        if dvw_t is not None:
            dvw = dvw + dvw_t
        return (None, None, None, _grad_out(dtables, tables, sink), None, dvw, *NONE)


def encode_apply(xy, n_ls, n_ls_host, tables, vert_idx, vert_w, vstride, path=None, order=None, dp=None, link=None, sink=None):
    """Fused encoder dispatch: tiled form for the levels it can stage, direct form for the rest.
    order: optional cached slot_order(vert_idx) (frozen tables); dp: the model's DataParallel state; link: the forward
    pass's StepLink (shared with decoder_apply); sink: see table_view()."""
    plan = EncodePlan(xy.shape[0], n_ls_host, tables.shape[2], path)
    return EncodeFunction.apply(xy, n_ls, plan, tables, vert_idx, vert_w, vstride, order, dp, link, sink)


_MSE_WORKSPACE = {}


class MseFunction(torch.autograd.Function):
    """torch.nn.MSELoss() (reference utils.py:99) on device tensors: one launch for the value, one for the gradient
    (csrc/loss.hip) instead of three framework kernels.  apply(pred, label) -> 0-dim loss."""

    @staticmethod
    def forward(ctx, pred, label):
        pred, label = _c(pred), _c(label)
        if pred.shape != label.shape:
            raise ValueError(f"pred {tuple(pred.shape)} and label {tuple(label.shape)} differ")
        if pred.numel() == 0:
            raise ValueError("MSE of an empty batch")
        dev = pred.device
        ws = _MSE_WORKSPACE.get(dev)
        if ws is None:
            ws = _MSE_WORKSPACE[dev] = torch.zeros(query("gngf_mse_workspace_floats"), dtype=_f32, device=dev)
        loss = torch.empty((), dtype=_f32, device=dev)
        call("gngf_mse_fwd", ptr(pred, _f32, "pred"), ptr(label, _f32, "label"), ptr(loss), ptr(ws), pred.numel(), stream_ptr())
        ctx.save_for_backward(pred, label)
        return loss

    @staticmethod
    def backward(ctx, gout):
        pred, label = ctx.saved_tensors
        gout = _c(gout.to(_f32))
        dpred = torch.empty_like(pred)
        call("gngf_mse_bwd", ptr(pred), ptr(label), ptr(gout), ptr(dpred), pred.numel(), stream_ptr())
        return dpred, None


def mse_loss(pred, label):
    """torch.nn.MSELoss()(pred, label).  A prediction that came out of the fused decoder together with the loss against this
    very label tensor (decoder_apply(..., mse_target=label)) hands that value over instead of launching the loss kernels."""
    fused = getattr(pred, "_gngf_fused_mse", None)
    if fused is not None and fused[0] is label and pred._version == 0:
        return fused[1]
    return MseFunction.apply(pred, label)


class JsKlFunction(torch.autograd.Function):
    """The distribution term of the reference's Loss (utils.py:122-174) on the batch-mean distribution:
    apply(pbar (L,T), gamma, eps) -> (L,) = -(gamma + eps) JS(pbar_l, uniform) + eps KL(uniform || pbar_l).
    Two launches forward (per-workgroup partial sums in double, then one thread per level), one backward
    (csrc/loss.hip) instead of ~30 framework kernels over the 32 MiB of p-bar."""

    @staticmethod
    def forward(ctx, pbar, gamma, eps):
        pbar = _c(pbar)
        L, T = pbar.shape
        out = torch.empty((L,), dtype=_f32, device=pbar.device)
        ws = torch.empty((query("gngf_js_kl_workspace_doubles", L),), dtype=torch.float64, device=pbar.device)
        call("gngf_js_kl_fwd", ptr(pbar, _f32, "pbar"), ptr(out), ptr(ws), L, T, float(gamma), float(eps), stream_ptr())
        ctx.save_for_backward(pbar)
        ctx.cfg = (float(gamma), float(eps))
        return out

    @staticmethod
    def backward(ctx, gout):
        (pbar,) = ctx.saved_tensors
        gamma, eps = ctx.cfg
        gout = _c(gout.to(_f32))
        dpbar = torch.empty_like(pbar)
        call("gngf_js_kl_bwd", ptr(pbar), ptr(gout), ptr(dpbar), pbar.shape[0], pbar.shape[1], gamma, eps, stream_ptr())
        return dpbar, None, None


def js_kl_rows(pbar, gamma, eps):
    return JsKlFunction.apply(pbar, gamma, eps)


# Keep the decoder's activated hidden layers (512 B / pixel) from forward to backward instead of recomputing them: the
# stores and loads ride under the MFMAs of kernels that leave most of the HBM bandwidth unused (decoder backward 345 -> ~230 us
# at 2^20 px).  False: recompute (no extra memory).
# TUNING.decoder_save_hidden (default True)
# The slab reduction of the decoder backward (8 us + a launch gap of ~6 us inside a replayed step) rides on the NEXT launch of
# the backward pass — the tiled encoder backward, which comes right behind it and only needs max |d enc| (taken from the
# slabs' last words) — as extra workgroups of that kernel.  If no tiled encoder backward follows, the reduction is launched
# on its own when the backward pass ends.
# TUNING.decoder_reduce_rides (default True)

def _at_end_of_backward(fn):
    """Runs fn when the running backward pass ends (autograd engine callback); outside a backward pass (a Function's
    backward called by hand) at once."""
    try:
        torch.autograd.Variable._execution_engine.queue_callback(fn)
    except RuntimeError:
        fn()


# Training steps whose pixel loss is MSELoss(rgb, target) with a gradient known up front (the loss weight): forward and backward
# of the decoder in ONE launch (gngf_decoder_train) — the hidden layers stay in registers, nothing is saved for the backward.
# The promise is CHECKED ON THE DEVICE by the consumers of the results (csrc/gngf_common.h::promise_broken: the slab reduction
# and the tiled encoder backward compare the promised scalar with the gradient autograd delivers, no synchronisation, always
# on): a broken promise turns every gradient of the step into NaN instead of handing over gradients for the wrong value.
# TUNING.decoder_train_fusion (default True)
# Without the training kernel (64 input features) the BACKWARD kernel of the two-kernel path clears the encoder's table-gradient
# buffer on the way, instead of rider workgroups of the binning launch (4 GiB at BASELINE config 5: 0.69 ms of a 2.9 ms step there)
# TUNING.decoder_bwd_clears (default True)
_GLOSS_SCALARS = {}


def _decoder_grad_buffers(ws, P, in_dim, out_dim, dev):
    # the six parameter gradients are consecutive views of ONE buffer: the data-parallel exchange all-reduces that
    # buffer in place (parallel.allreduce_gradients) instead of packing and unpacking a bucket
    flat = torch.empty((sum(w.numel() for w in ws),), dtype=_f32, device=dev)
    grads, off = [], 0
    for w in ws:
        grads.append(flat[off:off + w.numel()].view(w.shape))
        off += w.numel()
    nslabs, nslab = _lib.query("gngf_decoder_bwd_slabs", P), _lib.query("gngf_decoder_slab_floats", in_dim, out_dim)
    slabs = torch.empty((nslabs * nslab,), dtype=_f32, device=dev)
    return flat, grads, slabs, nslabs, nslab


def _decoder_fwd(ctx, enc, leaky, ws, target, gloss_known=None, link=None):
    ctx.ws_given = tuple(ws)                      # the tensors autograd will accumulate into (see _grads_are_adopted)
    ctx.link = link = link if link is not None else StepLink()
    enc = _c(enc)
    ws = [_c(w) for w in ws]
    P, in_dim = enc.shape
    out_dim = ws[4].shape[0]
    dev = enc.device
    rgb = torch.empty((P, out_dim), dtype=_f32, device=dev)
    hidden = None
    ctx.train = None
    train = (TUNING.decoder_train_fusion and gloss_known is not None and target is not None and in_dim == 32 and P > 0
             and TUNING.decoder_reduce_rides and any(ctx.needs_input_grad))
    if TUNING.decoder_save_hidden and P > 0 and any(ctx.needs_input_grad) and not train:
        hidden = torch.empty((_lib.query("gngf_decoder_hidden_floats", P),), dtype=_f32, device=dev)
    if target is not None:
        if P == 0:
            raise ValueError("MSE of an empty batch")
        target = _c(target)
        if tuple(target.shape) != (P, out_dim) or target.dtype != _f32:
            raise ValueError(f"target {tuple(target.shape)} / {target.dtype} does not match the decoder output ({P}, {out_dim}) float32")
    if train:
        key = (dev, float(gloss_known))
        gl = _GLOSS_SCALARS.get(key)
        if gl is None:
            if len(_GLOSS_SCALARS) > 64:
                _GLOSS_SCALARS.clear()
            gl = _GLOSS_SCALARS[key] = torch.full((), float(gloss_known), dtype=_f32, device=dev)
        target_c = _c(target)
        denc = torch.empty_like(enc)
        flat, grads, slabs, nslabs, nslab = _decoder_grad_buffers(ws, P, in_dim, out_dim, dev)
        zr, link.zero_request = link.zero_request, None        # the encoder's table-gradient buffer: cleared by this launch
        call("gngf_decoder_train", ptr(enc, _f32, "enc"), ptr(target_c, _f32, "target"), ptr(gl), *[ptr(w, _f32) for w in ws], ptr(rgb),
             ptr(denc), *[ptr(None)] * 6, ptr(slabs), ptr(None), ptr(zr), 0 if zr is None else zr.numel(), P, in_dim, out_dim,
             int(leaky), stream_ptr())
        ctx.train = {"denc": denc, "flat": flat, "grads": grads, "slabs": slabs, "nslabs": nslabs, "nslab": nslab,
                     "gloss": float(gloss_known), "promised": gl}
    else:
        call("gngf_decoder_fwd", ptr(enc, _f32, "enc"), *[ptr(w, _f32) for w in ws], ptr(rgb), ptr(hidden), P, in_dim, out_dim, int(leaky),
             stream_ptr())
    mse = None
    if target is not None:
        # The loss VALUE (csrc/loss.hip::mse_fwd_kernel).  Nothing in the step's critical path reads it — the backward kernel
        # forms the loss gradient from rgb and the target itself — so a caller that owns the whole step (train.GraphedStep:
        # link.loss_value_aside) defers it: it then rides on the tiled encoder backward's launch as extra workgroups (or is
        # launched by link.join_loss_value() at the end of the step); otherwise it is launched here, in stream order.
        wsp = _MSE_WORKSPACE.get(dev)
        if wsp is None:
            wsp = _MSE_WORKSPACE[dev] = torch.zeros(query("gngf_mse_workspace_floats"), dtype=_f32, device=dev)
        mse = torch.empty((), dtype=_f32, device=dev)
        if link.loss_value_aside:
            link.loss_value = {"pred": rgb, "label": target, "loss": mse, "ws": wsp, "device": dev,
                               "stream": torch.cuda.current_stream(dev).cuda_stream, "stream_obj": torch.cuda.current_stream(dev)}
        else:
            call("gngf_mse_fwd", ptr(rgb), ptr(target), ptr(mse), ptr(wsp), rgb.numel(), stream_ptr())
    ctx.save_for_backward(enc, rgb, target, *ws)
    ctx.hidden = hidden
    ctx.cfg = (P, in_dim, out_dim, int(leaky))
    return rgb, mse


def _grads_are_adopted(ctx):
    """True when autograd will ADOPT the six gradient views this backward returns (every weight's .grad is None and no tensor
    hook reads them first): only then may the slab reduction that FILLS them be launched later in the backward pass.  With an
    existing .grad (gradient accumulation over micro-batches, zero_grad(set_to_none=False), a second backward), AccumulateGrad
    runs `p.grad += g` in stream order right after this node — the reduction must have been launched by then."""
    for w in getattr(ctx, "ws_given", ()):
        if not isinstance(w, torch.Tensor) or not w.requires_grad:
            continue
        if w.grad is not None or getattr(w, "_backward_hooks", None) or not w.is_leaf:
            return False
        if getattr(w, "_post_accumulate_grad_hooks", None):
            return False
    return True


def _schedule_reduce(ctx, link, dev, slabs, flat, grads, P, in_dim, out_dim):
    rec = {"slabs": slabs, "flat": flat, "gptrs": [g.data_ptr() for g in grads], "P": P, "in_dim": in_dim,
           "out_dim": out_dim, "device": dev, "stream": torch.cuda.current_stream(dev).cuda_stream,
           "stream_obj": torch.cuda.current_stream(dev)}
    # (addresses, not the view tensors: a second reference to a gradient tensor would make autograd's AccumulateGrad
    # CLONE it — before the reduction has filled it — instead of adopting it; `flat`, their base, keeps the memory alive)
    link.flush_reduce()                            # (a reduction of an earlier backward through the same link: launch it now)
    link.pending_reduce = rec
    if _grads_are_adopted(ctx):
        _at_end_of_backward(link.flush_reduce)     # rides on the tiled encoder backward if one follows, else launched at the end
    else:
        link.flush_reduce()                        # the gradients are read (accumulated into) right after this node: reduce now


def _decoder_bwd(ctx, drgb, gloss):
    """drgb (P,out_dim) | None and gloss (0-dim) | None: gradient w.r.t. rgb and, for the fused loss, w.r.t. the MSE value."""
    enc, rgb, target, W0, b0, W1, b1, W2, b2 = ctx.saved_tensors
    P, in_dim, out_dim, leaky = ctx.cfg
    dev = enc.device
    link = ctx.link
    tr = getattr(ctx, "train", None)
    if tr is not None:
        # the forward launch already ran the backward with the promised loss gradient: hand its results over
        ctx.train = None
        if drgb is not None or gloss is None:
            raise RuntimeError("fused training decoder: rgb must feed the MSE loss only (use net.fused_mse(target) without gloss= otherwise)")
        denc, flat, grads, slabs, nslabs, nslab = tr["denc"], tr["flat"], tr["grads"], tr["slabs"], tr["nslabs"], tr["nslab"]
        arrived = _c(gloss.detach().to(_f32)).reshape(())
        link.promise = (tr["promised"], arrived)   # compared on the device by whoever consumes slabs / d enc (no sync)
        link.absmax = ((slabs[nslab - 1:], nslabs, nslab), denc.data_ptr(), denc._version)
        _schedule_reduce(ctx, link, dev, slabs, flat, grads, P, in_dim, out_dim)
        return denc, grads
    if gloss is not None and drgb is not None:
        # rgb ALSO feeds something else: fold both into one explicit gradient (plain framework ops; not the training step's path)
        drgb = drgb + gloss * (2.0 / (P * out_dim)) * (rgb - target)
        gloss = None
    fused = gloss is not None
    if fused:
        gloss = _c(gloss.to(_f32))
    else:
        drgb = _c(drgb) if drgb is not None else torch.zeros_like(rgb)
    denc = torch.empty_like(enc)
    ws = (W0, b0, W1, b1, W2, b2)
    flat, grads, slabs, nslabs, nslab = _decoder_grad_buffers(ws, P, in_dim, out_dim, dev)
    common = (ptr(enc), ptr(rgb), ptr(None if fused else drgb, _f32, "grad"), ptr(target if fused else None), ptr(gloss if fused else None),
              ptr(W0), ptr(b0), ptr(W1), ptr(b1), ptr(W2), ptr(denc))
    link.promise = None
    # the encoder's table-gradient buffer, if it was left for a decoder kernel to clear (StepLink.defer_zero) and the training
    # kernel did not run (64 input features: BASELINE config 5): this launch clears it between its MFMAs
    zr = None
    if TUNING.decoder_bwd_clears and link.zero_request is not None and link.zero_request.numel() % 4 == 0:
        zr, link.zero_request = link.zero_request, None
    zargs = (ptr(zr), 0 if zr is None else zr.numel())
    if TUNING.decoder_reduce_rides and P > 0:
        call("gngf_decoder_bwd", *common, *[ptr(None)] * 6, ptr(slabs), ptr(None), ptr(ctx.hidden), *zargs, P, in_dim, out_dim, leaky, stream_ptr())
        _schedule_reduce(ctx, link, dev, slabs, flat, grads, P, in_dim, out_dim)
        hint = (slabs[nslab - 1:], nslabs, nslab)         # the per-slab maxima: all the encoder backward needs from the slabs
    else:
        absmax = torch.empty((1,), dtype=_f32, device=dev)
        call("gngf_decoder_bwd", *common, *[ptr(g) for g in grads], ptr(slabs), ptr(absmax), ptr(ctx.hidden), *zargs, P, in_dim, out_dim, leaky,
             stream_ptr())
        hint = (absmax, 1, 0)
    ctx.hidden = None
    link.absmax = (hint, denc.data_ptr(), denc._version)
    return denc, grads


class DecoderFunction(torch.autograd.Function):
    """The reference's default decoder (in -> 64 -> 64 -> out, ReLU|LeakyReLU, Sigmoid; models.py:382-392) as ONE fused
    MFMA kernel per direction (csrc/decoder.hip).  apply(enc, leaky, link, W0, b0, W1, b1, W2, b2) -> rgb."""

    @staticmethod
    def forward(ctx, enc, leaky, link, W0, b0, W1, b1, W2, b2):
        rgb, _ = _decoder_fwd(ctx, enc, leaky, (W0, b0, W1, b1, W2, b2), None, link=link)
        return rgb

    @staticmethod
    def backward(ctx, drgb):
        denc, grads = _decoder_bwd(ctx, drgb, None)
        return (denc, None, None, *grads)


class DecoderMseFunction(torch.autograd.Function):
    """Decoder + the pixel loss of the training step in the same two launches: apply(enc, target, leaky, gloss_known, link,
    W0 .. b2) -> (rgb, mse) with mse = torch.nn.MSELoss()(rgb, target) (reference utils.py:99).  The loss gradient
    2 (rgb - target) / n is formed in the backward kernel's prologue (no loss-backward launch, no d rgb round trip), and the
    loss value — which nothing on the critical path needs — is a launch that train.GraphedStep moves onto the encoder
    backward's launch.  Same arithmetic, same bits as ops.MseFunction.  (Folding the value into the forward kernel's epilogue
    was tried: +25 us on that kernel inside the step for a 14 us launch saved.)"""

    @staticmethod
    def forward(ctx, enc, target, leaky, gloss_known, link, W0, b0, W1, b1, W2, b2):
        ctx.set_materialize_grads(False)
        rgb, mse = _decoder_fwd(ctx, enc, leaky, (W0, b0, W1, b1, W2, b2), target.detach(), gloss_known, link=link)
        return rgb, mse

    @staticmethod
    def backward(ctx, drgb, gloss):
        if drgb is None and gloss is None:
            ctx.train = None
            return (None,) * 11
        denc, grads = _decoder_bwd(ctx, drgb, gloss)
        return (denc, None, None, None, None, *grads)


def decoder_fused_ok(acts, params):
    if len(acts) != 3 or acts[2] != ACT_SIGMOID or acts[0] != acts[1] or acts[0] not in (ACT_RELU, ACT_LEAKY):
        return False
    W0, _, W1, _, W2, _ = params
    return W0.shape[0] == 64 and tuple(W1.shape) == (64, 64) and W2.shape[1] == 64 and W0.shape[1] <= 64 and W2.shape[0] <= 4


def decoder_apply(enc, acts, params, fused=None, mse_target=None, mse_gloss=None, link=None):
    """Decoder MLP dispatch (reference models.py:382-392,469-470): the fused kernel for the default 64/64 widths,
    the generic MFMA linear chain otherwise.  mse_target (P,out) float32: also evaluate MSELoss(rgb, mse_target) inside the
    fused kernels; the 0-dim loss is attached to the returned rgb as `rgb._gngf_fused_mse = (mse_target, loss)`.
    link: the forward pass's StepLink, shared with encode_apply (None: a private one)."""
    if (fused is None or fused) and decoder_fused_ok(acts, params):
        if (mse_target is not None and enc.shape[0] > 0 and mse_target.is_cuda and mse_target.dtype == _f32
                and tuple(mse_target.shape) == (enc.shape[0], params[4].shape[0])):
            # mse_gloss: the gradient that WILL arrive at the loss value (a promise of the caller who owns the step, e.g. the
            # loss weight l_mse with loss.backward() seeded with 1): forward and backward then run in one launch
            rgb, mse = DecoderMseFunction.apply(enc, mse_target, acts[0] == ACT_LEAKY, mse_gloss, link, *params)
            rgb._gngf_fused_mse = (mse_target, mse)
            return rgb
        return DecoderFunction.apply(enc, acts[0] == ACT_LEAKY, link, *params)
    if fused:
        raise ValueError("fused decoder needs hidden widths [64, 64], in <= 64, out <= 4")
    return MlpFunction.apply(enc, acts, *params)


# ------------------------------------------------------------------------------------------------ per-kernel launch closures
def encode_kernels(xy, n_ls, n_ls_host, tables, vert_idx, vert_w, vstride, genc, path=None):
    """{name: zero-arg launcher} for the encoder kernels the current dispatch uses (bench.py times each one alone
    with HIP events on the launch stream)."""
    L, T, F = tables.shape
    P = xy.shape[0]
    mode = MODE_HASH if vert_idx is None else MODE_VERTEX_TABLE
    K = 0 if vert_idx is None else vert_idx.shape[1]
    NV = 0 if vert_idx is None else vert_idx.shape[0]
    enc = torch.empty((P, L * F), dtype=_f32, device=xy.device)
    dtables = _grad_buffer(tables)
    plan = EncodePlan(P, n_ls_host, F, path)
    s = stream_ptr
    out = {}
    if plan.Ls > 0:
        G = torch.empty((plan.vtot, F), dtype=_f32, device=xy.device)
        dG = torch.zeros_like(G)
        ws = TiledWorkspace(plan, xy, vertex=(tables, vert_idx, vert_w, n_ls, vstride, G))     # as the training step does (pixels sorted inside the items)
        am = genc.abs().max().reshape(1)
        out["prepare"] = lambda: TiledWorkspace(plan, xy, vertex=(tables, vert_idx, vert_w, n_ls, vstride, G), zero_dG=dG, zero=dtables)
        out["bin_pixels"] = lambda: TiledWorkspace(plan, xy)
        out["vertex_fwd"] = lambda: _vertex_fwd(plan, tables, vert_idx, vert_w, n_ls, vstride, G)
        out["encode_fwd:tiled"] = lambda: call("gngf_encode_tiled_fwd", ptr(ws.sorted), ptr(ws.items), ptr(ws.n_items),
                                               plan.max_items, ptr(n_ls), plan.n_ls_c, ptr(G), ptr(enc), L, plan.Ls, F,
                                               plan.tile_shift, plan.lds_bytes, s())
        order = slot_order(vert_idx, plan.n_ls_host[:plan.Ls], vstride) if vert_idx is not None else None
        out["encode_bwd:tiled"] = lambda: _pixel_bwd(plan, ws, n_ls, genc, dG, L, F, (am, 1, 0), None)
        if F == 2 and plan.Ls <= 16 and plan.interleaved(backward=True):
            dG64 = torch.zeros((plan.vtot * F + 2,), dtype=_i64, device=xy.device)

            def bwd64():
                dG64.zero_()                    # (the training step clears it in rider workgroups of the binning launch)
                _pixel_bwd(plan, ws, n_ls, genc, dG, L, F, (am, 1, 0), None, None, dG64)
            out["encode_bwd:tiled+dG64"] = bwd64
        out["vertex_bwd"] = lambda: _vertex_bwd(plan, tables, vert_idx, vert_w, n_ls, vstride, dG, dtables, None, order)
    if plan.Ls < L:
        out["encode_fwd:direct"] = lambda: call("gngf_encode_fwd", ptr(xy), *_tab(tables), ptr(vert_idx), ptr(vert_w), ptr(n_ls),
                                                ptr(enc), P, L, F, T, K, mode, vstride, NV, plan.Ls, L, ptr(None), s())
        out["encode_bwd:direct"] = lambda: call("gngf_encode_bwd", ptr(xy), *_tab(tables), ptr(vert_idx), ptr(vert_w), ptr(n_ls),
                                                ptr(genc), ptr(dtables), ptr(None), P, L, F, T, K, mode, vstride, NV, plan.Ls,
                                                L, s())
    return out
