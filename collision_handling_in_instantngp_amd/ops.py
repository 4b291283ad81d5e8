"""torch.autograd.Function wrappers over the C-ABI (include/gngf.h).  Device memory + stream plumbing only;
all arithmetic happens in the HIP kernels.  CPU tensors raise (no fallback)."""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

BLEND_CODES = {True: 0, None: 1, False: 2}   # should_softmax_topk_features -> GNGF_BLEND_*
MODE_HASH, MODE_VERTEX_TABLE = 0, 1

_f32, _i32, _i64 = torch.float32, torch.int32, torch.int64


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


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
    def forward(ctx, tables, idx, probs, blend_code):
        tables, idx = _c(tables), _c(idx)
        L, T, F = tables.shape
        P = idx.shape[0]
        K = 0 if idx.dim() == 3 else idx.shape[-1]
        if K:
            probs = _c(probs)
        out = torch.empty((P, F, L, 4), dtype=_f32, device=tables.device)
        call("gngf_mrhe_fwd", ptr(tables, _f32, "tables"), ptr(idx, _i64, "indices"),
             ptr(probs if K else None, _f32, "probs"), ptr(out), P, L, F, T, K, blend_code, stream_ptr())
        ctx.save_for_backward(tables, idx, probs if K else None)
        ctx.cfg = (P, L, F, T, K, blend_code)
        return out

    @staticmethod
    def backward(ctx, gout):
        tables, idx, probs = ctx.saved_tensors
        P, L, F, T, K, blend_code = ctx.cfg
        gout = _c(gout)
        dtables = torch.zeros_like(tables)
        dprobs = torch.empty_like(probs) if (K and ctx.needs_input_grad[2]) else None
        call("gngf_mrhe_bwd", ptr(tables), ptr(idx), ptr(probs), ptr(gout, _f32, "grad"), ptr(dtables), ptr(dprobs),
             P, L, F, T, K, blend_code, stream_ptr())
        return dtables, None, dprobs, None


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
        call("gngf_encode_fwd", ptr(xy, _f32, "xy"), ptr(tables, _f32, "tables"), ptr(vert_idx, _i32, "vert_idx"),
             ptr(vert_w, _f32, "vert_w"), ptr(n_ls, _i32, "n_ls"), ptr(enc), P, L, F, T, K, mode, vstride, NV,
             stream_ptr())
        ctx.save_for_backward(xy, n_ls, tables, vert_idx, vert_w)
        ctx.cfg = (P, L, F, T, K, mode, vstride, NV)
        return enc

    @staticmethod
    def backward(ctx, genc):
        xy, n_ls, tables, vert_idx, vert_w = ctx.saved_tensors
        P, L, F, T, K, mode, vstride, NV = ctx.cfg
        genc = _c(genc)
        dtables = torch.zeros_like(tables)
        dvw = torch.zeros_like(vert_w) if (vert_w is not None and ctx.needs_input_grad[4]) else None
        call("gngf_encode_bwd", ptr(xy), ptr(tables), ptr(vert_idx), ptr(vert_w), ptr(n_ls), ptr(genc, _f32, "grad"),
             ptr(dtables), ptr(dvw), P, L, F, T, K, mode, vstride, NV, stream_ptr())
        return None, None, dtables, None, dvw, None
