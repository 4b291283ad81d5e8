"""ctypes binding of libgngf_hip.so (C-ABI declared in include/gngf.h).

The product has NO CPU fallback: if the HIP library is missing, or a tensor is not a contiguous CUDA/HIP
tensor of the expected dtype, the call raises.  torch is used only for device memory and the current stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNGF_LIB_PATH") or os.path.join(_HERE, "libgngf_hip.so")   # override: experiments only

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float

# name -> argtypes (all return int = hipError_t).  Must mirror include/gngf.h exactly.
SIGNATURES = {
    "gngf_abi_version": [],
    "gngf_hash_indices": [_P, _P, _P, _L, _I, _L, _P],
    "gngf_mrhe_fwd": [_P, _I, _P, _P, _P, _L, _I, _I, _L, _I, _I, _P],
    "gngf_mrhe_bwd": [_P, _I, _P, _P, _P, _P, _P, _L, _I, _I, _L, _I, _I, _P],
    "gngf_bilinear_fwd": [_P, _P, _P, _P, _L, _I, _I, _P],
    "gngf_bilinear_bwd": [_P, _P, _P, _P, _L, _I, _I, _P],
    "gngf_encode_fwd": [_P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _L, _I, _I, _I, _L, _I, _I, _P, _P],
    "gngf_encode_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _L, _I, _I, _I, _L, _I, _I, _P],
    "gngf_encode_bwd_bucketed_plan": [_L, _I, _L, _I, _I, _P],
    "gngf_encode_bwd_bucketed": [_P, _P, _P, _P, _L, _I, _I, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "gngf_clear_hashed_rows": [_P, _P, _I, _I, _L, _L, _P],
    "gngf_bin_pixels": [_P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "gngf_encode_tiled_prepare": [_P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _I, _L,
                                  _P, _L, _P, _P, _P],
    "gngf_vertex_grid_fwd": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _I, _L, _P],
    "gngf_vertex_grid_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _I, _L, _P],
    "gngf_encode_tiled_fwd": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gngf_bin_pixels2": [_P, _P, _L, _P],
    "gngf_encode_tiled_fwd_fused": [_P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _L, _I, _I, _I, _L, _I, _I, _P, _P],
    "gngf_set_tiled_interleaved": [_I],
    "gngf_tiled_interleaved_applies": [_P, _I, _I, _I, _I, _I],
    "gngf_debug_il_stamps": [_P],
    "gngf_encode_tiled_bwd": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I,
                              _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P, _I, _P, _P],
    "gngf_vertex_grid_bwd_sorted": [_P, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _L, _I, _I, _L, _P],
    "gngf_decoder_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "gngf_decoder_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _P],
    "gngf_decoder_reduce": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "gngf_decoder_train": [_P] * 20 + [_L, _L, _I, _I, _I, _P],
    "gngf_decoder_bwd_last_span_ns": [_P],
    "gngf_decoder_hidden_floats": [_L],
    "gngf_decoder_bwd_slabs": [_L],
    "gngf_decoder_slab_floats": [_I, _I],
    "gngf_set_gemm_split_bf16": [_I],
    "gngf_set_decoder_split_bf16": [_I],
    "gngf_set_decoder_bwd_hybrid": [_I],
    "gngf_linear_fwd": [_P, _P, _P, _P, _L, _I, _I, _I, _P],
    "gngf_linear_bwd_input": [_P, _P, _P, _P, _L, _I, _I, _I, _P],
    "gngf_linear_bwd_weight": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "gngf_gemm_acc": [_P, _P, _P, _L, _L, _L, _I, _I, _P],
    "gngf_softmax_topk": [_P, _P, _P, _P, _L, _L, _I, _P],
    "gngf_softmax_bwd_lowrank": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _L, _I, _P],
    "gngf_hpd_bwd_dot": [_P, _P, _P, _P, _P, _P, _I, _P, _L, _L, _I, _P],
    "gngf_hpd_bwd_fused_applies": [_L, _L, _I, _I, _I],
    "gngf_hpd_bwd_prepare": [_P, _P, _L, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _P],
    "gngf_hpd_bwd_fused": [_P] * 8 + [_L, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _P],
    "gngf_logits_topk_pbar": [_P, _P, _P, _P, _P, _I, _P, _L, _L, _I, _P],
    "gngf_linear_fwd_rowstats": [_P, _P, _P, _P, _P, _L, _I, _I, _P],
    "gngf_rowstats_topk": [_P, _P, _P, _P, _P, _L, _L, _I, _P],
    "gngf_pbar_accumulate": [_P, _P, _P, _I, _P, _L, _L, _P],
    "gngf_topk": [_P, _P, _P, _L, _L, _I, _P],
    "gngf_softmax_bwd": [_P, _P, _P, _P, _P, _P, _I, _P, _L, _L, _I, _P],
    "gngf_vertex_coords": [_P, _L, _L, _I, _P],
    "gngf_blend_fwd": [_P, _P, _L, _I, _I, _P],
    "gngf_blend_bwd": [_P, _P, _P, _L, _I, _I, _P],
    "gngf_vertex_multiplicity": [_P, _P, _P, _L, _I, _I, _L, _P],
    "gngf_multiplicity_weights": [_P, _P, _L, _I, _F, _P],
    "gngf_expand_vertex_table": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _L, _P],
    "gngf_mse_workspace_floats": [],
    "gngf_mse_blocks": [_L],
    "gngf_mse_fwd": [_P, _P, _P, _P, _L, _P],
    "gngf_mse_bwd": [_P, _P, _P, _P, _L, _P],
    "gngf_js_kl_workspace_doubles": [_I],
    "gngf_js_kl_fwd": [_P, _P, _P, _I, _L, _F, _F, _P],
    "gngf_js_kl_bwd": [_P, _P, _P, _I, _L, _F, _F, _P],
    "gngf_slot_bitmap_words": [_I, _I, _L],
    "gngf_distinct_slot_counts": [_P, _L, _I, _I, _I, _L, _P, _P, _P],
    "gngf_mark_batch_slots": [_P, _P, _L, _I, _P, _I, _L, _I, _L, _P, _P, _P],
    "gngf_count_slot_bits": [_P, _I, _I, _L, _P, _P],
    "gngf_adam_block_elems": [],
    "gngf_adam_step": [_P, _I, _L, _P, _P, _P, _I, _F, _F, _F, _F, _P],
}

ABI_VERSION = 13


class BinJob(ctypes.Structure):
    """include/gngf.h: gngf_bin_job — one binning job as a host struct (passed by pointer: ctypes.byref)."""
    _fields_ = [("xy", _P), ("P", _L), ("tile_shift", _I), ("NB", _I), ("chunk", _I), ("blockhist", _P), ("persistent_ws", _P),
                ("tile_off", _P), ("tile_item_base", _P), ("items", _P), ("n_items", _P), ("sorted", _P)]


_RETURNS_INT64 = {"gngf_decoder_hidden_floats", "gngf_slot_bitmap_words"}
_lib = None


class GngfLibraryError(RuntimeError):
    pass


def load():
    """Loads the shared library once.  Raises GngfLibraryError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise GngfLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C collision_handling_in_instantngp_amd/csrc`.  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GngfLibraryError(f"{LIB_PATH} does not export {name}: stale build?") from e
        fn.argtypes = argtypes
        fn.restype = _L if name in _RETURNS_INT64 else _I
    ver = lib.gngf_abi_version()
    if ver != ABI_VERSION:
        raise GngfLibraryError(f"ABI version mismatch: library {ver}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def stream_ptr():
    return _P(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous CUDA tensor (None -> NULL).  Raises on CPU tensors: no CPU path."""
    if t is None:
        return _P(0)
    if not t.is_cuda:
        raise GngfLibraryError(f"{name} is on {t.device}: the gfx950 path needs CUDA/HIP tensors (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return _P(t.data_ptr())


def query(name, *args):
    """Entry points that return a size rather than a hipError_t."""
    return getattr(load(), name)(*args)


# Optional in-step profiling (bench.py): when PROFILE is a dict, every entry point is bracketed by HIP events recorded
# on torch's current stream — the stream the kernels are launched on — and the pairs are collected per entry point.
PROFILE = None


def call(name, *args):
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args)
        e1.record()
        PROFILE.setdefault(name, []).append((e0, e1))
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with hipError_t {rc}"
                           + (" (hipErrorInvalidValue: rejected arguments)" if rc == 1 else ""))
