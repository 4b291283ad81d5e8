"""ctypes wrapper of the C oracle (oracle/gngf_oracle_c.c) — TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libgngf_oracle_c.so")
_lib = None


def available():
    return os.path.isfile(_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_PATH)
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return lib().orc_num_threads()


def encode_fwd(xy, tables, n_ls, vert_idx=None, vert_w=None, vstride=0):
    L, T, F = tables.shape
    P = xy.shape[0]
    K = 0 if vert_idx is None else vert_idx.shape[1]
    enc = np.empty((P, L * F), np.float32)
    lib().orc_encode_fwd(_p(xy), _p(tables), _p(vert_idx), _p(vert_w), _p(n_ls), _p(enc), ctypes.c_int64(P), L, F,
                         ctypes.c_int64(T), K, vstride)
    return enc


def encode_bwd(xy, tables, n_ls, genc, vert_idx=None, vert_w=None, vstride=0):
    L, T, F = tables.shape
    P = xy.shape[0]
    K = 0 if vert_idx is None else vert_idx.shape[1]
    dt = np.zeros_like(tables)
    lib().orc_encode_bwd(_p(xy), _p(tables), _p(vert_idx), _p(vert_w), _p(n_ls), _p(genc), _p(dt), ctypes.c_int64(P), L, F,
                         ctypes.c_int64(T), K, vstride)
    return dt


def encode_bwd_f64(xy, table_shape, n_ls, genc, vert_idx=None, vert_w=None, vstride=0, exact_products=False):
    """the table gradient with the reference's fp32 terms summed in double precision: (L,T,F) float64
    (exact_products: the products g * c (* w) in double as well — the exact gradient of the fp32 inputs)"""
    L, T, F = table_shape
    P = xy.shape[0]
    K = 0 if vert_idx is None else vert_idx.shape[1]
    dt = np.zeros((L, T, F), np.float64)
    lib().orc_encode_bwd_f64(_p(xy), _p(vert_idx), _p(vert_w), _p(n_ls), _p(genc), _p(dt), ctypes.c_int64(P), L, F,
                             ctypes.c_int64(T), K, vstride, int(bool(exact_products)))
    return dt


def decoder_fwd(x, W, B):
    P, in_dim = x.shape
    out_dim = W[2].shape[0]
    h1 = np.empty((P, 64), np.float32)
    h2 = np.empty((P, 64), np.float32)
    y = np.empty((P, out_dim), np.float32)
    lib().orc_decoder_fwd(_p(x), _p(W[0]), _p(B[0]), _p(W[1]), _p(B[1]), _p(W[2]), _p(B[2]), _p(h1), _p(h2), _p(y),
                          ctypes.c_int64(P), in_dim, out_dim)
    return y, h1, h2


def decoder_bwd(x, h1, h2, y, dy, W):
    P, in_dim = x.shape
    out_dim = W[2].shape[0]
    dx = np.empty_like(x)
    g = [np.empty_like(W[0]), np.empty(64, np.float32), np.empty_like(W[1]), np.empty(64, np.float32), np.empty_like(W[2]),
         np.empty(out_dim, np.float32)]
    lib().orc_decoder_bwd(_p(x), _p(h1), _p(h2), _p(y), _p(dy), _p(W[0]), _p(W[1]), _p(W[2]), _p(dx), *[_p(a) for a in g],
                          ctypes.c_int64(P), in_dim, out_dim)
    return dx, g
