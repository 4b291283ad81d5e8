"""GPU timing: backward of the direct levels at the cfg4 / cfg5 shapes — fp32 atomics (gngf_encode_bwd) vs the bucketed form
(gngf_encode_bwd_bucketed) at several LDS image sizes.  Usage: python tools/perf_bucket.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from collision_handling_in_instantngp_amd import _lib, ops

DEV = "cuda"


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ORDER = None
IMAGES = tuple(int(v) for v in os.environ.get("IMAGES", "16384,32768,65536,131072").split(","))


def run(tag, P, L, F, T, levels):
    g = torch.Generator(device=DEV).manual_seed(1)
    xy = torch.rand((P, 2), device=DEV, generator=g)
    n_all = [16] * (L - len(levels)) + list(levels)
    n_ls = torch.tensor(n_all, dtype=torch.int32, device=DEV)
    genc = torch.randn((P, L * F), device=DEV, generator=g)
    dt = torch.zeros((L, T, F), device=DEV)
    tables = torch.zeros((1,), device=DEV)
    l0, l1 = L - len(levels), L
    global ORDER
    ORDER = None
    if os.environ.get("TILE_ORDER", "0") == "1":      # walk the pixels in the tiled form's binned order (64 x 64 tiles)
        plan = ops.EncodePlan(P, [16, 1024], F, "tiled")
        ORDER = ops.TiledWorkspace(plan, xy).sorted
    us = timed(lambda: _lib.call("gngf_encode_bwd", _lib.ptr(xy), _lib.ptr(tables), 0, _lib.ptr(None), _lib.ptr(None), _lib.ptr(n_ls),
                                 _lib.ptr(genc), _lib.ptr(dt), _lib.ptr(None), P, L, F, T, 0, ops.MODE_HASH, 0, 0, l0, l1, _lib.stream_ptr()))
    print(f"{tag}: atomics {us:8.1f} us", flush=True)
    ref = None
    for image in IMAGES:
        plan = (ctypes.c_int64 * 6)()
        if _lib.query("gngf_encode_bwd_bucketed_plan", P, F, T, l1 - l0, image, plan) != 1:
            print(f"{tag}: image {image}: not served")
            continue
        matrix = torch.empty((plan[3],), dtype=torch.int32, device=DEV)
        base = torch.empty((plan[4],), dtype=torch.int32, device=DEV)
        items = torch.empty((plan[5],), dtype=torch.uint8, device=DEV)
        for acc in (1, 0):
            fn = lambda: _lib.call("gngf_encode_bwd_bucketed", _lib.ptr(xy), _lib.ptr(n_ls), _lib.ptr(genc), _lib.ptr(dt), P, L, F, T, l0, l1,
                                   image, acc, _lib.ptr(matrix), _lib.ptr(base), _lib.ptr(items), _lib.ptr(ORDER), _lib.stream_ptr())
            us = timed(fn)
            print(f"{tag}: image {image:6d} B ({plan[1]} buckets/level, matrix {plan[3] * 4 / 2**20:.1f} MiB, items {plan[5] / 2**20:.0f} MiB) "
                  f"accumulate={acc}: {us:8.1f} us", flush=True)
        dt.zero_()
        fn = lambda: _lib.call("gngf_encode_bwd_bucketed", _lib.ptr(xy), _lib.ptr(n_ls), _lib.ptr(genc), _lib.ptr(dt), P, L, F, T, l0, l1,
                               image, 0, _lib.ptr(matrix), _lib.ptr(base), _lib.ptr(items), _lib.ptr(ORDER), _lib.stream_ptr())
        fn()
        torch.cuda.synchronize()
        if ref is None:
            ref = dt[l0:l1].clone()
        else:
            print(f"{tag}: image {image}: max |difference to the first image size| {float((dt[l0:l1] - ref).abs().max()):.3e}")


if __name__ == "__main__":
    run("cfg4 (F=2, T=2^22, N=2830,4095)", 2 ** 20, 16, 2, 2 ** 22, (2830, 4095))
    run("cfg5 (F=4, T=2^24, N=2352..8191)", 2 ** 20, 16, 4, 2 ** 24, (2352, 3565, 5404, 8191))
