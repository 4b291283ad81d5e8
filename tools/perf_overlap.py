"""Does the binning of the NEXT batch hide behind the pixel-stage backward of the current one?  (binning depends on the pixel
coordinates only.)  Times, on the headline shape: the backward alone, the binning alone, and both issued on two streams.
usage: [GNGF_LIB_PATH=<variant .so>] python tools/perf_overlap.py [bin_blocks_max] [pixels_per_block]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collision_handling_in_instantngp_amd import ops, _lib
from collision_handling_in_instantngp_amd import models as mdl
import bench
dev = torch.device("cuda")
if len(sys.argv) > 1: ops.BIN_BLOCKS_MAX = int(sys.argv[1])
if len(sys.argv) > 2: ops.BIN_PIXELS_PER_BLOCK = int(sys.argv[2])
c = bench.SHAPES["cfg2"]
xy, target, _b = bench.make_batch("cfg2", 2**20, 0, dev)
n_host = [int(v) for v in mdl.level_resolutions(c["n_min"], c["n_max"], c["L"])]
n_ls = torch.tensor(n_host, dtype=torch.int32, device=dev)
tables = (torch.rand((c["L"], c["T"], c["F"]), device=dev) - 0.5) * 2e-4
genc = torch.randn((2**20, c["L"] * c["F"]), device=dev)
ks = ops.encode_kernels(xy, n_ls, n_host, tables, None, None, 0, genc)
bwd, fwd, vb = ks["encode_bwd:tiled+dG64"], ks["encode_fwd:tiled"], ks["vertex_bwd"]
prep = ks["bin_pixels"]          # the four-launch binning chain (count, row scan, scan, scatter)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fa, fb, n=20):
    for rep in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        for _ in range(n):
            if fa:
                with torch.cuda.stream(s1): fa()
            if fb:
                with torch.cuda.stream(s2): fb()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def both(): bwd(); vb()
print(f"bin blocks <= {ops.BIN_BLOCKS_MAX}, {ops.BIN_PIXELS_PER_BLOCK} px per block", flush=True)
print(f"backward alone           {timed(bwd, None):7.1f} us")
print(f"backward + vertex bwd    {timed(both, None):7.1f} us")
print(f"forward alone            {timed(fwd, None):7.1f} us")
print(f"binning alone            {timed(None, prep):7.1f} us")
print(f"backward || binning      {timed(bwd, prep):7.1f} us")
print(f"(bwd + vertex) || binning{timed(both, prep):7.1f} us")
print(f"forward || binning       {timed(fwd, prep):7.1f} us")
