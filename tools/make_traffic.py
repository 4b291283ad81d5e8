"""profiles/traffic.json from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs):
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE is in KiB and on gfx950 reports exactly half of
the bytes of wide (16 B/lane) reads (MI355X_MICROARCH.md §HBM); WRITE_SIZE (KiB) is exact for 16 B/lane stores."""
import json, os, subprocess, sys
summ = json.loads(subprocess.check_output([sys.executable, "tools/pmc_summary.py"] + sys.argv[1:]))
groups = {     # round 3 kernel names (the round-2 names stay listed: general shapes still run those kernels)
    # round 4 (gngf_bin_pixels2: the head-of-step binning of a step whose vertex stage is fused into the pixel stage)
    "bin_pixels(count+scatter)": ["gngf::bin_count_reserve_kernel", "gngf::bin_scatter3_kernel"],
    "prepare(bin+vertex_fwd+clears)": ["gngf::bin_count_vride_kernel", "gngf::bin_rowscan_kernel", "gngf::bin_scan_kernel",
                                       "gngf::bin_scatter_ride_kernel<2", "gngf::bin_scatter_kernel"],
    "encode_fwd:tiled": ["gngf::tiled_fwd_kernel<2", "gngf::tiled_fwd_il_kernel"],
    "encode_bwd:tiled": ["gngf::tiled_bwd_kernel<2", "gngf::gather_partials_kernel<2", "gngf::tiled_bwd_il_kernel", "gngf::dg64_to_float_kernel",
                         "gngf::vertex_bwd_hash64_kernel"],
    "decoder_train": ["gngf::decoder_bwd_kernel<32, false, true, false, true, true>"],     # forward + backward in one launch
    "vertex_bwd": ["gngf::vertex_bwd_sorted_kernel<2", "gngf::vertex_bwd_kernel<2"],
    "encode_fwd:direct": ["gngf::encode_fwd_kernel"],
    "encode_bwd:direct": ["gngf::encode_bwd_kernel"],
    "encode_bwd:direct(bucketed)": ["gngf::bucket_count_kernel", "gngf::bucket_prefix_kernel", "gngf::bucket_scatter_kernel",
                                    "gngf::bucket_sum_kernel"],
}
def pick(prefix, counter):          # kernel names carry their full template argument lists: match by prefix
    return sum(v.get(counter, 0.0) for k, v in summ.items() if k.startswith(prefix))
# FETCH_SIZE correction, calibrated per kernel on a known byte count as the guide asks (the 128 MiB enc / d-enc rows):
# kernels whose dominant reads are 16 B per lane report exactly half (factor 2: decoder_fwd reads 128 MiB of enc and
# FETCH_SIZE says 64.2 MiB); tiled_bwd reads its 128 MiB of d-enc rows as 8 B per lane and FETCH_SIZE already says
# 154 MiB = rows + binned pixels (factor 1).
fetch_factor = {"encode_bwd:tiled": 1.0, "prepare(bin+vertex_fwd+clears)": 1.0, "bin_pixels(count+scatter)": 1.0, "vertex_bwd": 1.0}
out = {}
for name, ks in groups.items():
    f = sum(pick(k, "FETCH_SIZE") for k in ks)
    w = sum(pick(k, "WRITE_SIZE") for k in ks)
    if f or w:
        ff = fetch_factor.get(name, 2.0)
        out[name] = {"hbm_bytes_per_launch": (ff * f + w) * 1024, "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w, "fetch_factor": ff,
                     "note": "bytes = (fetch_factor*FETCH_SIZE + WRITE_SIZE)*1024; gfx950 FETCH_SIZE counts wide (16 B/lane) reads at half"}
# matrix-pipe occupancy of the dominant kernel from the SQ pass of the same round (profiles/<tag>_pmc_sq_counters.json, optional third
# argument): SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) — what bench.py prints as roofline.mfma_busy next to `frac`,
# so that "0.9 of the fp32 MFMA peak" (144 of a tile's 244 fp32-MFMA equivalents run on the 16x faster bf16 pipe) is not read as
# "matrix pipes 90 % busy" 
sq_path = os.environ.get("GNGF_SQ_COUNTERS")
if sq_path and os.path.isfile(sq_path):
    sq = json.load(open(sq_path))
    for name, ks in groups.items():
        busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k, v in sq.items() if any(k.startswith(p_) for p_ in ks))
        cu = sum(v.get("SQ_BUSY_CU_CYCLES", 0.0) for k, v in sq.items() if any(k.startswith(p_) for p_ in ks))
        valu = sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in sq.items() if any(k.startswith(p_) for p_ in ks))
        if name in out and cu > 0:
            out[name]["mfma_busy"] = busy / (4.0 * cu)
            out[name]["valu_insts_per_dispatch"] = valu
# stamp: the kernel chain the passes were taken on (bench.py reports the figures only for that chain), commit, kernel names
sys.path.insert(0, os.getcwd())
from collision_handling_in_instantngp_amd import ops as _ops   # noqa: E402
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = os.environ.get("GNGF_COMMIT", "unknown")
out["_meta"] = {"chain": _ops.STEP_CHAIN_SIGNATURE, "commit": commit,
                "kernels": sorted(k for k in summ if any(k.startswith(p_) for ks in groups.values() for p_ in ks))}
json.dump(out, open(os.environ.get("GNGF_TRAFFIC_OUT", "profiles/traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
