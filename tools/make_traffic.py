"""profiles/traffic.json from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs):
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE is in KiB and on gfx950 reports exactly half of
the bytes of wide (16 B/lane) reads (MI355X_MICROARCH.md §HBM); WRITE_SIZE (KiB) is exact for 16 B/lane stores."""
import json, subprocess, sys
summ = json.loads(subprocess.check_output([sys.executable, "tools/pmc_summary.py"] + sys.argv[1:]))
groups = {
    "encode_fwd:tiled": ["gngf::tiled_fwd_kernel<2>"],
    "encode_bwd:tiled": ["gngf::tiled_bwd_kernel<2>", "gngf::gather_partials_kernel<2>"],
    "decoder_fwd": ["gngf::decoder_fwd_kernel<32>"],
    "decoder_bwd": ["gngf::decoder_bwd_kernel<32>", "gngf::decoder_reduce_kernel"],
    "vertex_fwd": ["gngf::vertex_fwd_kernel<2, true>"],
    "vertex_bwd": ["gngf::vertex_bwd_sorted_kernel<2>"],
}
out = {}
for name, ks in groups.items():
    f = sum(summ.get(k, {}).get("FETCH_SIZE", 0.0) for k in ks)
    w = sum(summ.get(k, {}).get("WRITE_SIZE", 0.0) for k in ks)
    if f or w:
        out[name] = {"hbm_bytes_per_launch": (2 * f + w) * 1024, "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w,
                     "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; gfx950 FETCH_SIZE counts 128-B requests at 64 B"}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
