#!/bin/bash
# SQ counters of the pixel-stage kernels, all layout / load-scheduling variants (tools/perf_tiled_il.py), two --pmc passes.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_tiled_a -o p -- python3 $ROOT/tools/perf_tiled_il.py > $OUT/pmc_tiled_a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_tiled_b -o p -- python3 $ROOT/tools/perf_tiled_il.py > $OUT/pmc_tiled_b.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_tiled_a $OUT/pmc_tiled_b > $OUT/pmc_tiled.json
rm -rf $OUT/pmc_tiled_a $OUT/pmc_tiled_b
python3 - <<PY
import json
d = json.load(open("$OUT/pmc_tiled.json"))
for k, v in d.items():
    if "tiled" in k or "gather" in k:
        print(k[:60], {a: round(b / 1e6, 2) for a, b in v.items() if a != "dispatches"})
PY
