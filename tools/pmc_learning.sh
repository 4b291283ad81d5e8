#!/bin/bash
# SQ counters of every kernel of one learning-mode step (two --pmc passes)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lrn_a -o p -- python3 $ROOT/bench.py --mode gngf_learning --steps 1 --warmup 1 --no-extra-modes --no-cpu-baseline > $OUT/pmc_lrn_a.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_lrn_b -o p -- python3 $ROOT/bench.py --mode gngf_learning --steps 1 --warmup 1 --no-extra-modes --no-cpu-baseline > $OUT/pmc_lrn_b.log 2>&1 &&
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_lrn_a $OUT/pmc_lrn_b > $OUT/pmc_learning.json
rm -rf $OUT/pmc_lrn_a $OUT/pmc_lrn_b
python3 - <<PY
import json
d = json.load(open("$OUT/pmc_learning.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1].get("dispatches", 0))[:8]:
    print(k[:52].ljust(52), int(v["dispatches"]), {a: round(b / 1e6, 1) for a, b in v.items() if a != "dispatches"})
PY
