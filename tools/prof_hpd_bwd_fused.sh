#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hbf -o p -- python3 $ROOT/tools/perf_hpd_bwd_fused.py > $OUT/prof_hbf.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/prof_hbf/**/p_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1e3:10.1f} us')
PY
tail -4 $OUT/prof_hbf.log
