#!/bin/bash
# the wave-specialised fused kernels (GNGF_HPD_BWD_WS=1) against the plain ones: numerics test, then per-kernel times of both
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
GNGF_HPD_BWD_WS=1 timeout -k 10 200 python -m pytest tests/test_gpu_dense.py -x -q -m gpu -k "formed_in_the_gemm_loaders" 2>&1 | tail -3 > $OUT/t.log; cat $OUT/t.log
grep -q " passed" $OUT/t.log && ! grep -q "failed\|fault" $OUT/t.log || exit 1
cd /tmp && export TMPDIR=/tmp
for w in 0 1; do
  GNGF_HPD_BWD_WS=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pws$w -o p -- python3 $ROOT/tools/perf_hpd_bwd_fused.py > $OUT/pws$w.log 2>&1
  if grep -q "Memory access fault" $OUT/pws$w.log; then echo fault; exit 1; fi
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pws$w/**/p_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "hpd_d" in r["Name"]: print("ws=$w", r["Name"][:42], f'{float(r["AverageNs"])/1e3:9.1f} us')
PY
done
