#!/bin/bash
# kernel statistics of the cfg4 / cfg5 steps (rocprofv3 --kernel-trace --stats), summaries to gpurun_out/
set -u
TAG=${1:-r05x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in cfg5_hash_fp16 cfg4_hash; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$m -o p -- python3 $ROOT/bench.py --mode $m --steps 60 --warmup 3 --no-extra-modes --no-cpu-baseline --no-full-outputs > $OUT/prof_${TAG}_$m.log 2>&1 || exit 1
  rm -f $OUT/prof_${TAG}_$m/*kernel_trace.csv
  (cd $ROOT && python3 tools/prof_summary.py $OUT/prof_${TAG}_$m 16)
done
