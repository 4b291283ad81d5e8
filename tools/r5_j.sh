#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bucket.py tests/test_gpu_big_shapes.py -m gpu -q -x > gpurun_out/r5_j_test.log 2>&1; rc=$?
tail -n 3 gpurun_out/r5_j_test.log
if [ $rc -ne 0 ]; then exit $rc; fi
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pair in cfg4:cfg4_hash:100 cfg5:cfg5_hash_fp16:60; do
  d=${pair%%:*}; rest=${pair#*:}; m=${rest%%:*}; n=${rest##*:}
  rm -rf $OUT/prof_r05_$d
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r05_$d -o p -- python3 $ROOT/bench.py --mode $m --steps $n --warmup 3 --no-extra-modes --no-cpu-baseline --no-full-outputs > $OUT/prof_r05_$d.log 2>&1 || exit 1
  rm -f $OUT/prof_r05_$d/*kernel_trace.csv
done
cd $ROOT
for m in cfg5_hash_fp16 cfg4_hash; do
  timeout -k 10 300 python bench.py --mode $m --no-extra-modes --no-cpu-baseline --no-full-outputs > gpurun_out/r5_j_${m}.json 2> gpurun_out/r5_j_${m}.err || exit 1
  python -c "
import json
d=json.loads(open('gpurun_out/r5_j_${m}.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['ms_per_step_windows'])"
done
