#!/bin/bash
# round 5: the bucketed backward walking the pixels in tile order — tests, kernel timings (both orders), cfg4 / cfg5 steps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bucket.py tests/test_gpu_big_shapes.py -m gpu -q -x > gpurun_out/r5_c_test.log 2>&1; rc=$?
tail -n 5 gpurun_out/r5_c_test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
IMAGES=65536 TILE_ORDER=0 timeout -k 10 300 python tools/perf_bucket.py > gpurun_out/r5_c_pb0.log 2>&1 || exit 1
IMAGES=65536 TILE_ORDER=1 timeout -k 10 300 python tools/perf_bucket.py > gpurun_out/r5_c_pb1.log 2>&1 || exit 1
grep -h "us" gpurun_out/r5_c_pb0.log gpurun_out/r5_c_pb1.log
for m in cfg5_hash_fp16 cfg4_hash; do
  for v in 1 0; do
    timeout -k 10 300 python bench.py --mode $m --no-extra-modes --no-cpu-baseline --no-full-outputs --set BUCKETED_TILE_ORDER=$v > gpurun_out/r5_c_${m}_$v.json 2> gpurun_out/r5_c_${m}_$v.err || exit 1
    python -c "
import json,sys
d=json.loads(open('gpurun_out/r5_c_${m}_$v.json').read().strip().splitlines()[-1]); print('$m tile_order=$v', d['ms_per_step'], d['ms_per_step_windows'], {k: round(x,1) for k,x in d['kernel_ms'].items()})"
  done
done
exit $rc
