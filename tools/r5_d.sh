#!/bin/bash
# round 5: direct forward in tile order (A/B per shape) + the tests that aborted once (GC during capture, now disabled there)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bucket.py tests/test_gpu_big_shapes.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r5_d_test.log 2>&1; rc=$?
tail -n 5 gpurun_out/r5_d_test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi
for m in cfg5_hash_fp16 cfg4_hash; do
  for v in 1 0; do
    timeout -k 10 300 python bench.py --mode $m --no-extra-modes --no-cpu-baseline --no-full-outputs --set DIRECT_FWD_TILE_ORDER=$v > gpurun_out/r5_d_${m}_$v.json 2> gpurun_out/r5_d_${m}_$v.err || exit 1
    python -c "
import json,sys
d=json.loads(open('gpurun_out/r5_d_${m}_$v.json').read().strip().splitlines()[-1]); print('$m fwd_tile_order=$v', d['ms_per_step'], d['ms_per_step_windows'], {k: round(x,3) for k,x in d['kernel_ms'].items()})"
  done
done
exit $rc
