#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bucket.py tests/test_gpu_big_shapes.py -m gpu -q -x > gpurun_out/r5_m_test.log 2>&1; rc=$?
tail -n 3 gpurun_out/r5_m_test.log
if [ $rc -ne 0 ]; then exit $rc; fi
for m in cfg4_hash cfg5_hash_fp16; do
  timeout -k 10 300 python bench.py --mode $m --no-extra-modes --no-cpu-baseline --no-full-outputs > gpurun_out/r5_m_${m}.json 2> gpurun_out/r5_m_${m}.err || exit 1
  python -c "
import json
d=json.loads(open('gpurun_out/r5_m_${m}.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['ms_per_step_windows'], {k: round(x,3) for k,x in d['kernel_ms'].items()})"
done
