#!/bin/bash
# learning mode: the apply pass of the softmax backward walking 1 / 4 / 8 / 16 row blocks per workgroup (same box, alternating)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -m gpu -q -x -k "softmax or lowrank or hpd" > gpurun_out/r5_p_test.log 2>&1; rc=$?
tail -n 3 gpurun_out/r5_p_test.log
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2; do
  for rg in 1 8 4 16; do
    GNGF_SOFTMAX_BWD_ROW_GROUPS=$rg timeout -k 10 300 python bench.py --mode gngf_learning --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r5_p.json 2> gpurun_out/r5_p.err || { tail -3 gpurun_out/r5_p.err; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/r5_p.json').read().strip().splitlines()[-1]); print('learning row_groups=$rg', round(d['ms_per_step'],1), {k: round(v,2) for k,v in d['modes']['gngf_learning'].get('entry_ms',{}).items() if 'softmax' in k})"
  done
done
