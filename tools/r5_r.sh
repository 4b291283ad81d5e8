#!/bin/bash
# learning mode, same box: dz formed in the loaders of the dW / dh GEMMs (HPD_BWD_FUSED) on / off; tests first
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_reference_headline.py tests/test_gpu_model.py -m gpu -q -x -k "formed_in or split or hpd or lowrank or G12 or kept_logits or headline or epilogue or learning or gngf" > gpurun_out/r5_r_test.log 2>&1; rc=$?; else rc=0; echo skipped > gpurun_out/r5_r_test.log; fi
tail -n 3 gpurun_out/r5_r_test.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r5_r_test.log; then exit 1; fi
for rep in 1 2; do
  for f in 1 0; do
    timeout -k 10 300 python bench.py --mode gngf_learning --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 --set HPD_BWD_FUSED=$f > gpurun_out/r5_r.json 2> gpurun_out/r5_r.err || { tail -3 gpurun_out/r5_r.err; exit 1; }
    if grep -q "Memory access fault" gpurun_out/r5_r.err; then tail -3 gpurun_out/r5_r.err; exit 1; fi
    python -c "
import json
d=json.loads(open('gpurun_out/r5_r.json').read().strip().splitlines()[-1]); print('learning fused=$f', round(d['ms_per_step'],1), {k: round(v,1) for k,v in sorted(d['modes']['gngf_learning'].get('entry_ms',{}).items(), key=lambda kv: -kv[1])[:7]})"
  done
done
