#!/bin/bash
# the GEMM epilogue's store mode decided once instead of per value: dense tests, the three GEMMs stand-alone, three learning steps
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -m gpu -q -x > gpurun_out/r5_v_test.log 2>&1; rc=$?
tail -n 2 gpurun_out/r5_v_test.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r5_v_test.log; then exit 1; fi
timeout -k 10 200 python tools/perf_gemm_split.py 2>&1 | grep "^split=1\|^split=2" | head -4
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --mode gngf_learning --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r5_v.json 2> gpurun_out/r5_v.err || { tail -3 gpurun_out/r5_v.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/r5_v.json').read().strip().splitlines()[-1]); print('learning', round(d['ms_per_step'],1), {k: round(v,1) for k,v in sorted(d['modes']['gngf_learning'].get('entry_ms',{}).items(), key=lambda kv: -kv[1])[:5]})"
done
