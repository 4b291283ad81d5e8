#!/bin/bash
# statistics epilogue of the logits GEMM with the transposing reduction: tests, then the learning step (compare with the committed build's line)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_reference_headline.py -m gpu -q -x -k "epilogue or split or hpd or G12 or kept_logits or headline" > gpurun_out/r5_t_test.log 2>&1; rc=$?
tail -n 3 gpurun_out/r5_t_test.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r5_t_test.log; then exit 1; fi
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --mode gngf_learning --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r5_t.json 2> gpurun_out/r5_t.err || { tail -3 gpurun_out/r5_t.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/r5_t.json').read().strip().splitlines()[-1]); print('learning', round(d['ms_per_step'],1), {k: round(v,1) for k,v in sorted(d['modes']['gngf_learning'].get('entry_ms',{}).items(), key=lambda kv: -kv[1])[:5]})"
done
timeout -k 10 200 python tools/perf_gemm_split.py 2>&1 | grep "split=1" | head -2
