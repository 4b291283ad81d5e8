#!/bin/bash
# learning mode, same box: forward pass pipelined over two streams (logits GEMM beside the p-bar pass) or not
mkdir -p gpurun_out
for rep in 1 2; do
  for f in 1 0; do
    timeout -k 10 300 python bench.py --mode gngf_learning --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 --set HPD_PIPELINE_FWD=$f > gpurun_out/r5_s.json 2> gpurun_out/r5_s.err || { tail -3 gpurun_out/r5_s.err; exit 1; }
    if grep -q "Memory access fault" gpurun_out/r5_s.err; then tail -3 gpurun_out/r5_s.err; exit 1; fi
    python -c "
import json
d=json.loads(open('gpurun_out/r5_s.json').read().strip().splitlines()[-1]); print('learning pipeline_fwd=$f', round(d['ms_per_step'],1), {k: round(v,1) for k,v in sorted(d['modes']['gngf_learning'].get('entry_ms',{}).items(), key=lambda kv: -kv[1])[:7]})"
  done
done
