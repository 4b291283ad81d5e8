#!/bin/bash
# same box, two library builds alternating (GNGF_LIB_PATH): the statistics epilogue of the logits GEMM with the plain 32-lane reductions
# (build/libgngf_base.so = the previous commit's linear.hip) against the transposing one
mkdir -p gpurun_out
for rep in 1 2 3; do
  for lib in build/libgngf_base.so ""; do
    GNGF_LIB_PATH=$lib timeout -k 10 300 python bench.py --mode gngf_learning --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r5_u.json 2> gpurun_out/r5_u.err || { tail -3 gpurun_out/r5_u.err; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/r5_u.json').read().strip().splitlines()[-1]); print('learning', '${lib:-in-tree (transposing)}', round(d['ms_per_step'],1), {k: round(v,1) for k,v in sorted(d['modes']['gngf_learning'].get('entry_ms',{}).items(), key=lambda kv: -kv[1])[:3]})"
  done
done
