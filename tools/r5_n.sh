#!/bin/bash
# same-box A/B of two library builds on the cfg4 / cfg5 steps: build/libgngf_dev.so (the riders clear one row per vertex) vs the tree's
# (levels whose vertices are >= a third of the table's rows are cleared densely)
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in build/libgngf_dev.so ""; do
    for m in cfg4_hash cfg5_hash_fp16; do
      GNGF_LIB_PATH=$lib timeout -k 10 300 python bench.py --mode $m --no-extra-modes --no-cpu-baseline --no-full-outputs > gpurun_out/r5_n.json 2> gpurun_out/r5_n.err || { tail -3 gpurun_out/r5_n.err; exit 1; }
      python -c "
import json
d=json.loads(open('gpurun_out/r5_n.json').read().strip().splitlines()[-1]); print('$m lib=${lib:-tree}', round(d['ms_per_step'],4), [round(x,4) for x in d['ms_per_step_windows']], round(d['kernel_ms'].get('prepare(bin+vertex_fwd+clears)',0),4))"
    done
  done
done
