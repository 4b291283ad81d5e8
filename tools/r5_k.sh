#!/bin/bash
# cfg4: 14 staged levels on the generic kernels (default) vs 12 on the level-interleaved ones + 4 direct (TILED_CELLS_PER_PIXEL=1)
mkdir -p gpurun_out
for v in 4 1; do
  timeout -k 10 300 python bench.py --mode cfg4_hash --no-extra-modes --no-cpu-baseline --no-full-outputs --set TILED_CELLS_PER_PIXEL=$v > gpurun_out/r5_k_$v.json 2> gpurun_out/r5_k_$v.err || { tail -5 gpurun_out/r5_k_$v.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/r5_k_$v.json').read().strip().splitlines()[-1]); print('cfg4 cells_per_pixel=$v', d['ms_per_step'], d['ms_per_step_windows'][:3], d['config'].get('step_config'), {k: round(x,3) for k,x in d['kernel_ms'].items()})"
done
