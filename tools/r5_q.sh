#!/bin/bash
# learning mode, same box, alternating: per-wave split (17) | three bf16 planes in LDS (1) | planes + two-plane dW / dh
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_reference_headline.py -m gpu -q -x -k "split or hpd or lowrank or G12 or kept_logits or headline or epilogue" > gpurun_out/r5_q_test.log 2>&1; rc=$?
tail -n 3 gpurun_out/r5_q_test.log
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2; do
  for cfg in "17 0" "1 0" "1 1"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --mode gngf_learning --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 --set HPD_GEMM_KERNEL=$1 --set HPD_BWD_TWO_PLANES=$2 > gpurun_out/r5_q.json 2> gpurun_out/r5_q.err || { tail -3 gpurun_out/r5_q.err; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/r5_q.json').read().strip().splitlines()[-1]); print('learning kernel=$1 two_planes=$2', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['modes']['gngf_learning'].get('entry_ms',{}).items() if 'linear' in k or 'gemm' in k})"
  done
done
