#!/bin/bash
# A/B of library builds: bash tools/ab_libs.sh build/libA.so build/libB.so ...   (the default library runs first and last)
for lib in "" "$@" ""; do
  if [ -n "$lib" ]; then export GNGF_LIB_PATH=$PWD/$lib; else unset GNGF_LIB_PATH; fi
  python bench.py --no-extra-modes --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print(os.environ.get('GNGF_LIB_PATH', 'default')[-24:], round(d['ms_per_step'], 4), {a: round(b * 1e3, 1) for a, b in k.items()})"
done
