#!/bin/bash
# A/B of the pixel-stage workgroup sizes: build/libgngf_tb_<fwd>_<bwd>.so variants vs the default library
for lib in "" build/libgngf_tb_512_256.so build/libgngf_tb_1024_256.so build/libgngf_tb_256_512.so build/libgngf_tb_256_1024.so build/libgngf_tb_512_512.so; do
  if [ -n "$lib" ]; then export GNGF_LIB_PATH=$PWD/$lib; else unset GNGF_LIB_PATH; fi
  python bench.py --no-extra-modes --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print(os.environ.get('GNGF_LIB_PATH', 'default')[-24:], round(d['ms_per_step'], 4), 'fwd', round(k['encode_fwd:tiled'] * 1e3, 1), 'bwd', round(k['encode_bwd:tiled'] * 1e3, 1))"
done
