#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_big_shapes.py tests/test_gpu_bucket.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r5_o_test.log 2>&1; rc=$?
tail -n 3 gpurun_out/r5_o_test.log
if [ $rc -ne 0 ]; then exit $rc; fi
tools/r5_n.sh
