#!/bin/bash
# kernel-trace statistics of one learning-mode step (the only profile that changes when the HPD kernels do); then the counters of the fused kernels
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_learning -o p -- python3 $ROOT/bench.py --mode gngf_learning --steps 1 --warmup 1 --no-extra-modes --no-cpu-baseline > $OUT/prof_${TAG}_learning.log 2>&1 || exit 1
if grep -q "Memory access fault" $OUT/prof_${TAG}_learning.log; then exit 1; fi
rm -f $OUT/prof_${TAG}_learning/*kernel_trace.csv
cd $ROOT && tools/pmc_hpd_bwd_fused.sh
