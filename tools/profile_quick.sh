#!/bin/bash
# kernel-trace statistics of one bench mode (quick look between optimisation steps): bash tools/profile_quick.sh <mode> <tag>
set -u
MODE=${1:-gngf_frozen}; TAG=${2:-q}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$MODE -o p -- python3 $ROOT/bench.py --mode $MODE --steps 200 --warmup 3 --no-extra-modes --no-cpu-baseline --no-full-outputs > $OUT/prof_${TAG}_$MODE.log 2>&1
rm -f $OUT/prof_${TAG}_$MODE/*kernel_trace.csv
python3 $ROOT/tools/prof_summary.py $OUT/prof_${TAG}_$MODE 16
tail -c 400 $OUT/prof_${TAG}_$MODE.log | grep -o '"ms_per_step": [0-9.]*' | head -1
