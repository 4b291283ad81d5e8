#!/bin/bash
# Profiles of the round (run on the GPU box through gpurun): kernel-trace statistics of the three bench modes and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE — each in its own run, with --kernel-trace only) of the headline mode.
# Usage: bash tools/profile_round.sh r02      -> gpurun_out/prof_<tag>_*/
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 rocprofv3 "$@" > $OUT/prof_${TAG}_$name.log 2>&1; echo "rc $?"; }
run frozen   --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_frozen -o p -- python3 $ROOT/bench.py --steps 300 --warmup 3 --no-extra-modes --no-cpu-baseline
run hash     --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_hash -o p -- python3 $ROOT/bench.py --mode hash --steps 300 --warmup 3 --no-extra-modes --no-cpu-baseline
run learning --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_learning -o p -- python3 $ROOT/bench.py --mode gngf_learning --steps 1 --warmup 1 --no-extra-modes --no-cpu-baseline
run cfg4     --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_cfg4 -o p -- python3 $ROOT/bench.py --mode cfg4_hash --steps 100 --warmup 3 --no-extra-modes --no-cpu-baseline --no-full-outputs
run cfg5     --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_cfg5 -o p -- python3 $ROOT/bench.py --mode cfg5_hash_fp16 --steps 60 --warmup 3 --no-extra-modes --no-cpu-baseline --no-full-outputs
run fetch    --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_pmc_fetch -o p -- python3 $ROOT/bench.py --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline
run write    --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_pmc_write -o p -- python3 $ROOT/bench.py --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline
# keep what travels back small: the per-dispatch traces of the long runs are not needed (the stats are)
run sq1      --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_${TAG}_pmc_sq1 -o p -- python3 $ROOT/bench.py --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline --no-full-outputs
run sq2      --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/prof_${TAG}_pmc_sq2 -o p -- python3 $ROOT/bench.py --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline --no-full-outputs
run fetch4   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_pmc_fetch_cfg4 -o p -- python3 $ROOT/bench.py --mode cfg4_hash --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline --no-full-outputs
run write4   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_pmc_write_cfg4 -o p -- python3 $ROOT/bench.py --mode cfg4_hash --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline --no-full-outputs
if [ "${ONLY_CFG5_PMC:-0}" != "0" ] || [ "${WITH_CFG5_PMC:-1}" != "0" ]; then
run fetch5   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_pmc_fetch_cfg5 -o p -- python3 $ROOT/bench.py --mode cfg5_hash_fp16 --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline --no-full-outputs
run write5   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_pmc_write_cfg5 -o p -- python3 $ROOT/bench.py --mode cfg5_hash_fp16 --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline --no-full-outputs
fi
rm -f $OUT/prof_${TAG}_frozen/*kernel_trace.csv $OUT/prof_${TAG}_hash/*kernel_trace.csv $OUT/prof_${TAG}_cfg4/*kernel_trace.csv $OUT/prof_${TAG}_cfg5/*kernel_trace.csv
du -sh $OUT/prof_${TAG}_* | cat
