TAG=r04; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 rocprofv3 "$@" > $OUT/prof_${TAG}_$name.log 2>&1; echo "rc $?"; }
run fetch5   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_pmc_fetch_cfg5 -o p -- python3 $ROOT/bench.py --mode cfg5_hash_fp16 --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline --no-full-outputs
run write5   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_pmc_write_cfg5 -o p -- python3 $ROOT/bench.py --mode cfg5_hash_fp16 --steps 8 --warmup 1 --ramp-steps 4 --no-extra-modes --no-cpu-baseline --no-full-outputs
du -sh $OUT/prof_${TAG}_pmc_*cfg5
