ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_x_cfg4 -o p -- python3 $ROOT/bench.py --mode cfg4_hash --steps 100 --warmup 3 --no-extra-modes --no-cpu-baseline --no-full-outputs > $OUT/prof_x_cfg4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_x_cfg5 -o p -- python3 $ROOT/bench.py --mode cfg5_hash_fp16 --steps 60 --warmup 3 --no-extra-modes --no-cpu-baseline --no-full-outputs > $OUT/prof_x_cfg5.log 2>&1
rm -f $OUT/prof_x_cfg4/*kernel_trace.csv $OUT/prof_x_cfg5/*kernel_trace.csv
cd $ROOT
python - <<EOF2
import csv
for c in ("cfg4","cfg5"):
    print("==", c)
    for r in csv.DictReader(open(f"gpurun_out/prof_x_{c}/p_kernel_stats.csv")):
        if float(r["Percentage"]) > 0.3: print(r["Name"][:80].ljust(80), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:9.1f}', r["Percentage"])
EOF2
