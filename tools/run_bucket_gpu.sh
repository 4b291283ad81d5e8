python -m pytest tests/test_gpu_bucket.py -q > gpurun_out/t_bucket.log 2>&1; grep -E "^E  .*Error|passed|failed" gpurun_out/t_bucket.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp && IMAGES=${IMAGES:-65536} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bucket -o pb -- python3 $GRAFT_REPO_ROOT/tools/perf_bucket.py > $GRAFT_REPO_ROOT/gpurun_out/perf_bucket2.log 2>&1
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof_bucket/*kernel_trace.csv gpurun_out/prof_bucket/*.db
python - <<EOF2
import csv,glob
for f in glob.glob("gpurun_out/prof_bucket/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gngf" in r["Name"]: print(r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"], r["Percentage"])
EOF2
grep -E "us$|difference" gpurun_out/perf_bucket2.log
