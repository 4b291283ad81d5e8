#!/bin/bash
# usage: tools/run_tests_then_bench.sh TAG "pytest args" ["bench args"]
# GPU tests (output to gpurun_out/TAG_test.log), then — unless the tests were killed by the timeout — one bench.py run.
TAG=$1; TESTS=$2; BENCH=${3:-}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest $TESTS -m gpu -q -x --timeout=900 > gpurun_out/${TAG}_test.log 2>&1
rc=$?
tail -n 15 gpurun_out/${TAG}_test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed by timeout: no further GPU step"; exit $rc; fi
timeout -k 10 600 python bench.py $BENCH > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rb=$?
tail -c 600 gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], {m: (r.get("ms_per_step"), (r.get("roofline_survey") or {}).get("frac_of_8TBs")) for m, r in d["modes"].items()})
except Exception as e:
    print("no bench line:", e)
PY
exit $(( rc != 0 ? rc : rb ))
