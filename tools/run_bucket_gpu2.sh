python bench.py --mode cfg4_hash --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline --no-full-outputs > gpurun_out/b_cfg4.json 2> gpurun_out/b_cfg4.err; python - <<EOF3
import json
d=json.loads(open("gpurun_out/b_cfg4.json").read().strip().splitlines()[-1])
print("cfg4", d["ms_per_step"], d.get("ms_per_step_windows"))
EOF3
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bucket.py tests/test_gpu_encode.py -x -q > gpurun_out/t_fullsize.log 2>&1; tail -3 gpurun_out/t_fullsize.log
