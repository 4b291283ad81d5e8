for im in 65536 131072; do
python bench.py --mode cfg5_hash_fp16 --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline --no-full-outputs --set BUCKET_IMAGE_BYTES=$im > gpurun_out/ab_c5_im$im.json 2> gpurun_out/ab_c5_im$im.err
python - <<EOF3
import json
try:
    d=json.loads(open("gpurun_out/ab_c5_im$im.json").read().strip().splitlines()[-1])
    print("cfg5 image $im:", d["ms_per_step"], d.get("ms_per_step_windows"))
except Exception as e:
    print("cfg5 $im failed", e); print(open("gpurun_out/ab_c5_im$im.err").read()[-600:])
EOF3
done
python bench.py --mode cfg5_hash_fp16 --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline --no-full-outputs --set BUCKETED_DIRECT_BWD=0 > gpurun_out/ab_c5_off.json 2> gpurun_out/ab_c5_off.err
python - <<EOF3
import json
d=json.loads(open("gpurun_out/ab_c5_off.json").read().strip().splitlines()[-1])
print("cfg5 atomics:", d["ms_per_step"], d.get("ms_per_step_windows"))
EOF3
python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
