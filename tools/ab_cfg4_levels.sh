# A/B on one box: how many levels of the 4096^2 shape (and of the 8192^2 one) are staged, now that the direct levels' backward is bucketed
for c in 4.0 2.5 1.2 0.6; do
python bench.py --mode cfg4_hash --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline --no-full-outputs --set TILED_CELLS_PER_PIXEL=$c > gpurun_out/ab_c4_$c.json 2> gpurun_out/ab_c4_$c.err
python - <<EOF3
import json
try:
    d=json.loads(open("gpurun_out/ab_c4_$c.json").read().strip().splitlines()[-1])
    print("cfg4 cells/pixel $c:", d["ms_per_step"], d.get("ms_per_step_windows"))
except Exception as e:
    print("cfg4 $c failed", e); print(open("gpurun_out/ab_c4_$c.err").read()[-600:])
EOF3
done
for c in 4.0 1.0; do
python bench.py --mode cfg5_hash_fp16 --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline --no-full-outputs --set TILED_CELLS_PER_PIXEL=$c --set BUCKETED_MIN_DENSITY=0.2 > gpurun_out/ab_c5_$c.json 2> gpurun_out/ab_c5_$c.err
python - <<EOF3
import json
try:
    d=json.loads(open("gpurun_out/ab_c5_$c.json").read().strip().splitlines()[-1])
    print("cfg5 (bucketed forced) cells/pixel $c:", d["ms_per_step"], d.get("ms_per_step_windows"))
except Exception as e:
    print("cfg5 $c failed", e); print(open("gpurun_out/ab_c5_$c.err").read()[-600:])
EOF3
done
