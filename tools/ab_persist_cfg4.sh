# same-box A/B, alternating: the 4096^2 shape with the dense clear hidden in the training decoder (default) vs the step-to-step buffer
for rep in 1 2 3; do for v in 4294967296 0; do python bench.py --mode cfg4_hash --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline --no-full-outputs --set PERSISTENT_MIN_BYTES=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 PERSISTENT_MIN_BYTES=$v', round(d['ms_per_step'],4), [round(x,4) for x in d['ms_per_step_windows']])"; done; done
