# A/B of ops switches on any bench mode: bash tools/ab_mode.sh <mode> "tag:--set X=0" ...
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/abm
MODE=$1; shift
for spec in "$@"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 python bench.py --mode $MODE --no-extra-modes --no-cpu-baseline --no-full-outputs --steps 40 --warmup 5 $args > gpurun_out/abm/$tag.json 2> gpurun_out/abm/$tag.err || { tail -3 gpurun_out/abm/$tag.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/abm/$tag.json').read().strip().splitlines()[-1])
print('%-10s %.4f ms  windows %s' % ('$tag', d['ms_per_step'], ['%.4f'%w for w in (d.get('ms_per_step_windows') or [])]), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items() if isinstance(v,float)})
PY
done
