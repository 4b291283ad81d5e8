# learning-mode A/B (bench.py --set ...): ms/step and the per-entry event times (summed over both streams)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/abl
for spec in "$@"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 python bench.py --mode gngf_learning --no-extra-modes --no-cpu-baseline --steps 3 --warmup 1 $args > gpurun_out/abl/$tag.json 2> gpurun_out/abl/$tag.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/abl/$tag.json').read().strip().splitlines()[-1])
e=d['modes']['gngf_learning'].get('entry_ms') or {}
print('%-10s %.1f ms  ' % ('$tag', d['ms_per_step']), {k.replace('gngf_',''): round(v,1) for k,v in list(e.items())[:8]})
PY
done
