# same-box A/B of ops switches on the headline step (bench.py --set): prints ms/step, five more windows, and the decoder's own
# in-graph span (it moves by +-5 us from run to run with the chip's clocks: compare ms/step MINUS that span)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-modes --no-full-outputs --steps 200 --warmup 10 "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || return 1; python - <<PY
import json
d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1])
sp=d['roofline'].get('in_graph_launch_ms') or 0
print('%-14s %.4f  minus decoder span %.1f us   windows %s   decoder %.1f' % ('$tag', d['ms_per_step'], (d['ms_per_step']-sp)*1e3, ['%.4f'%w for w in d['ms_per_step_windows']], sp*1e3))
PY
}
for spec in "$@"; do
  tag=${spec%%:*}; args=${spec#*:}
  run $tag $args || exit 1
done
