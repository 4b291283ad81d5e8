set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-modes --no-full-outputs --steps 200 --warmup 10 "$@" > gpurun_out/r4b/$tag.json 2> gpurun_out/r4b/$tag.err || return 1; python - <<PY
import json
d=json.loads(open('gpurun_out/r4b/$tag.json').read().strip().splitlines()[-1])
print('$tag', d['ms_per_step'], d['ms_per_step_windows'], d['roofline'].get('in_graph_launch_ms'))
PY
}
run hash_base --mode hash && run hash_nopipe --mode hash --set BIN_PIPELINE=0 && run hash_192 --mode hash --set BIN_BLOCKS_MAX=192 --set BIN_PIXELS_PER_BLOCK=4096 && run hash_256 --mode hash --set BIN_BLOCKS_MAX=256 --set BIN_PIXELS_PER_BLOCK=4096 && run hash_160 --mode hash --set BIN_BLOCKS_MAX=160 --set BIN_PIXELS_PER_BLOCK=4096 && run gngf_base && run gngf_192 --set BIN_BLOCKS_MAX=192 --set BIN_PIXELS_PER_BLOCK=4096
