bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1; grep -c "rc 0" gpurun_out/profile_round_r04.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err; tail -c 300 gpurun_out/r4_bench_final.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --no-extra-modes --no-cpu-baseline > gpurun_out/r4_g2_bench.json 2> gpurun_out/r4_g2_bench.err; tail -c 200 gpurun_out/r4_g2_bench.json
for v in 0 1; do python bench.py --mode cfg4_hash --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline --no-full-outputs --set PERSISTENT_TABLE_GRAD=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 persistent=$v', d['ms_per_step'], d['ms_per_step_windows'])"; done
