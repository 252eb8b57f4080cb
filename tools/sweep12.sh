run() { echo "== $ARGS $*"; env "$@" timeout -k 10 200 python bench.py $ARGS --no-sub-records --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'], 'count', r['count_star'])
" || tail -3 gpurun_out/sweep_err.txt; }
for sh in 1 2 4 8; do
ARGS="--workload job_light_01 --steps 20 --warmup 3 --executors 256 --routing opportunistic"; run POLR_POOL_SHARE=$sh
done
for sh in 2 4; do
ARGS="--workload job_light_01 --steps 20 --warmup 3 --executors 32"; run POLR_POOL_SHARE=$sh
ARGS="--workload job_q18 --steps 20 --warmup 3 --executors 32"; run POLR_POOL_SHARE=$sh
done
