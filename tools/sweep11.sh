run() { echo "== $ARGS $*"; env "$@" timeout -k 10 200 python bench.py $ARGS --no-sub-records --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'], 'frac', r['roofline']['frac'], 'count', r['count_star'])
" || tail -3 gpurun_out/sweep_err.txt; }
for lot in 1 2 8; do
ARGS="--workload job_light_01 --steps 20 --warmup 3 --executors 256 --routing opportunistic"; run POLR_POOL_HI_LOTTERY=$lot
ARGS="--workload job_light_01 --steps 20 --warmup 3 --executors 32"; run POLR_POOL_HI_LOTTERY=$lot
ARGS="--scale 100 --steps 10 --warmup 3 --executors 256"; run POLR_POOL_HI_LOTTERY=$lot
done
ARGS="--workload job_light_01 --steps 20 --warmup 3 --executors 512 --routing dynamic"; run A=1
