run() { echo "== $ARGS $*"; env "$@" timeout -k 10 200 python bench.py $ARGS --no-sub-records --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'], 'inter', r['total_intermediates'], 'count', r['count_star'])
" || tail -3 gpurun_out/sweep_err.txt; }
for r in opportunistic dynamic exponential_backoff; do for e in 256 512 1024; do
ARGS="--workload job_light_01 --steps 20 --warmup 3 --executors $e --routing $r"; run A=1
done; done
ARGS="--workload job_light_01 --steps 20 --warmup 3 --executors 16"; run A=1
ARGS="--workload job_q18 --steps 20 --warmup 3 --executors 64"; run A=1
ARGS="--workload job_q18 --steps 20 --warmup 3 --executors 16"; run A=1
