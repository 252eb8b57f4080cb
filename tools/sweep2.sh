run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py $ARGS --steps 10 --warmup 2 --no-sub-records --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'])
"; }
ARGS="--scale 5 --routing init_once --executors 32"
run A=1
run POLR_POOL_HI_TUPLES=0
run POLR_POOL_SHARE=4
run POLR_POOL_SHARE=16
run POLR_POOL_SHARE=16 POLR_POOL_HI_TUPLES=0
ARGS="--scale 5 --routing init_once --executors 1"
run A=1
run POLR_POOL_SHARE=16
ARGS="--workload job_light_01 --executors 8"
run A=1
run POLR_POOL_SHARE=8
ARGS="--workload job_light_01 --executors 1"
run A=1
run POLR_POOL_SHARE=16
