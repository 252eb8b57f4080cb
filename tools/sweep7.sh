run() { echo "== $ARGS $*"; env "$@" timeout -k 10 200 python bench.py $ARGS --no-sub-records --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'], 'frac', r['roofline']['frac'])
" || tail -3 gpurun_out/sweep_err.txt; }
ARGS="--scale 100 --steps 10 --warmup 3 --executors 256"
run POLR_POOL_UNITS_X=1
run POLR_POOL_UNITS_X=2
run POLR_POOL_UNITS_X=4
ARGS="--scale 100 --steps 10 --warmup 3 --executors 128"
run POLR_POOL_UNITS_X=1
ARGS="--scale 100 --steps 10 --warmup 3 --executors 256 --routing init_once"
run POLR_POOL_UNITS_X=1
run POLR_POOL_UNITS_X=4
ARGS="--scale 100 --steps 10 --warmup 3 --executors 1 --routing default_path --pin-path 3"
run POLR_POOL_UNITS_X=1
run POLR_POOL_UNITS_X=2
