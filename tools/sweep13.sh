run() { echo "== $ARGS $*"; env "$@" timeout -k 10 200 python bench.py $ARGS --no-sub-records --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'], 'frac', r['roofline']['frac'])
" || tail -3 gpurun_out/sweep_err.txt; }
for e in 256 384 512; do for ux in 2 4; do
ARGS="--scale 100 --steps 10 --warmup 3 --executors $e"; run POLR_POOL_UNITS_X=$ux
done; done
ARGS="--scale 100 --steps 10 --warmup 3 --executors 256 --workload ssb_skew_q42"; run A=1
ARGS="--scale 100 --steps 10 --warmup 3 --executors 256 --workload ssb_skew_q43"; run A=1
ARGS="--scale 100 --steps 10 --warmup 3 --executors 256 --workload ssb_skew_q31"; run A=1
