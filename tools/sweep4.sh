run() { echo "== $ARGS $*"; env "$@" timeout -k 10 200 python bench.py $ARGS --no-sub-records --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'ev_ms/step', r['roofline']['ms_per_step_with_events'], 'kernel_ms', r['roofline']['kernel_ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'], 'frac', r['roofline']['frac'])
" || tail -3 gpurun_out/sweep_err.txt; }
for e in 96 128 160 192 256 512; do
ARGS="--scale 100 --steps 10 --warmup 3 --executors $e"; run A=1
done
ARGS="--scale 100 --steps 10 --warmup 3 --executors 128 --sync-every-step"; run A=1
