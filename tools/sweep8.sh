run() { echo "== $ARGS $*"; env "$@" timeout -k 10 200 python bench.py $ARGS --no-sub-records --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'], 'frac', r['roofline']['frac'], 'B/t', r['roofline']['algorithmic_bytes_per_tuple'], 'inter', r['total_intermediates'], 'count', r['count_star'])
" || tail -3 gpurun_out/sweep_err.txt; }
for m in 120 480 1920; do for e in 64 256; do
ARGS="--scale 100 --steps 10 --warmup 3 --executors $e --morsels $m"; run A=1
done; done
ARGS="--scale 100 --steps 10 --warmup 3 --executors 256"; run A=1
