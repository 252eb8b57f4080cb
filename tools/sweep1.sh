set -x
for args in "--workload job_light_01 --executors 8" "--workload job_light_01 --executors 32" "--scale 5 --routing default_path --executors 1" "--scale 5 --routing default_path --executors 32" "--scale 5 --executors 8" "--scale 5 --executors 128" "--scale 5 --routing init_once --executors 32"; do
  timeout -k 10 120 python bench.py $args --steps 10 --warmup 2 --no-sub-records --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT', '$args', 'ms/step', r['ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'kernel_ms', r['roofline']['kernel_ms_per_step'], 'rounds', r['routing_rounds'], 'frac', r['roofline']['frac'], 'E', r['config']['executors_per_gpu'], r['launch_info'])
"
done
