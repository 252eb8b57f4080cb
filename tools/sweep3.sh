run() { echo "== $ARGS $*"; env "$@" timeout -k 10 200 python bench.py $ARGS --no-sub-records --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('RESULT ms/step', r['ms_per_step'], 'Gt/s', round(r['value']/1e9,2), 'rounds', r['routing_rounds'], 'E', r['config']['executors_per_gpu'], 'frac', r['roofline']['frac'], 'B/t', r['roofline']['algorithmic_bytes_per_tuple'], 'inter', r['total_intermediates'], 'tpp', r['tuples_per_path'], 'gen_s', r['generate_s'], r['launch_info']['waves_per_workgroup'], r['launch_info']['workgroups_per_cu'], r['launch_info']['lds_tables'], r['config']['build_tables'], r['config']['join_orders'])
" || tail -3 gpurun_out/sweep_err.txt; }
ARGS="--scale 100 --steps 5 --warmup 1 --routing default_path --executors 1"; run A=1
ARGS="--scale 100 --steps 5 --warmup 1 --routing default_path --executors 64"; run A=1
ARGS="--scale 100 --steps 5 --warmup 1 --executors 64"; run A=1
ARGS="--scale 100 --steps 5 --warmup 1 --executors 128"; run A=1
ARGS="--scale 100 --steps 5 --warmup 1 --executors 256"; run A=1
ARGS="--scale 100 --steps 5 --warmup 1 --executors 128"; run POLR_POOL_HI_TUPLES=0
ARGS="--scale 100 --steps 5 --warmup 1 --executors 128 --routing init_once"; run A=1
