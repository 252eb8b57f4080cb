#!/bin/bash
# Runs ON THE GPU BOX: tuning sweep of the generic pool kernel on the two JOB shapes (knobs travel through
# polr_ctx_set_pool_tuning; bench.py reads them from POLR_POOL_* for sweeps like this one)
OUT=gpurun_out/sweep_generic_r03.txt
: > $OUT
run() {
	echo "== $*" >> $OUT
	env "$@" python3 bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-sub-records $EXTRA 2>> gpurun_out/sweep_generic_r03.err | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('   ms/step %.4f kernel_ms %.4f G tuples/s %.2f frac %.4f rounds %d E %d' % (d['ms_per_step'], r['kernel_ms_per_step'], d['value']/1e9, r['frac'], d['routing_rounds'], d['config']['executors_per_gpu']))
" >> $OUT
}
for WL in job_q18 job_light_01; do
	EXTRA=""
	for ux in 1 2 4; do for hu in 64 256; do
		run WLNAME=$WL POLR_POOL_UNITS_X=$ux POLR_POOL_HI_UNIT=$hu
	done; done
	for e in 8 16 64 128; do
		EXTRA="--executors $e"
		run WLNAME=$WL POLR_POOL_UNITS_X=1 POLR_POOL_HI_UNIT=64
	done
done
cat $OUT
