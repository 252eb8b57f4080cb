#!/bin/bash
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records "$@" > gpurun_out/s19.json || exit 1
python - "$POLR_POOL_UNITS_X $*" <<PY
import json,sys
d=json.loads(open("gpurun_out/s19.json").read().strip().splitlines()[-1])
print("units_x", sys.argv[1], "| ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "rounds", d.get("routing_rounds"))
PY
}
for W in ssb_skew_q41 ssb_skew_q42 ssb_skew_q43 ssb_skew_q31 ssb_skew_q21; do
for X in 2 4; do
export POLR_POOL_UNITS_X=$X
run --workload $W --executors 256
run --workload $W --executors 512
done
done
