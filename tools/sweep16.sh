#!/bin/bash
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records "$@" > gpurun_out/s16.json || exit 1
python - "$POLR_POOL_UNITS_X $*" <<PY
import json,sys
d=json.loads(open("gpurun_out/s16.json").read().strip().splitlines()[-1])
print("units_x", sys.argv[1], "| ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "rounds", d.get("routing_rounds"))
PY
}
for X in 1 2 3; do
export POLR_POOL_UNITS_X=$X
run --executors 256
run --executors 512
run --executors 768
done
