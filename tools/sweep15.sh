#!/bin/bash
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records "$@" > gpurun_out/s15.json || exit 1
python - "$*" <<PY
import json,sys
d=json.loads(open("gpurun_out/s15.json").read().strip().splitlines()[-1])
print(sys.argv[1], "| ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "rounds", d.get("routing_rounds"))
PY
}
run --morsels 128
run --morsels 512
run --morsels 2048
run --executors 128
run --executors 512
run --executors 512 --morsels 512
run --executors 128 --morsels 512
