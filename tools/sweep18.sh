#!/bin/bash
run() { timeout -k 10 200 python bench.py --workload job_light_01 --steps 50 --warmup 5 --no-cpu-baseline --no-sub-records "$@" > gpurun_out/s18.json || exit 1
python - "$*" <<PY
import json,sys
d=json.loads(open("gpurun_out/s18.json").read().strip().splitlines()[-1])
print(sys.argv[1], "| ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_per_step"], "G tuples/s", round(d["value"]/1e9,2), "rounds", d.get("routing_rounds"))
PY
}
for E in 8 32 128; do run --executors $E; done
for R in opportunistic dynamic; do for E in 64 256 640 1274; do run --routing $R --executors $E; done; done
export POLR_DIAG_TIMELINE=gpurun_out/tl_jl.npz
timeout -k 10 200 python bench.py --workload job_light_01 --steps 5 --warmup 3 --no-cpu-baseline --no-sub-records > gpurun_out/tl_jl.json && python tools/diag_timeline.py gpurun_out/tl_jl.npz
