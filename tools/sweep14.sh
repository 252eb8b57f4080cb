#!/bin/bash
# morsel sizes with the timeline diagnostic
for M in 0 64 256 1024; do
  export POLR_DIAG_TIMELINE=gpurun_out/tl_m$M.npz
  timeout -k 10 200 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-sub-records --morsels $M > gpurun_out/tl_m$M.json || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/tl_m$M.json").read().strip().splitlines()[-1])
print("morsels $M ms/step", d["ms_per_step"], "rounds", d.get("routing_rounds"), "frac", d["roofline"]["frac"])
PY
  python tools/diag_timeline.py gpurun_out/tl_m$M.npz | grep -v "^path\|units of"
done
