#!/bin/bash
export POLR_POOL_UNITS_X=2
for M in 0 512; do
export POLR_DIAG_TIMELINE=gpurun_out/tl_e512_m$M.npz
timeout -k 10 200 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-sub-records --executors 512 --morsels $M > gpurun_out/tl_e512_m$M.json || exit 1
echo "== E=512 X=2 morsels $M"
python tools/diag_timeline.py gpurun_out/tl_e512_m$M.npz
done
