#!/bin/bash
# Runs ON THE GPU BOX: the round-2 profile set of the default bench.py run (SSB-skew Q4.1, SF100, adaptive_reinit,
# 256 executors): kernel trace + stats, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) restricted to the pool kernel.
#   bash tools/collect_profiles_r02.sh
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-sub-records"
echo "[1/4] kernel trace"; rocprofv3 --kernel-trace --stats --kernel-include-regex "polr_" --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.log || echo "trace exit $?"
echo "[2/4] FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "polr_pool" --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_fetch.json 2> $OUT/fetch.log || echo "fetch exit $?"
echo "[3/4] WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "polr_pool" --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_write.json 2> $OUT/write.log || echo "write exit $?"
echo "[4/4] SQ"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-include-regex "polr_pool" --output-format csv -d $OUT/sq -o s -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_sq.json 2> $OUT/sq.log || echo "sq exit $?"
find $OUT -name "*.csv" | head -20
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for tag in ("fetch", "write", "sq"):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "pool" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
    for c, v in acc.items():
        res[c] = {"per_dispatch": v / cnt[c], "dispatches": cnt[c]}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    res["kernel_stats"] = [r for r in csv.DictReader(open(f))]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernel_stats"}, indent=1))
for r in res.get("kernel_stats", []):
    print(r)
PY
