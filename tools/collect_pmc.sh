#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 PMC pass (counters in $PMC) of a bench.py configuration; prints the per-kernel sums
#   PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY" bash tools/collect_pmc.sh <tag> <bench.py args...>
TAG=$1
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $PMC --kernel-include-regex "polr_pool" --output-format csv -d $OUT/p -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-sub-records --no-kernel-events "$@" > $OUT/bench.json 2> $OUT/p.log || echo "rocprofv3 exit status $?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
files = glob.glob(out + "/p/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")[:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    if "pool" not in k:
        continue
    print("KERNEL", k)
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print("  %-28s total %.4g  per dispatch %.4g  (%d dispatches)" % (c, v, v / n, n))
PY
