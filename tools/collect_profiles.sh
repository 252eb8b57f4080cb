#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace statistics and the two PMC passes (FETCH_SIZE,
# WRITE_SIZE -- separate passes, never together with a trace domain) of the default bench.py command.
# Outputs under gpurun_out/prof_<tag>/; tools/summarize_profiles.py turns them into profiles/*.
set -e
# (rocprofv3 on this image can crash in its exit handler AFTER writing its csv files: tolerate that, the
# presence of the csv files is checked at the end)
TAG=${1:-r01}
shift || true
ARGS="$@"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.log || echo "rocprofv3 trace pass: exit status $?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log || echo "rocprofv3 fetch pass: exit status $?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events $ARGS > $OUT/bench_write.json 2> $OUT/write.log || echo "rocprofv3 write pass: exit status $?"
ls $OUT/trace/*kernel_stats.csv $OUT/fetch/*counter_collection.csv $OUT/write/*counter_collection.csv
