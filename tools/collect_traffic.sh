#!/bin/bash
# Runs ON THE GPU BOX: the two PMC passes (FETCH_SIZE, WRITE_SIZE) of an arbitrary bench.py configuration
#   bash tools/collect_traffic.sh <tag> <bench.py args...>
set -e
TAG=$1
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.log || echo "rocprofv3 fetch pass: exit status $?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_write.json 2> $OUT/write.log || echo "rocprofv3 write pass: exit status $?"
ls $OUT/fetch/*counter_collection.csv $OUT/write/*counter_collection.csv
