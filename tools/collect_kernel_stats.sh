#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel statistics of an arbitrary bench.py configuration
#   bash tools/collect_kernel_stats.sh <tag> <bench.py args...>   -> gpurun_out/kstats_<tag>/
TAG=$1
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.log || echo "rocprofv3 exit status $?"
ls $OUT/trace/*kernel_stats.csv
