#!/bin/bash
# Runs ON THE GPU BOX: the round-3 profile set of ONE bench.py command -- kernel trace + stats, then the PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ) in runs of their own, restricted to the pool kernels.
#   bash tools/collect_profiles_r03.sh <tag> <bench.py arguments...>
# e.g. bash tools/collect_profiles_r03.sh job_q18 --workload job_q18 --steps 20 --warmup 3
# (the program itself follows `--`: python3 bench.py, no wrapper in between)
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r03_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$* --no-cpu-baseline --no-sub-records"
echo "[1/7] kernel trace: $ARGS"
rocprofv3 --kernel-trace --stats --kernel-include-regex "polr_" --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.log || echo "trace exit $?"
echo "[2/7] FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "polr_pool" --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_fetch.json 2> $OUT/fetch.log || echo "fetch exit $?"
echo "[3/7] WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "polr_pool" --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_write.json 2> $OUT/write.log || echo "write exit $?"
echo "[4/7] SQ"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-include-regex "polr_pool" --output-format csv -d $OUT/sq -o s -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_sq.json 2> $OUT/sq.log || echo "sq exit $?"
echo "[5/7] SQ, second set"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_IFETCH --kernel-include-regex "polr_pool" --output-format csv -d $OUT/sq2 -o s -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_sq2.json 2> $OUT/sq2.log || echo "sq2 exit $?"
echo "[6/7] TA / TCP"
rocprofv3 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum --kernel-include-regex "polr_pool" --output-format csv -d $OUT/tcp -o s -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_tcp.json 2> $OUT/tcp.log || echo "tcp exit $?"
echo "[7/7] TCC"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-include-regex "polr_pool" --output-format csv -d $OUT/tcc -o s -- python3 $ROOT/bench.py $ARGS --no-kernel-events > $OUT/bench_under_tcc.json 2> $OUT/tcc.log || echo "tcc exit $?"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]
res = {"tag": tag}
for sub in ("fetch", "write", "sq", "sq2", "tcp", "tcc"):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "pool" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
    for c, v in acc.items():
        res[c] = {"per_dispatch": v / cnt[c], "dispatches": cnt[c]}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    res["kernel_stats"] = [r for r in csv.DictReader(open(f))]
try:
    res["bench_line"] = json.loads(open(out + "/bench_under_trace.json").read().strip().splitlines()[-1])
except Exception as e:
    res["bench_line_error"] = str(e)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("kernel_stats", "bench_line")}, indent=1))
for r in res.get("kernel_stats", []):
    print(r)
PY
