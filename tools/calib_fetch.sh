#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE of the calibration kernels (tools/micro/fetch_calib.hip) against their known bytes.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "stream 0" "gather 5" "gather 12" "gather 40"; do
  set -- $spec
  tag=$1_$2
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "calib_(stream|gather)" --output-format csv -d $OUT/$tag -o c -- $ROOT/tools/micro/fetch_calib $1 $2 > $OUT/$tag.json 2> $OUT/$tag.log || echo "$tag exit $?"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = []
for f in sorted(glob.glob(out + "/*.json")):
    tag = f.split("/")[-1][:-5]
    try:
        known = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(tag, "no output", e); continue
    vals = []
    for c in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(c)):
            if row["Counter_Name"] == "FETCH_SIZE" and known["kernel"] in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        print(tag, "no counter rows"); continue
    fetch = sum(vals) / len(vals) * 1024  # KiB -> bytes
    known["fetch_size_bytes_per_launch"] = fetch
    known["fetch_over_known"] = fetch / known["known_bytes_per_launch"]
    res.append(known)
    print(json.dumps(known))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
