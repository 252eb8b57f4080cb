#!/usr/bin/env python3
"""profiles/traffic_<name>.json from the PMC passes of tools/collect_traffic.sh <tag> ...:
   python tools/summarize_traffic.py <tag> <name>"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
base = os.path.join(ROOT, "gpurun_out", "traffic_" + tag)
line = json.load(open(os.path.join(base, "bench_fetch.json")))
kernel = line["roofline"]["kernel"]


def total(path, counter):
    t, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            t += float(r["Counter_Value"])
            n += 1
    return t * 1024.0, n


f, nf = total(os.path.join(base, "fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
w, nw = total(os.path.join(base, "write", "write_counter_collection.csv"), "WRITE_SIZE")
ms = line["roofline"]["kernel_ms_per_step"]
out = {
    "workload": name, "kernel": kernel, "config": line["config"], "launches_counted": nf,
    "fetch_bytes_per_launch_raw": f / nf, "write_bytes_per_launch": w / nw,
    "hbm_bytes_per_launch": f / nf + w / nw,
    "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_step"] / line["roofline"]["launches_per_step"],
    "kernel_ms_per_launch": ms / line["roofline"]["launches_per_step"],
    "raw_fetch_TBps": f / nf / (ms / line["roofline"]["launches_per_step"]) / 1e9,
    "probe_tuples_per_s": line["value"],
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/collect_traffic.sh), KB summed per "
            "dispatch of the kernel.  FETCH_SIZE on gfx950 counts 64-B fabric read requests and reports half of a "
            "wide coalesced stream (MI355X_MICROARCH.md); uncalibrated for this kernel's mix of 4-B gathers and 32-B "
            "slot-group reads, so the raw value is a lower bound of the bytes moved (true value between 1x and 2x).",
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_%s.json" % name), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("note", "config")}, indent=1))
