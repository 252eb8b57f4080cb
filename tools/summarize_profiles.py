#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the committed
summaries under profiles/: kernel statistics, the bench line printed under the profiler and the
HBM traffic of the dominant kernel per step (FETCH_SIZE / WRITE_SIZE, separate PMC passes).

  python tools/summarize_profiles.py r01_resident [workload]
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "job_light_01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
line = json.load(open(os.path.join(src, "bench_under_rocprof.json")))
kernel = line["roofline"]["kernel"]


def pmc(path, counter, steps_line):
    total, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
            total += float(row["Counter_Value"])
            n += 1
    steps = steps_line["steps"] + steps_line["warmup"]
    return total * 1024.0 / steps, n, steps  # the counters are in KB


fetch_line = json.load(open(os.path.join(src, "bench_fetch.json")))
write_line = json.load(open(os.path.join(src, "bench_write.json")))
fetch, n_f, steps_f = pmc(os.path.join(src, "fetch", "fetch_counter_collection.csv"), "FETCH_SIZE", fetch_line)
write, n_w, steps_w = pmc(os.path.join(src, "write", "write_counter_collection.csv"), "WRITE_SIZE", write_line)
launches = n_f / steps_f
out = {
    "workload": workload,
    "command": "tools/collect_profiles.sh %s: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, no trace "
               "domains) -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events" % tag,
    "kernel": kernel,
    "config": line["config"],
    "dispatches_counted": n_f,
    "steps_counted": steps_f,
    "launches_per_step": launches,
    "fetch_bytes_per_step_raw": fetch,
    "write_bytes_per_step": write,
    "hbm_bytes_per_step": fetch + write,
    "hbm_bytes_per_launch": (fetch + write) / launches,
    "algorithmic_bytes_per_step": line["roofline"]["algorithmic_bytes_per_step"],
    "note": "FETCH_SIZE/WRITE_SIZE (KB) summed over every dispatch of the kernel, divided by the steps of the run. "
            "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads exactly half of a wide (16 B/lane) coalesced stream; "
            "this kernel's reads are 4-B selection/key gathers and 32-B slot-group reads, for which the counter is "
            "uncalibrated, so the raw value is reported uncorrected (the true read traffic lies between 1x and 2x of it). "
            "The random 32-B slot-group probes move whole 64/128-B lines: over-fetch relative to the algorithmic "
            "bytes is inherent to hash probing, not a re-read.",
}
json.dump(out, open(os.path.join(dst, "traffic_%s.json" % workload), "w"), indent=1)
print(json.dumps(out, indent=1))
