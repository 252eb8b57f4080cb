#!/usr/bin/env python3
"""Copies what tools/collect_profiles_r03.sh left under gpurun_out/ (scratch) into profiles/ (tracked):

    profiles/r03_<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of the bench command
    profiles/r03_<tag>_pmc_pool.csv          the counter rows of the pool kernel's dispatches, all PMC passes
    profiles/r03_<tag>_pmc_summary.json      per-dispatch means of every counter, the bench line printed under the
                                             trace, the workload signature bench.py matches a replayed
                                             `roofline.traffic` against, and the corrected HBM traffic

    python tools/publish_profiles_r03.py [tag ...]        (default: every gpurun_out/prof_r03_* directory)
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles")

# gfx950 tallies a 128-byte fabric request as 64 bytes (MI355X_MICROARCH.md, HBM section; calibrated on the flat
# pipeline's own access patterns in profiles/r02_fetch_size_calibration.json; TCC_EA0_RDREQ x 128 B of the same runs
# agrees with 2 x FETCH_SIZE to 0.1 %)
FETCH_CORRECTION = 2.0


def signature(line):
    """The key bench.py looks a replayed `roofline.traffic` up by (bench.py: `pmc_signature`)."""
    roof = line.get("roofline") or {}
    if roof.get("pmc_signature"):
        return roof["pmc_signature"]
    cfg = line.get("config", {})
    desc = cfg.get("workload", "")
    name = re.split(r"[ :]", desc, 1)[0]
    m = re.search(r"at SF([0-9.]+)", desc) or re.search(r"cardinalities x([0-9.]+)", desc)
    sig = {"workload": name, "scale": float(m.group(1)) if m else None, "routing": cfg.get("routing"),
           "join_enumerator": cfg.get("join_enumerator"), "max_join_orders": cfg.get("max_join_orders")}
    if "executors_per_gpu" in cfg:
        sig["executors_per_gpu"] = cfg["executors_per_gpu"]
    elif "executors_per_pipeline" in cfg:  # (job_full)
        sig["executors_per_gpu"] = cfg["executors_per_pipeline"]
    sig["n_gpus"] = line.get("n_gpus", 1)
    return sig


def publish(tag):
    src = os.path.join(SRC, "prof_r03_" + tag)
    summary = json.load(open(os.path.join(src, "summary.json")))
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(DST, "r03_%s_kernel_stats.csv" % tag))
    rows = []
    for sub in ("fetch", "write", "sq", "sq2", "tcp", "tcc"):
        for f in sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                if "polr_pool" in r.get("Kernel_Name", ""):
                    rows.append({"pass": sub, "dispatch": r.get("Dispatch_Id"), "kernel": r["Kernel_Name"].split("(")[0],
                                 "grid": r.get("Grid_Size"), "workgroup": r.get("Workgroup_Size"),
                                 "lds_bytes": r.get("LDS_Block_Size"), "vgprs": r.get("VGPR_Count"),
                                 "counter": r["Counter_Name"], "value": r["Counter_Value"]})
    if rows:
        with open(os.path.join(DST, "r03_%s_pmc_pool.csv" % tag), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    out = {k: v for k, v in summary.items() if k not in ("kernel_stats",)}
    out["polr_kernels"] = [r for r in summary.get("kernel_stats", []) if "polr_" in r.get("Name", "")]
    line = summary.get("bench_line", {})
    out["workload_signature"] = signature(line)
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        fetch = summary["FETCH_SIZE"]["per_dispatch"] * 1024.0
        write = summary["WRITE_SIZE"]["per_dispatch"] * 1024.0
        launches = float(line.get("roofline", {}).get("launches_per_step", 1.0) or 1.0)
        out["fetch_size_correction"] = FETCH_CORRECTION
        out["launches_per_step"] = launches
        out["traffic_bytes_per_launch_corrected"] = int(fetch * FETCH_CORRECTION + write)
        ea = summary.get("TCC_EA0_RDREQ_sum", {}).get("per_dispatch")
        out["traffic_note"] = (
            "FETCH_SIZE / WRITE_SIZE are KiB per dispatch of the pool kernel (mean over the dispatches of a --pmc pass of "
            "their own): raw %.3f GB read + %.3f GB written.  gfx950 tallies a 128-byte fabric request as 64 bytes "
            "(MI355X_MICROARCH.md, HBM section; profiles/r02_fetch_size_calibration.json calibrates it on this kernel's "
            "stream and gather patterns), so FETCH_SIZE is doubled"
            % (fetch / 1e9, write / 1e9)
            + ("; cross-check: TCC_EA0_RDREQ %.1f M requests x 128 B = %.3f GB" % (ea / 1e6, ea * 128 / 1e9) if ea else ""))
    json.dump(out, open(os.path.join(DST, "r03_%s_pmc_summary.json" % tag), "w"), indent=1)
    print(tag, "->", out.get("workload_signature"), out.get("traffic_bytes_per_launch_corrected"))


if __name__ == "__main__":
    tags = sys.argv[1:] or [os.path.basename(d)[len("prof_r03_"):] for d in sorted(glob.glob(os.path.join(SRC, "prof_r03_*")))]
    for t in tags:
        publish(t)
