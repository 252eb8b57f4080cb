#!/usr/bin/env python3
"""tests/golden/make_golden_lip_bp.py -- what THE REFERENCE does with `PRAGMA enable_lip` and with
`SET multiplexer_routing TO 'backpressure'` on the SSB-skew Q4.1 sample instance of tests/golden/ssb_skew_sample.json.

  * enable_lip WITHOUT POLAR: COUNT(*) (the bloom pre-filter of src/execution/operator/join/physical_hash_join.cpp:
    206-257,579-635 must not change the answer);
  * enable_lip WITH enable_polr: the reference dies with SIGSEGV on this four-join pipeline (every key type tried, with
    and without PRIMARY KEYs) -- recorded as such: there is no reference run of LIP under the multiplexer to pin against;
  * BACKPRESSURE at 1 and 4 threads (src/parallel/pipeline.cpp:147-156: one task per join order over one shared source
    state): COUNT(*), and per task the tuples it took -- each task's multiplexer is a DEFAULT_PATH one and reports them
    under ITS path 0 -- and the intermediates it logged.  The split between the tasks depends on timing at 4 threads;
    their sum is the source.
Build container only.  Output: tests/golden/lip_backpressure.json"""
import glob
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ssb_skew as g  # noqa: E402
from oracle import ref_run  # noqa: E402
from polr_amd import ssb_skew  # noqa: E402


def go(ref, threads, pre):
    workdir = tempfile.mkdtemp(prefix="polr_golden_")
    lines = []
    for name, cols in ref["tables"].items():
        lines += ref_run.table_lines(workdir, name, cols, pk=ref["pk"].get(name))
    lines += ["sql SET threads TO %d" % threads] + ["sql " + s for s in ref["settings"]] + ["sql " + s for s in pre]
    lines.append("query q " + ref["query"])
    open(workdir + "/s.txt", "w").write("\n".join(lines) + "\n")
    p = subprocess.run([ref_run.DRIVER, workdir + "/s.txt", workdir + "/out"], capture_output=True, text=True)
    out = {"returncode": p.returncode}
    if p.returncode == 0:
        out["count_star"] = int(open(workdir + "/out/q.csv").read().strip().splitlines()[1])
        tuples = [int(l.split(":")[1]) for l in p.stdout.splitlines() if ":" in l and l.split(":")[0].strip().isdigit()]
        out["tuple_counts_printed"] = tuples
        out["intms_per_task"] = sorted(int(open(f).read().strip()) for f in glob.glob(workdir + "/out/tmp/*-intms.txt"))
    return out


def main():
    wl = ssb_skew.workload("q4.1", **g.SHAPE)
    ref = wl["ref"]
    n = len(next(iter(ref["tables"]["lineorder"].values())))
    polr = ["PRAGMA enable_polr", "PRAGMA enable_log_tuples_routed", "PRAGMA disable_caching",
            "SET join_enumerator TO 'sample'", "SET max_join_orders TO 3"]
    gold = {"_provenance": __doc__, "shape": g.SHAPE, "query": "q4.1", "source_rows": n,
            "lip_without_polar": go(ref, 1, ["PRAGMA enable_lip"]),
            "lip_with_polar": go(ref, 1, polr + ["PRAGMA enable_lip", "SET multiplexer_routing TO 'adaptive_reinit'"]),
            "backpressure_threads_1": go(ref, 1, polr + ["SET multiplexer_routing TO 'backpressure'"]),
            "backpressure_threads_4": go(ref, 4, polr + ["SET multiplexer_routing TO 'backpressure'"]),
            "default_path_threads_1": go(ref, 1, polr + ["SET multiplexer_routing TO 'default_path'"])}
    for k_, v in gold.items():
        if isinstance(v, dict) and "returncode" in v:
            print(k_, v)
    json.dump(gold, open(os.path.join(HERE, "lip_backpressure.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
