#!/usr/bin/env python3
"""tests/golden/make_golden_sf1000.py -- BASELINE.json configs[4] (SSB-skew SF1000, broadcast build + partitioned probe
over 8 GPUs) pinned against THE REFERENCE on what one rank sees: contiguous 8 M-row samples of the lineorder partition
of rank 7 of 8 (rows 5.25 G .. 6 G of the 6 G-row table, skew phase two) and of rank 3 of 8 (rows 2.25 G .. 3 G, phase
one), the FULL SF1000 dimension tables (customer 30 M + 2 500, supplier 2 M, part 2 M, date 2 556), Q4.1; the reference
(oracle/_ref, one thread, POLAR on, join_enumerator sample, max_join_orders 3, adaptive_reinit) answers COUNT(*).
The rows are those polr_amd.ssb_skew generates on any rank (pure function of the row index; tests/test_ssb_skew.py).
Build container only.  Output: tests/golden/ssb_sf1000_samples.json"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
sys.path.insert(0, ROOT)
from polr_amd import ssb_skew  # noqa: E402
from polr_amd import dist as pdist  # noqa: E402
from oracle import ref_run  # noqa: E402

SCALE, QUERY, ROWS = 1000, "q4.1", 7_999_488


def main():
    z = ssb_skew.sizes(SCALE)
    t0 = time.time()
    wl = ssb_skew.workload(QUERY, sf=SCALE, n_lo=z["n_lo"], host_probe=False)
    inst = wl["instance"]
    print("instance: %.1f s; customer %d supplier %d part %d" % (time.time() - t0, len(inst.c_custkey), inst.n_s, inst.n_p), flush=True)
    gold = {"_provenance": __doc__, "query": QUERY, "scale": SCALE, "world": 8, "sample_rows_each": ROWS, "samples": [],
            "params": {k: v for k, v in inst.params().items() if k != "year_band_ends"}}
    for rank, frac in ((7, 0.15), (7, 0.85), (3, 0.5)):
        lo, hi = pdist.probe_partition(z["n_lo"], 8, rank, 1024)
        s0 = ((lo + int((hi - lo - ROWS) * frac)) // 1024) * 1024
        cols = inst.lineorder(s0, s0 + ROWS)
        ref = ssb_skew.reference_form(inst, QUERY, cols)
        settings = list(ref["settings"]) + ["SET multiplexer_routing TO 'adaptive_reinit'", "SET join_enumerator TO 'sample'",
                                            "SET max_join_orders TO 3"]
        t1 = time.time()
        ms, wall, result = ref_run.time_polar_pipeline(ref["tables"], ref["query"], settings, 1, repeat=1, pk=ref["pk"])
        count = int(result.strip().splitlines()[1].split(",")[0])
        gold["samples"].append({"rank": rank, "partition": [lo, hi], "start": s0, "start_in_partition": s0 - lo,
                                "count_star": count, "reference_pipeline_ms": ms})
        print("rank %d rows %d.. : COUNT(*) %d, pipeline %s ms, %.1f s" % (rank, s0, count, ms, time.time() - t1), flush=True)
    json.dump(gold, open(os.path.join(HERE, "ssb_sf1000_samples.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
