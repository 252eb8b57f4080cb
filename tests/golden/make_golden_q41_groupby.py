#!/usr/bin/env python3
"""tests/golden/make_golden_q41_groupby.py -- SSB-skew Q4.1 AS THE REFERENCE SHIPS IT
(benchmark/ssb-skew/queries/q4-1.sql: SELECT d_year, c_nation, SUM(lo_revenue - lo_supplycost) AS profit ... GROUP BY
d_year, c_nation ORDER BY d_year, c_nation) on the sample instance of tests/golden/ssb_skew_sample.json, answered by the
reference itself with POLAR on (join_enumerator sample, max_join_orders 3, adaptive_reinit).  Strings are codes here
(c_nation = nation code).  Build container only.  Output: tests/golden/ssb_q41_groupby.json"""
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

SQL = ("SELECT d_year, c_nation, SUM(lo_revenue - lo_supplycost) AS profit FROM lineorder JOIN customer ON lo_custkey = "
       "c_custkey JOIN supplier ON lo_suppkey = s_suppkey JOIN part ON lo_partkey = p_partkey JOIN date ON lo_orderdate = "
       "d_datekey WHERE c_region = 1 AND s_region = 1 AND (p_mfgr = 1 OR p_mfgr = 2) GROUP BY d_year, c_nation "
       "ORDER BY d_year, c_nation")


def main():
    wl = ssb_skew.workload("q4.1", **g.SHAPE)
    inst = wl["instance"]
    cols = inst.lineorder(0, inst.n_lo, cols=list(ssb_skew.PROBE_COLS) + ["lo_revenue", "lo_supplycost"])
    import numpy as np
    for c in ("lo_revenue", "lo_supplycost"):  # (INTEGER columns in SSB: the difference may be negative)
        cols[c] = cols[c].astype(np.int32)
    ref = ssb_skew.reference_form(inst, "q4.1", cols)
    workdir = tempfile.mkdtemp(prefix="polr_golden_")
    lines = []
    for name, tcols in ref["tables"].items():
        lines += ref_run.table_lines(workdir, name, tcols, pk=ref["pk"].get(name))
    lines += ["sql SET threads TO 1"] + ["sql " + s for s in ref["settings"]]
    lines += ["sql PRAGMA enable_polr", "sql SET join_enumerator TO 'sample'", "sql SET max_join_orders TO 3",
              "sql SET multiplexer_routing TO 'adaptive_reinit'", "query q " + SQL]
    open(workdir + "/s.txt", "w").write("\n".join(lines) + "\n")
    p = subprocess.run([ref_run.DRIVER, workdir + "/s.txt", workdir + "/out"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    rows = [[int(x) for x in l.split(",")] for l in open(workdir + "/out/q.csv").read().strip().splitlines()[1:]]
    gold = {"_provenance": __doc__, "shape": g.SHAPE, "sql": SQL, "rows": rows}
    json.dump(gold, open(os.path.join(HERE, "ssb_q41_groupby.json"), "w"), separators=(",", ":"))
    print(len(rows), "groups; first", rows[:3], "sum", sum(r[2] for r in rows))


if __name__ == "__main__":
    main()
