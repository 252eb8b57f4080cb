#!/usr/bin/env python3
"""On the GPU box (a 256-thread host): does the reference answer SSB-skew Q4.1 AS SHIPPED (GROUP BY sink) when the
database is opened with fewer threads (POLR_REF_OPEN_THREADS)?  Runs the small sample instance of
tests/golden/ssb_skew_sample.json through oracle/_ref/ref_driver with and without the cap.
    python tools/dbg_ref_groupby.py"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
from oracle import ref_run  # noqa: E402
from polr_amd import ssb_skew  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "ssb_q41_groupby.json")))


def main():
    wl = ssb_skew.workload("q4.1", **GOLD["shape"])
    inst = wl["instance"]
    cols = inst.lineorder(0, inst.n_lo, cols=list(ssb_skew.PROBE_COLS) + ["lo_revenue", "lo_supplycost"])
    for c in ("lo_revenue", "lo_supplycost"):
        cols[c] = cols[c].astype(np.int32)
    ref = ssb_skew.reference_form(inst, "q4.1", cols)
    for cap in (None, "8", "32"):
        for threads in (1, 8):
            workdir = tempfile.mkdtemp(prefix="polr_dbg_")
            lines = []
            for name, tcols in ref["tables"].items():
                lines += ref_run.table_lines(workdir, name, tcols, pk=ref["pk"].get(name))
            lines += ["sql SET threads TO %d" % threads] + ["sql " + s for s in ref["settings"]]
            lines += ["sql PRAGMA enable_polr", "sql SET join_enumerator TO 'sample'", "sql SET max_join_orders TO 3",
                      "sql SET multiplexer_routing TO 'adaptive_reinit'", "query q " + GOLD["sql"]]
            open(workdir + "/s.txt", "w").write("\n".join(lines) + "\n")
            env = dict(os.environ)
            if cap:
                env["POLR_REF_OPEN_THREADS"] = cap
            p = subprocess.run([ref_run.DRIVER, workdir + "/s.txt", workdir + "/out"], capture_output=True, text=True, env=env)
            ok = p.returncode == 0
            rows = None
            if ok:
                rows = [[int(x) for x in l.split(",")] for l in open(workdir + "/out/q.csv").read().strip().splitlines()[1:]]
            print("open threads %s, SET threads %d: %s%s" % (cap or "hardware (%d)" % os.cpu_count(), threads,
                  "ok, %d groups, equal to the fixture: %s" % (len(rows), rows == GOLD["rows"]) if ok else "FAILED ",
                  "" if ok else (p.stdout + p.stderr)[-200:].replace("\n", " | ")), flush=True)


if __name__ == "__main__":
    main()
