#!/usr/bin/env python3
"""Golden answers for the ungrouped aggregate sink, produced by the REFERENCE itself
(oracle/_ref, compiled from its own sources): for every scenario of make_golden.py, with POLAR enabled,

    SELECT count(*), count(c), sum(c), min(c), max(c)  for every output column c  FROM <the multiplexed join>

-> tests/golden/aggregates.json.  Run here only (the reference does not travel):  python tests/golden/make_golden_agg.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "duckdb-polr_amd", "python"))


def make_ssb_q11():
    """BASELINE.json configs[0] (SSB Q1.1 shape) through the reference: filtered scan of lineorder, filtered date
    build side, ungrouped aggregates -> tests/golden/ssb_q11.json"""
    import shutil
    import subprocess
    import tempfile
    from polr_amd import workloads
    wl = workloads.ssb_q11()
    workdir = tempfile.mkdtemp(prefix="polr_golden_")
    try:
        lines = []
        mg.table_script(lines, workdir, "lineorder", wl["probe"]["cols"])
        mg.table_script(lines, workdir, "date", wl["date_full"])
        exprs = ["count(*)", "sum(lo_extendedprice)", "min(lo_extendedprice)", "max(lo_extendedprice)", "max(lo_quantity)",
                 "min(lo_discount)", "sum(lo_discount)", "sum(d_year)", "count(d_year)"]
        lines += ["sql SET threads TO 1", "sql PRAGMA enable_polr",
                  "query q SELECT %s FROM lineorder JOIN date ON lo_orderdate = d_datekey WHERE %s" %
                  (", ".join(exprs), wl["sql_where"]),
                  "query f SELECT count(*) FROM lineorder WHERE lo_discount >= 1 AND lo_discount <= 3 AND lo_quantity < 25"]
        script = os.path.join(workdir, "s.txt")
        open(script, "w").write("\n".join(lines) + "\n")
        outdir = os.path.join(workdir, "out")
        proc = subprocess.run([mg.DRIVER, script, outdir], capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(proc.stdout + proc.stderr)
        vals = [int(v) for v in open(os.path.join(outdir, "q.csv")).read().strip().splitlines()[1].split(",")]
        n_filtered = int(open(os.path.join(outdir, "f.csv")).read().strip().splitlines()[1])
        gold = {"exprs": exprs, "values": vals, "filtered_rows": n_filtered}
        json.dump(gold, open(os.path.join(HERE, "ssb_q11.json"), "w"), indent=1)
        print("ssb_q11", gold)
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


def make_ssb_q41_groups():
    """SSB Q4.1's GROUP BY d_year, c_nation over the SSB-skew shaped star join (sf 0.2), from the reference with
    POLAR enabled -> tests/golden/ssb_q41_groups.json.  sum(lo_revenue - lo_supplycost) is linear: the fixture
    keeps both sums (and the reference's own value of the difference as a cross-check)."""
    import shutil
    import subprocess
    import tempfile
    from polr_amd import workloads
    wl = workloads.ssb_skew_q41(sf=0.2)
    workdir = tempfile.mkdtemp(prefix="polr_golden_")
    try:
        lines = []
        mg.table_script(lines, workdir, wl["probe"]["name"], wl["probe"]["cols"])
        for j in wl["joins"]:
            cols = {kn: kk for kn, kk in zip(j["key_names"], j["keys"])}
            cols.update(j["payload"])
            mg.table_script(lines, workdir, j["name"], cols)
        select = ("d_year, c_nation, count(*), sum(lo_revenue), sum(lo_supplycost), min(lo_revenue), "
                  "max(lo_supplycost), sum(CAST(lo_revenue AS BIGINT) - CAST(lo_supplycost AS BIGINT))")
        query = mg.workload_sql(wl, select)[0] + " GROUP BY d_year, c_nation ORDER BY d_year, c_nation"
        lines += ["sql SET threads TO 1", "sql SET disabled_optimizers TO 'join_order'", "sql PRAGMA enable_polr",
                  "sql SET join_enumerator TO 'dfs_min_card'", "sql SET max_join_orders TO 3", "query q " + query]
        script = os.path.join(workdir, "s.txt")
        open(script, "w").write("\n".join(lines) + "\n")
        outdir = os.path.join(workdir, "out")
        proc = subprocess.run([mg.DRIVER, script, outdir], capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(proc.stdout + proc.stderr)
        rows = [[int(v) for v in l.split(",")] for l in open(os.path.join(outdir, "q.csv")).read().strip().splitlines()[1:]]
        gold = {"columns": ["d_year", "c_nation", "count_star", "sum_revenue", "sum_supplycost", "min_revenue",
                            "max_supplycost", "sum_profit"], "rows": rows}
        json.dump(gold, open(os.path.join(HERE, "ssb_q41_groups.json"), "w"))
        print("ssb_q41_groups", len(rows), "groups", rows[0])
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


def main():
    make_ssb_q11()
    make_ssb_q41_groups()
    gold = {}
    for name, make in mg.SCENARIOS.items():
        wl = make()
        _, out_cols = mg.workload_sql(wl)
        exprs = ["count(*)"]
        for c in out_cols:
            exprs += ["count(%s)" % c, "sum(%s)" % c, "min(%s)" % c, "max(%s)" % c]
        r = mg.run_reference(wl, ["PRAGMA enable_polr", "SET join_enumerator TO 'each_last_once'",
                                  "SET max_join_orders TO 8"], select=", ".join(exprs))
        row = r["rows"][0]
        vals = [None if v != v else int(v) for v in row]
        assert all(v is None or abs(v) < 2 ** 53 for v in vals), "answer not exact in the float parse"
        cols = {}
        for i, c in enumerate(out_cols):
            cnt, s, mn, mx = vals[1 + 4 * i:5 + 4 * i]
            cols[c] = {"count": cnt, "sum": s, "min": mn, "max": mx}
        gold[name] = {"count_star": vals[0], "columns": cols, "out_cols": out_cols}
        print(name, vals[0], len(out_cols), "columns")
    json.dump(gold, open(os.path.join(HERE, "aggregates.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
