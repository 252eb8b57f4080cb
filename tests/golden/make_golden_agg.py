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


def main():
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
