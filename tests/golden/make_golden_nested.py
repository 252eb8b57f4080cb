#!/usr/bin/env python3
"""tests/golden/make_golden_nested.py -- SelSampleEnumeration over a NESTED (bushy build-side) plan, from THE REFERENCE.

SSB-skew Q4.1 with its customer dimension written as a join of its own -- lineorder JOIN (customer JOIN nation WHERE
n_region = AMERICA) -- so that the build side of the first multiplexed join is a join tree and the reference's
CreateJoinOrderNodes recurses into it (src/parallel/polar_enumeration_algo.cpp:248-287, :401-409).  The rows of that
build side are exactly Q4.1's filtered customers, so the multiplexed pipeline (and the oracle that identifies the join
orders of the bank from the reference's ALTERNATE matrix) is Q4.1's; only the plan statistics SAMPLE reads differ.
Build container only (needs oracle/_ref).  Output: tests/golden/sample_nested.json"""
import itertools
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ssb_skew as g  # noqa: E402  (run / parse_alt and the module paths)
from polr_amd import ssb_skew  # noqa: E402
import common  # noqa: E402

SHAPE = g.SHAPE


def main():
    gold = {"shape": SHAPE, "cases": {}}
    wl = ssb_skew.workload("q4.1", **SHAPE)
    inst = wl["instance"]
    ref = dict(wl["ref"])
    nation = np.arange(50, dtype=np.uint16)
    n_region = np.where(nation < 25, nation // 5, ssb_skew.R_OCEANIA).astype(np.uint8)
    assert np.array_equal(n_region[inst.c_nation], inst.c_region)
    tables = dict(ref["tables"])
    tables["nation"] = {"n_nationkey": nation, "n_region": n_region}
    ref["tables"] = tables
    ref["pk"] = dict(ref["pk"], nation="n_nationkey")
    ref["query"] = ("SELECT COUNT(*) FROM lineorder JOIN (SELECT c_custkey FROM customer JOIN nation ON c_nation = "
                    "n_nationkey WHERE n_region = 1) c ON lo_custkey = c.c_custkey JOIN supplier ON lo_suppkey = s_suppkey "
                    "JOIN part ON lo_partkey = p_partkey JOIN date ON lo_orderdate = d_datekey "
                    "WHERE s_region = 1 AND (p_mfgr = 1 OR p_mfgr = 2)")
    flat = ref["node_info"]
    node_info = [list(flat[0]),
                 [0, False, False, [[len(inst.c_custkey), False, True], [50, True, True]]]] + [list(x) for x in flat[2:]]
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    k = len(ojoins)
    for mjo in (3, 8):
        base = ["PRAGMA enable_polr", "PRAGMA enable_log_tuples_routed", "PRAGMA disable_caching",
                "SET join_enumerator TO 'sample'", "SET max_join_orders TO %d" % mjo]
        log, intms, counts, answer = g.run(ref, base + ["SET multiplexer_routing TO 'alternate'"])
        case = {"max_join_orders": mjo, "node_info": node_info, "count_star": answer, "sql": ref["query"]}
        if log is None:
            case["paths"] = None
        else:
            want = np.asarray(g.parse_alt(log), dtype=np.uint64)
            found = {}
            for perm in itertools.permutations(range(k)):
                res = common.orc.run_pipeline(pcols, ojoins, [list(perm)], routing="alternate", caching=False,
                                              collect_output=False)
                col = res["alt_matrix"][:, 0]
                for p in range(want.shape[1]):
                    if np.array_equal(col, want[:, p]):
                        found.setdefault(p, []).append(list(perm))
            assert all(len(found.get(p, [])) == 1 for p in range(want.shape[1])), found
            case["paths"] = [found[p][0] for p in range(want.shape[1])]
            case["alternate_intms"] = intms
        gold["cases"]["q4.1-nested/%d" % mjo] = case
        print(mjo, case["paths"], answer, flush=True)
    path = os.path.join(HERE, "sample_nested.json")
    json.dump(gold, open(path, "w"), separators=(",", ":"))
    print("wrote", path)


if __name__ == "__main__":
    main()
