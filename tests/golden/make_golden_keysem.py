#!/usr/bin/env python3
"""tests/golden/make_golden_keysem.py -- an IS NOT DISTINCT FROM join under the multiplexer, and what the reference does
with a CAST'ed join key, from THE REFERENCE (polr_amd.workloads.key_semantics):

    SELECT COUNT(*) FROM fact [JOIN dim_a ON fact.a = dim_a.k]                    (INTEGER = BIGINT: CAST(fact.a) on the left)
                              JOIN dim_b ON fact.b IS NOT DISTINCT FROM dim_b.k   (NULLs on both sides, repeated keys)
                              JOIN dim_c ON fact.c = dim_c.k  JOIN dim_d ON fact.d = dim_d.k

with PRAGMA enable_polr, join order pinned (disabled_optimizers 'join_order,statistics_propagation'), each_last_once.
  * WITHOUT dim_a the reference multiplexes the three joins: the ALTERNATE matrix (every chunk through every join
    order), COUNT(*), total intermediates and the routing traces of four strategies are kept, and the generator checks
    that the oracle reproduces all of it.
  * WITH dim_a the reference does NOT multiplex the pipeline at all: POLARConfig::GenerateJoinOrders tests the left side
    of a condition for ExpressionType::CAST (src/parallel/polar_config.cpp:78) where a bound cast carries OPERATOR_CAST
    (src/planner/expression/bound_cast_expression.cpp:13), so any CAST'ed key ends in "Let's not POLAR".  Recorded as
    such, with the query's COUNT(*) -- which the device's by-value keys (POLR_KEY_BY_VALUE) and the oracle (on a pre-cast
    column) must return when THEY multiplex all four joins.
Build container only.  Output: tests/golden/key_semantics.json"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ssb_skew as g  # noqa: E402  (run(): one reference run, its logs parsed)
from polr_amd import workloads  # noqa: E402
from polr_amd import host as phost  # noqa: E402
import common  # noqa: E402

ROUTINGS = ["adaptive_reinit", "init_once", "opportunistic", "dynamic"]


def oracle_run(wl, paths, routing):
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    pcols = list(pcols)
    if wl["joins"][0]["name"] == "dim_a":
        pcols[1] = pcols[1].astype(np.int64)  # CAST(fact.a AS BIGINT), as the reference's binder writes the condition
    return common.orc.run_pipeline(pcols, ojoins, paths, routing=routing, caching=False, collect_output=False,
                                   probe_valid=pvalid)


def main():
    base = ["PRAGMA enable_polr", "PRAGMA enable_log_tuples_routed", "PRAGMA disable_caching",
            "SET join_enumerator TO 'each_last_once'", "SET max_join_orders TO 8"]
    # ---- with the CAST'ed key: the reference does not multiplex
    wl4 = workloads.key_semantics(cast=True)
    log, intms, counts, answer4 = g.run(wl4["ref"], base + ["SET multiplexer_routing TO 'alternate'"])
    assert log is None, "the reference multiplexed a pipeline with a CAST'ed key: fixture and docs are out of date"
    paths4 = phost.generate_join_orders("each_last_once", len(wl4["probe"]["cols"]), [len(j["payload"]) for j in wl4["joins"]],
                                        wl4["cond_left_index"], [len(j["keys"][0]) for j in wl4["joins"]], max_join_orders=8)[0]
    res4 = oracle_run(wl4, paths4, "alternate")
    assert res4["num_output_rows"] == answer4, (res4["num_output_rows"], answer4)
    # ---- without it: three multiplexed joins, one of them IS NOT DISTINCT FROM
    wl = workloads.key_semantics(cast=False)
    ref = wl["ref"]
    paths = phost.generate_join_orders("each_last_once", len(wl["probe"]["cols"]), [len(j["payload"]) for j in wl["joins"]],
                                       wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8)[0]
    log, intms, counts, answer = g.run(ref, base + ["SET multiplexer_routing TO 'alternate'"])
    assert log is not None, "POLAR did not engage"
    want = np.asarray(g.parse_alt(log), dtype=np.uint64)
    assert want.shape[1] == len(paths), (want.shape, paths)
    res = oracle_run(wl, paths, "alternate")
    assert np.array_equal(res["alt_matrix"], want), "oracle ALTERNATE matrix differs from the reference's"
    assert res["num_output_rows"] == answer and res["num_intermediates"] == intms
    gold = {"with_cast": {"sql": wl4["ref"]["query"], "reference_multiplexed": False, "count_star": int(answer4),
                          "paths": np.asarray(paths4).tolist()},
            "sql": ref["query"], "paths": np.asarray(paths).tolist(), "count_star": int(answer), "alternate": want.tolist(),
            "alternate_intms": int(intms), "traces": {}}
    for routing in ROUTINGS:
        log2, intms2, counts2, answer2 = g.run(ref, base + ["SET multiplexer_routing TO '%s'" % routing])
        rounds = g.parse_rounds(log2)
        if routing != "dynamic":  # (DYNAMIC draws noise: its trace is the reference's own, pinned for the totals only)
            res2 = oracle_run(wl, paths, routing)
            assert list(res2["intermediates_per_round"]) == rounds and res2["num_intermediates"] == intms2, routing
        assert answer2 == answer
        gold["traces"][routing] = {"rounds": rounds, "intms": int(intms2), "tuple_counts": counts2}
        print(routing, len(rounds), "rounds", intms2, flush=True)
    path = os.path.join(HERE, "key_semantics.json")
    json.dump(gold, open(path, "w"), separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes; count", answer, "alternate", want.shape, want.sum(axis=0))


if __name__ == "__main__":
    main()
