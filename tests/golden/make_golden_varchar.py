#!/usr/bin/env python3
"""tests/golden/make_golden_varchar.py -- VARCHAR join keys under the multiplexer, from THE REFERENCE
(polr_amd.workloads.varchar_keys):

    SELECT COUNT(*) FROM fact JOIN dim_s ON fact.s = dim_s.k      (strings of 2..46 characters, NULLs on both sides, repeated keys)
                              JOIN dim_t ON dim_s.t = dim_t.k     (a VARCHAR build column of the join before as the key)
                              JOIN dim_c ON fact.c = dim_c.k

with PRAGMA enable_polr, join order pinned, each_last_once: the ALTERNATE matrix, COUNT(*), total intermediates and three
routing traces.  The generator checks that the oracle reproduces all of it on DICTIONARY CODES of the strings (the oracle has
no strings; equal strings <-> equal codes).  Build container only.  Output: tests/golden/varchar_keys.json"""
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

ROUTINGS = ["adaptive_reinit", "init_once", "opportunistic"]


def oracle_run(wl, paths, routing):
    pcols, pvalid, ojoins = common.oracle_joins(wl["codes"])
    return common.orc.run_pipeline(pcols, ojoins, paths, routing=routing, caching=False, collect_output=False,
                                   probe_valid=pvalid)


def main():
    wl = workloads.varchar_keys()
    ref = wl["ref"]
    paths = phost.generate_join_orders("each_last_once", 3, [3, 2, 2], wl["cond_left_index"],
                                       [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8)[0]
    base = ["PRAGMA enable_polr", "PRAGMA enable_log_tuples_routed", "PRAGMA disable_caching",
            "SET join_enumerator TO 'each_last_once'", "SET max_join_orders TO 8"]
    log, intms, counts, answer = g.run(ref, base + ["SET multiplexer_routing TO 'alternate'"])
    assert log is not None, "POLAR did not engage"
    want = np.asarray(g.parse_alt(log), dtype=np.uint64)
    assert want.shape[1] == len(paths), (want.shape, paths)
    res = oracle_run(wl, paths, "alternate")
    assert np.array_equal(res["alt_matrix"], want), "oracle ALTERNATE matrix differs from the reference's"
    assert res["num_output_rows"] == answer and res["num_intermediates"] == intms
    gold = {"sql": ref["query"], "paths": np.asarray(paths).tolist(), "count_star": int(answer), "alternate": want.tolist(),
            "alternate_intms": int(intms), "traces": {}}
    for routing in ROUTINGS:
        log2, intms2, counts2, answer2 = g.run(ref, base + ["SET multiplexer_routing TO '%s'" % routing])
        rounds = g.parse_rounds(log2)
        res2 = oracle_run(wl, paths, routing)
        assert list(res2["intermediates_per_round"]) == rounds and res2["num_intermediates"] == intms2, routing
        assert answer2 == answer
        gold["traces"][routing] = {"rounds": rounds, "intms": int(intms2), "tuple_counts": counts2}
        print(routing, len(rounds), "rounds", intms2, flush=True)
    path = os.path.join(HERE, "varchar_keys.json")
    json.dump(gold, open(path, "w"), separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes; count", answer, "alternate", want.shape, want.sum(axis=0))


if __name__ == "__main__":
    main()
