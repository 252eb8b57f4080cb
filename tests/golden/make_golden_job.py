#!/usr/bin/env python3
"""tests/golden/make_golden_job.py -- BASELINE.json configs[3] pinned against THE REFERENCE: every one of the 113
JOB-shaped pipelines (polr_amd/job_family.py) as SQL over the synthetic IMDB-shaped tables (its `ref` form: left-deep
chain in pipeline order, join order pinned), run by the reference itself at a reduced scale with

    PRAGMA enable_polr; SET join_enumerator TO 'each_last_once'; SET max_join_orders TO 8;
    SET multiplexer_routing TO 'alternate'  (every source chunk through every join order)    and 'adaptive_reinit'

Per query the fixture keeps what the reference logged: how many joins it multiplexed (= the columns of its ALTERNATE log
tell the number of join orders: 1 + joins that can move last), COUNT(*), the per-join-order intermediates totals, a SHA-1
of the whole per-chunk x per-order matrix, the total intermediates of the ALTERNATE run and the ADAPTIVE_REINIT trace
(intermediates of every routing round).  The generator checks on the spot that the oracle (oracle/polr_oracle.c)
reproduces every number; tests/test_job_family.py repeats that without the reference, and on the device.
Build container only (needs oracle/_ref).  Output: tests/golden/job_family.json"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ssb_skew as g  # noqa: E402  (run(): one reference run, its logs parsed)
from polr_amd import job_family as jf  # noqa: E402
from polr_amd import host as phost  # noqa: E402
import common  # noqa: E402

SCALE = 0.004


def oracle_run(wl, paths, routing):
    import bench
    oj = [common.orc.JoinSpec(common.orc.HashTable(j["keys"], list(j["payload"].values())), j["key_src"]) for j in wl["joins"]]
    sel = wl["probe"].get("filter_sel")
    offs = None
    if sel is not None:
        offs = bench.chunk_offsets_for(sel, len(next(iter(wl["probe"]["cols"].values()))), 1024)
    return common.orc.run_pipeline(list(wl["probe"]["cols"].values()), oj, paths, routing=routing, caching=False,
                                   collect_output=False, sel=sel, chunk_offsets=offs)


def select_run(ref, settings):
    """the reference on the pipeline's MIN(...) select list: one row of strings (None = SQL NULL: no row survived)"""
    import subprocess
    import tempfile
    from oracle import ref_run
    workdir = tempfile.mkdtemp(prefix="polr_golden_")
    lines = []
    for name, cols in ref["tables"].items():
        lines += ref_run.table_lines(workdir, name, cols, pk=ref["pk"].get(name))
    lines += ["sql SET threads TO 1"] + ["sql " + s for s in ref["settings"]] + ["sql " + s for s in settings]
    lines.append("query q " + ref["query_select"])
    open(workdir + "/s.txt", "w").write("\n".join(lines) + "\n")
    p = subprocess.run([ref_run.DRIVER, workdir + "/s.txt", workdir + "/out"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    row = open(workdir + "/out/q.csv").read().split("\n")[1].split(",")
    return [None if x == "NULL" else x for x in row]


def oracle_select(wl, paths):
    """MIN of every select-list column over the oracle's output rows (bytes order: what the reference's string comparison
    is for these ASCII values)"""
    res = oracle_run_rows(wl, paths)
    rows = res["out_rows"]
    out = []
    for sj, col in wl["select"]:
        vals = wl["probe"]["strings"][col] if sj < 0 else wl["joins"][sj]["strings"][col]
        ids = rows[:, 0] if sj < 0 else rows[:, 1 + sj]
        out.append(min((vals[i] for i in ids.tolist()), default=None))
    return [None if v is None else v.decode() for v in out]


def oracle_run_rows(wl, paths):
    import bench
    oj = [common.orc.JoinSpec(common.orc.HashTable(j["keys"], list(j["payload"].values())), j["key_src"]) for j in wl["joins"]]
    sel = wl["probe"].get("filter_sel")
    offs = None
    if sel is not None:
        offs = bench.chunk_offsets_for(sel, len(next(iter(wl["probe"]["cols"].values()))), 1024)
    return common.orc.run_pipeline(list(wl["probe"]["cols"].values()), oj, paths[:1], routing="default_path", caching=False,
                                   collect_output=True, sel=sel, chunk_offsets=offs)


def matrix_sha(m):
    return hashlib.sha1(np.ascontiguousarray(m, dtype=np.uint64).tobytes()).hexdigest()


def main():
    shapes = jf.shapes()
    tables = jf.Tables(scale=SCALE)
    gold = {"scale": SCALE, "join_enumerator": "each_last_once", "max_join_orders": 8, "queries": {}}
    for name in sorted(shapes):
        wl = jf.workload(name, tables, shapes[name])
        ref = wl["ref"]
        pn = list(wl["probe"]["cols"].keys())
        k = len(wl["joins"])
        paths = phost.generate_join_orders("each_last_once", len(pn), [len(j["payload"]) for j in wl["joins"]],
                                           wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]],
                                           max_join_orders=8)[0]
        base = ["PRAGMA enable_polr", "PRAGMA enable_log_tuples_routed", "PRAGMA disable_caching",
                "SET join_enumerator TO 'each_last_once'", "SET max_join_orders TO 8"]
        log, intms, counts, answer = g.run(ref, base + ["SET multiplexer_routing TO 'alternate'"])
        assert log is not None, "%s: POLAR did not engage" % name
        want = np.asarray(g.parse_alt(log), dtype=np.uint64)
        assert want.shape[1] == len(paths), (name, want.shape, len(paths))
        res = oracle_run(wl, paths, "alternate")
        assert np.array_equal(res["alt_matrix"], want), "%s: oracle ALTERNATE matrix differs from the reference's" % name
        assert res["num_output_rows"] == answer and res["num_intermediates"] == intms, name
        log2, intms2, counts2, answer2 = g.run(ref, base + ["SET multiplexer_routing TO 'adaptive_reinit'"])
        rounds = g.parse_rounds(log2)
        res2 = oracle_run(wl, paths, "adaptive_reinit")
        assert answer2 == answer and list(res2["intermediates_per_round"]) == rounds and res2["num_intermediates"] == intms2, name
        # the query's own sink: MIN(<varchar>) of every select-list column this pipeline's tables own
        select_min = None
        if wl["select"]:
            select_min = select_run(ref, base + ["SET multiplexer_routing TO 'adaptive_reinit'"])
            assert select_min == oracle_select(wl, paths), (name, select_min, oracle_select(wl, paths))
        gold["queries"][name] = {"select": [[sj, c] for sj, c in wl["select"]], "select_min": select_min,
                                 "joins": [j["name"] for j in wl["joins"]], "n_join_orders": int(want.shape[1]),
                                 "n_chunks": int(want.shape[0]), "count_star": int(answer),
                                 "alternate_sums": want.sum(axis=0).tolist(), "alternate_sha1": matrix_sha(want),
                                 "alternate_intms": int(intms), "adaptive_rounds": rounds, "adaptive_intms": int(intms2),
                                 "adaptive_tuple_counts": counts2, "sql": ref["query"]}
        print(name, k, "joins", want.shape, "count", answer, "intms", intms, intms2, len(rounds), "rounds", flush=True)
    path = os.path.join(HERE, "job_family.json")
    json.dump(gold, open(path, "w"), separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
