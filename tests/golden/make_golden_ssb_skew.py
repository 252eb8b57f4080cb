#!/usr/bin/env python3
"""tests/golden/make_golden_ssb_skew.py -- golden vectors for the SSB-skew queries on the load.sql-transformed data
(polr_amd.ssb_skew), from THE REFERENCE ITSELF, with its DEFAULT join enumerator `sample` (SelSampleEnumeration).

For every (query, max_join_orders) the reference runs the query's pinned left-deep pipeline over full dimension tables
with PRIMARY KEYs and pushed-down filters and logs:
  ALTERNATE matrix -> which join orders SelSampleEnumeration put in the bank (each column equals the per-chunk
                      intermediates of exactly one permutation, identified with the oracle), COUNT(*)
  routing traces   -> per-round intermediates of the six deterministic strategies, totals, per-path tuple counts
Build container only (needs oracle/_ref).  Output: tests/golden/ssb_skew_sample.json
"""
import glob
import itertools
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from polr_amd import ssb_skew  # noqa: E402
from oracle import ref_run  # noqa: E402
import common  # noqa: E402

RAW = {}
SHAPE = dict(n_lo=300_000, n_c=30_000, n_s=20_000, n_p=40_000)
ROUTINGS = ["init_once", "opportunistic", "adaptive_reinit", "dynamic", "exponential_backoff", "default_path"]
CASES = [("q4.1", 3), ("q4.1", 8), ("q4.2", 3), ("q4.3", 3), ("q3.1", 3), ("q2.1", 8)]


def run(ref, settings):
    workdir = tempfile.mkdtemp(prefix="polr_golden_")
    try:
        lines = []
        for name, cols in ref["tables"].items():
            lines += ref_run.table_lines(workdir, name, cols, pk=ref["pk"].get(name))
        lines += ["sql SET threads TO 1"] + ["sql " + s for s in ref["settings"]] + ["sql " + s for s in settings]
        lines.append("query q " + ref["query"])
        script = os.path.join(workdir, "s.txt")
        open(script, "w").write("\n".join(lines) + "\n")
        proc = subprocess.run([ref_run.DRIVER, script, os.path.join(workdir, "out")], capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(proc.stdout + proc.stderr)
        logs = [f for f in glob.glob(os.path.join(workdir, "out", "tmp", "*.csv")) if "-" not in os.path.basename(f)]
        intms = glob.glob(os.path.join(workdir, "out", "tmp", "*-intms.txt"))
        counts = [int(l.split(":")[1]) for l in proc.stdout.splitlines()
                  if ":" in l and l.split(":")[0].strip().isdigit()]
        answer = int(open(os.path.join(workdir, "out", "q.csv")).read().strip().splitlines()[1])
        enum = glob.glob(os.path.join(workdir, "out", "tmp", "*-enumeration.csv"))
        RAW.clear()
        RAW.update({"log": open(logs[0]).read() if logs else None, "intms": open(intms[0]).read() if intms else None,
                    "enumeration_header": open(enum[0]).read().splitlines()[0] if enum else None,
                    "names": sorted(os.path.basename(f) for f in glob.glob(os.path.join(workdir, "out", "tmp", "*")))})
        return (open(logs[0]).read() if logs else None), (int(open(intms[0]).read().strip()) if intms else None), \
            counts, answer
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


def parse_alt(csv_text):
    return [[int(x) for x in l.rstrip(",").split(",")] for l in csv_text.strip().splitlines()[1:]]


def parse_rounds(csv_text):
    return [int(x) for x in csv_text.strip().splitlines()[1:]]


def main():
    gold = {"shape": SHAPE, "join_enumerator": "sample", "cases": {}}
    for query, mjo in CASES:
        wl = ssb_skew.workload(query, **SHAPE)
        ref = wl["ref"]
        base = ["PRAGMA enable_polr", "PRAGMA enable_log_tuples_routed", "PRAGMA disable_caching",
                "SET join_enumerator TO 'sample'", "SET max_join_orders TO %d" % mjo]
        log, intms, counts, answer = run(ref, base + ["SET multiplexer_routing TO 'alternate'"])
        case = {"query": query, "max_join_orders": mjo, "node_info": ref["node_info"], "count_star": answer,
                "sql": ref["query"]}
        if log is None:
            case["paths"] = None  # POLAR did not engage
            gold["cases"]["%s/%d" % (query, mjo)] = case
            continue
        if (query, mjo) == ("q4.1", 3):
            # the reference's files as it wrote them (data fixtures for the harness test): text of the ALTERNATE log,
            # of the totals file, header of the enumeration file, and the file names of the run
            raw = {"alternate_log": RAW["log"], "alternate_intms": RAW["intms"],
                   "enumeration_header": RAW["enumeration_header"], "file_names": RAW["names"]}
            ll, _i, _c, _a = run(ref, base + ["SET multiplexer_routing TO 'adaptive_reinit'",
                                              "PRAGMA enable_measure_pipeline"])
            raw["adaptive_log"] = RAW["log"]
            raw["adaptive_intms"] = RAW["intms"]
            raw["adaptive_file_names"] = RAW["names"]
            gold["raw_files"] = raw
        want = np.asarray(parse_alt(log), dtype=np.uint64)
        case["alternate"] = {"matrix": want.tolist(), "intms": intms}
        pcols, pvalid, ojoins = common.oracle_joins(wl)
        k = len(ojoins)
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
        case["routing"] = {}
        for routing in ROUTINGS:
            log, intms, counts, ans = run(ref, base + ["SET multiplexer_routing TO '%s'" % routing])
            assert ans == answer
            case["routing"][routing] = {"rounds": parse_rounds(log), "intms": intms, "tuple_counts": counts}
        gold["cases"]["%s/%d" % (query, mjo)] = case
        print(query, mjo, "paths", case["paths"], "count", answer, flush=True)
    path = os.path.join(HERE, "ssb_skew_sample.json")
    json.dump(gold, open(path, "w"), separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
