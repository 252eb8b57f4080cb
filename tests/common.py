"""Shared helpers for the test-suite: paths, oracle set-up from a workload dict, result digests."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
sys.path.insert(0, ROOT)

from oracle import polr_oracle as orc  # noqa: E402  (test infrastructure: the checker)
from polr_amd import workloads  # noqa: E402


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".json")
    with open(path) as f:
        return json.load(f)


def oracle_joins(wl):
    """Build the reference-layout hash tables (and perfect tables where the planner would) for a
    workload; returns (probe_cols list, probe_valid list, [JoinSpec])."""
    probe = wl["probe"]
    pcols = list(probe["cols"].values())
    pnames = list(probe["cols"].keys())
    pvalid = [probe.get("valid", {}).get(n) for n in pnames]
    joins = []
    for j in wl["joins"]:
        pv = [j.get("payload_valid", {}).get(n) for n in j["payload"].keys()]
        ht = orc.HashTable(j["keys"], list(j["payload"].values()), key_valid=j.get("key_valid"), payload_valid=pv,
                           null_equal=[bool(f & 2) for f in j.get("key_flags", [])] or None)
        if j.get("perfect") is not None:
            ht.make_perfect(*j["perfect"])
        names = list(j["payload"].keys())
        preds = [(op, src, names.index(col)) for op, src, col in j.get("preds", [])]
        joins.append(orc.JoinSpec(ht, j["key_src"], estimated_cardinality=len(j["keys"][0]), preds=preds))
    return pcols, pvalid, joins


def output_columns(wl):
    """(src_join, column array, validity) for every output column in the reference's order:
    probe columns, then each join's payload in original join order."""
    probe = wl["probe"]
    cols = []
    for n, a in probe["cols"].items():
        cols.append((-1, a, probe.get("valid", {}).get(n)))
    for x, j in enumerate(wl["joins"]):
        for n, a in j["payload"].items():
            cols.append((x, a, j.get("payload_valid", {}).get(n)))
    return cols


def rows_digest_from_columns(columns):
    """columns: list of (data, valid) per output column -> same digest as make_golden.rows_digest
    (rows as int64, NULL -> INT64_MIN, sorted lexicographically, sha256)."""
    if not columns:
        return hashlib.sha256(b"").hexdigest(), 0
    n = len(columns[0][0])
    a = np.empty((n, len(columns)), dtype=np.int64)
    for c, (data, valid) in enumerate(columns):
        col = data.astype(np.int64)
        if valid is not None:
            col = np.where(valid.astype(bool), col, np.iinfo(np.int64).min)
        a[:, c] = col
    if n:
        a = a[np.lexsort(a.T[::-1])]
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest(), n


def oracle_output_digest(wl, out_rows):
    k = len(wl["joins"])
    cols = []
    for src_join, arr, valid in output_columns(wl):
        cols.append(orc.materialize_column(out_rows, k, src_join, arr, valid))
    return rows_digest_from_columns(cols)
