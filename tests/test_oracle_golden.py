"""The oracle (oracle/polr_oracle.c) against everything the reference itself produced:

  * tests/golden/<scenario>.json  -- made by tests/golden/make_golden.py from the reference compiled
    out of its own sources (oracle/ref_build.mk): ALTERNATE matrices, per-round intermediates of every
    deterministic routing strategy with and without chunk caching, per-path tuple counts, result digests;
  * tests/golden/polr_test/       -- the reference's own known-answer test (test/polr/polr.test:6-53):
    its three input tables and its 20 expected rows.
CPU only; this is what pins the restatement before it is used as the checker of the HIP path."""
import os

import numpy as np
import pytest

import common
from common import orc, workloads

SCENARIOS = {
    "star_skew": lambda: workloads.star_skew(),
    "star_skew_nulls": lambda: workloads.star_skew(n_fact=60_000, with_nulls=True),
    "chain_dep": lambda: workloads.chain_dep(),
    "fanout": lambda: workloads.fanout(),
    "star_pred": lambda: workloads.star_pred(),
}
_cache = {}


def scenario(name):
    if name not in _cache:
        wl = SCENARIOS[name]()
        pcols, pvalid, joins = common.oracle_joins(wl)
        _cache[name] = (wl, pcols, pvalid, joins, common.load_golden(name))
    return _cache[name]


def scenario_paths(wl, enumerator):
    k = len(wl["joins"])
    deps = np.zeros((k, k), dtype=np.uint8)
    for i, j in enumerate(wl["joins"]):
        for sj, _ in j["key_src"]:
            if sj >= 0:
                deps[i, sj] = 1
    return orc.enumerate_join_orders(enumerator, k, deps, [len(j["keys"][0]) for j in wl["joins"]], 8)


@pytest.mark.parametrize("name", list(SCENARIOS))
@pytest.mark.parametrize("enumerator", ["each_last_once", "each_first_once"])
def test_alternate_matrix_and_rows(name, enumerator):
    wl, pcols, pvalid, joins, gold = scenario(name)
    g = gold["alternate"][enumerator]
    paths = scenario_paths(wl, enumerator)
    if g is None:
        assert len(paths) < 2
        return
    res = orc.run_pipeline(pcols, joins, paths, routing="alternate", caching=False, probe_valid=pvalid)
    assert np.array_equal(res["alt_matrix"], np.asarray(g["matrix"], dtype=np.uint64))
    assert res["num_intermediates"] == g["intms"]
    assert common.oracle_output_digest(wl, res["out_rows"]) == (g["rows_sha256"], g["n_rows"])
    assert (gold["plain"]["rows_sha256"], gold["plain"]["n_rows"]) == (g["rows_sha256"], g["n_rows"])


def _routing_cases():
    cases = []
    for name in SCENARIOS:
        gold = common.load_golden(name)
        for key in gold["routing"]:
            cases.append((name, key))
    return cases


@pytest.mark.parametrize("name,key", _routing_cases())
def test_routing_trace(name, key):
    wl, pcols, pvalid, joins, gold = scenario(name)
    g = gold["routing"][key]
    parts = key.split("/")
    enumerator, tag = parts[0], parts[1]
    caching = len(parts) < 3 or parts[2] == "cache"
    kw = {"regret_budget": 0.01, "init_tuple_count": 1024, "atc_multiplier": 1}
    routing = tag
    if len(parts) < 3:  # knob variants: the SET statements are stored with the vector
        caching = False
        routing = tag.split("_b")[0].split("_i")[0]
        for s in g["settings"]:
            var, val = s.replace("SET ", "").split(" TO ")
            kw[var] = float(val) if var == "regret_budget" else int(val)
    n_probe = len(pcols[0])
    if routing == "exponential_backoff":
        # regret_budget is re-purposed as the window cap: est_card/10240/10/threads (polar_config.cpp:115-120)
        kw["regret_budget"] = n_probe / 10240.0 / 10 / 1
    paths = scenario_paths(wl, enumerator)
    res = orc.run_pipeline(pcols, joins, paths, routing=routing, caching=caching, probe_valid=pvalid,
                           collect_output="rows_sha256" in g, **kw)
    assert list(res["intermediates_per_round"]) == g["rounds"]
    assert res["num_intermediates"] == g["intms"]
    assert res["input_tuple_count_per_path"] == g["tuple_counts"]
    if "rows_sha256" in g:
        assert common.oracle_output_digest(wl, res["out_rows"]) == (g["rows_sha256"], g["n_rows"])


def test_polr_test_known_answer():
    """test/polr/polr.test: table_a JOIN table_b ON a_a=b_a JOIN table_c ON a_b=c_b, 20 rows, with
    and without POLAR, every routing strategy"""
    d = os.path.join(common.GOLDEN, "polr_test")
    a = np.loadtxt(os.path.join(d, "table_a.csv"), delimiter=",", skiprows=1, dtype=np.int64)
    b = np.loadtxt(os.path.join(d, "table_b.csv"), delimiter=",", skiprows=1, dtype=np.int64)
    c = np.loadtxt(os.path.join(d, "table_c.csv"), delimiter=",", skiprows=1, dtype=np.int64)
    want = np.loadtxt(os.path.join(d, "expected.csv"), delimiter=",", dtype=np.int64)
    a_a, a_b = a[:, 0].astype(np.int32), a[:, 1].astype(np.int32)
    hb = orc.HashTable([b[:, 0].astype(np.int32)], [b[:, 1].astype(np.int32)])
    hc = orc.HashTable([c[:, 1].astype(np.int32)], [c[:, 0].astype(np.int32)])
    joins = [orc.JoinSpec(hb, [(-1, 0)]), orc.JoinSpec(hc, [(-1, 1)])]
    want = want[np.lexsort(want.T[::-1])]
    for routing in orc.ROUTING:
        if routing == "backpressure":
            continue
        res = orc.run_pipeline([a_a, a_b], joins, [[0, 1], [1, 0]], routing=routing)
        o = res["out_rows"]
        got = np.stack([a_a[o[:, 0]], a_b[o[:, 0]], b[o[:, 1], 0], b[o[:, 1], 1], c[o[:, 2], 0], c[o[:, 2], 1]], 1)
        got = got[np.lexsort(got.T[::-1])]
        assert np.array_equal(got, want), routing


def test_polr_minimal_known_answer():
    """test/polr/polr-minimal.test:6-28: range tables, 5 rows"""
    a_a = np.arange(10, dtype=np.int64)
    a_b = a_a + 10
    hb = orc.HashTable([np.arange(0, 5, dtype=np.int64)])
    hc = orc.HashTable([np.arange(10, 15, dtype=np.int64)])
    joins = [orc.JoinSpec(hb, [(-1, 0)]), orc.JoinSpec(hc, [(-1, 1)])]
    res = orc.run_pipeline([a_a, a_b], joins, [[0, 1], [1, 0]], routing="adaptive_reinit")
    o = res["out_rows"]
    got = sorted(zip(a_a[o[:, 0]].tolist(), a_b[o[:, 0]].tolist(), o[:, 1].tolist(), (o[:, 2] + 10).tolist()))
    assert got == [(i, i + 10, i, i + 10) for i in range(5)]


def test_hash_function_known_values():
    """murmurhash64 finaliser (hash.hpp:22-29): algebraic checks + Hash<int32>(-1) sign handling"""
    L = orc.lib()
    assert L.orc_murmurhash64(0) == 0
    x = 0x0123456789ABCDEF
    y = x ^ (x >> 32)
    y = (y * 0xd6e8feb86659fd93) & (2**64 - 1)
    y ^= y >> 32
    y = (y * 0xd6e8feb86659fd93) & (2**64 - 1)
    y ^= y >> 32
    assert L.orc_murmurhash64(x) == y
    v = np.array([-1], dtype=np.int32)
    assert L.orc_hash_value(v.ctypes.data, 4, 1) == L.orc_murmurhash64(0xFFFFFFFF)
    v16 = np.array([-1], dtype=np.int16)
    assert L.orc_hash_value(v16.ctypes.data, 2, 1) == L.orc_murmurhash64(0xFFFFFFFF)
    assert L.orc_combine_hash(3, 5) == ((3 * 0xbf58476d1ce4e5b9) & (2**64 - 1)) ^ 5


def test_row_layout_and_capacity():
    """RowLayout (row_layout.cpp:23-53) and PointerTableCapacity (join_hashtable.hpp:265-267)"""
    keys = [np.arange(100, dtype=np.uint32)]
    ht = orc.HashTable(keys, [np.arange(100, dtype=np.uint16)])
    # 1 validity byte (3 columns incl. hash) + 4 + 2 + 8: the SSB date-join row of SURVEY Appendix B
    assert ht.row_width == 15 and ht.col_offset(0) == 1 and ht.col_offset(1) == 5 and ht.col_offset(2) == 7
    assert ht.capacity == 32768
    big = orc.HashTable([np.arange(40_000, dtype=np.int32)])
    assert big.capacity == 131072
    heads = big.bucket_heads()
    assert (heads != 2**64 - 1).sum() <= 40_000


def _job_light():
    import bench
    wl = workloads.job_light_01(scale=0.1)
    sel = wl["probe"]["filter_sel"]
    n_rows = len(wl["probe"]["cols"]["movie_id"])
    offs = bench.chunk_offsets_for(sel, n_rows, 1024)
    return wl, sel, n_rows, offs


def job_light_budget(routing, n_rows):
    # EXPONENTIAL_BACKOFF re-purposes regret_budget as source est_card/10240/10/threads
    # (polar_config.cpp:115-120); the source's estimated_cardinality is the table's row count
    return n_rows / 10240.0 / 10 / 1 if routing == "exponential_backoff" else 0.01


@pytest.mark.parametrize("routing", ["alternate", "init_once", "opportunistic", "adaptive_reinit", "dynamic",
                                     "exponential_backoff", "default_path"])
def test_job_light_bench_pipeline(routing):
    """the bench.py workload at scale 0.1 through the reference's own SQL (filtered scan -> thinned source
    chunks, pinned left-deep pipeline, both joins probed with mc.movie_id, COUNT(*))"""
    gold = common.load_golden("job_light_01")
    wl, sel, n_rows, offs = _job_light()
    joins = [orc.JoinSpec(orc.HashTable(j["keys"], []), j["key_src"]) for j in wl["joins"]]
    res = orc.run_pipeline(list(wl["probe"]["cols"].values()), joins, [[0, 1], [1, 0]], routing=routing,
                           caching=False, sel=sel, chunk_offsets=offs, collect_output=False,
                           regret_budget=job_light_budget(routing, n_rows))
    assert res["num_output_rows"] == gold["count_star"]
    if routing == "alternate":
        assert np.array_equal(res["alt_matrix"], np.asarray(gold["alternate"]["matrix"], dtype=np.uint64))
        assert res["num_intermediates"] == gold["alternate"]["intms"]
    else:
        g = gold["routing"][routing]
        assert list(res["intermediates_per_round"]) == g["rounds"]
        assert res["num_intermediates"] == g["intms"]
        assert res["input_tuple_count_per_path"] == g["tuple_counts"]
