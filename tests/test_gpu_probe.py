"""Parity of the HIP path kernel (through the C ABI) with the oracle and the reference's golden
vectors: per-chunk x per-join-order intermediates (the ALTERNATE matrix the reference logs) and the
output row set of every join order.  Bit-exact: these are integer/index results."""
import numpy as np
import pytest

import common
from common import orc, workloads
from polr_amd import capi

pytestmark = pytest.mark.gpu

V = 1024

SCENARIOS = {
    "star_skew": lambda: workloads.star_skew(),
    "star_skew_nulls": lambda: workloads.star_skew(n_fact=60_000, with_nulls=True),
    "chain_dep": lambda: workloads.chain_dep(),
    "fanout": lambda: workloads.fanout(),
    "star_pred": lambda: workloads.star_pred(),
}


def scenario_paths(wl, enumerator):
    k = len(wl["joins"])
    deps = np.zeros((k, k), dtype=np.uint8)
    for i, j in enumerate(wl["joins"]):
        for sj, _ in j["key_src"]:
            if sj >= 0:
                deps[i, sj] = 1
    card = [len(j["keys"][0]) for j in wl["joins"]]
    return orc.enumerate_join_orders(enumerator, k, deps, card, 8)


def gpu_pipeline(ctx, wl, paths):
    joins = capi.build_joins(ctx, wl)
    probe = wl["probe"]
    names = list(probe["cols"].keys())
    pv = [probe.get("valid", {}).get(n) for n in names]
    n = len(probe["cols"][names[0]])
    pipe = capi.Pipeline(ctx, list(probe["cols"].values()), n, joins, paths, probe_valid=pv)
    return pipe, joins, n


def perfect_row_maps(wl, ojoins):
    """slot value -> build-table row for every join (identity for hash tables, idx->row for perfect)"""
    maps = []
    for oj in ojoins:
        maps.append(oj.ht.pht_orig_rows() if oj.ht.pht else None)
    return maps


@pytest.mark.parametrize("name", list(SCENARIOS))
@pytest.mark.parametrize("enumerator", ["each_last_once", "each_first_once"])
def test_alternate_matrix(gpu_ctx, name, enumerator):
    wl = SCENARIOS[name]()
    paths = scenario_paths(wl, enumerator)
    if len(paths) < 2:
        pytest.skip("POLAR does not engage: fewer than two join orders (polar_config.cpp:99)")
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    ref = orc.run_pipeline(pcols, ojoins, paths, routing="alternate", caching=False, probe_valid=pvalid,
                           collect_output=False)
    pipe, joins, n = gpu_pipeline(gpu_ctx, wl, paths)
    rounds = []
    for c in range((n + V - 1) // V):
        for p in range(len(paths)):
            rounds.append((c * V, min(V, n - c * V), p, 0))
    counts = pipe.probe_rounds(rounds)
    got = counts.sum(axis=1).reshape(-1, len(paths))
    assert got.shape == ref["alt_matrix"].shape
    assert np.array_equal(got, ref["alt_matrix"])
    gold = common.load_golden(name)["alternate"].get(enumerator)
    if gold is not None:
        assert np.array_equal(got, np.asarray(gold["matrix"], dtype=np.uint64))
        assert int(got.sum()) == gold["intms"]


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_output_row_set_every_path(gpu_ctx, name):
    wl = SCENARIOS[name]()
    paths = scenario_paths(wl, "each_last_once")
    k = len(wl["joins"])
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    ref = orc.run_pipeline(pcols, ojoins, paths[:1], routing="default_path", probe_valid=pvalid, collect_output=True)
    want = ref["out_rows"]
    want = want[np.lexsort(want.T[::-1])]
    want_digest = common.oracle_output_digest(wl, ref["out_rows"])
    gold = common.load_golden(name)
    assert want_digest[0] == gold["plain"]["rows_sha256"] and want_digest[1] == gold["plain"]["n_rows"]
    pipe, joins, n = gpu_pipeline(gpu_ctx, wl, paths)
    maps = perfect_row_maps(wl, ojoins)
    out = capi.Output(pipe, chunk_capacity=1024, max_chunks=max(64, 4 * (len(want) // 1024 + 1) + 4096))
    for p in range(len(paths)):
        out.reset()
        counts = pipe.probe_rounds([(0, n, p, 1)], out=out)
        ids = out.fetch_ids()
        assert ids.shape == (len(want), 1 + k)
        assert int(counts[0, k - 1]) == len(want)  # last join's output = result cardinality
        rows = ids.copy()
        for x in range(k):
            if maps[x] is not None:
                rows[:, 1 + x] = maps[x][ids[:, 1 + x]]
        rows = rows[np.lexsort(rows.T[::-1])]
        assert np.array_equal(rows, want), "path %d row set differs" % p
        # materialised columns (values + validity) hash to the reference's digest
        cols = []
        for src_join, arr, valid in common.output_columns(wl):
            names = list(wl["probe"]["cols"].keys()) if src_join < 0 else list(wl["joins"][src_join]["payload"].keys())
            src = wl["probe"]["cols"] if src_join < 0 else wl["joins"][src_join]["payload"]
            col_idx = [i for i, nme in enumerate(names) if src[nme] is arr][0]
            cols.append(out.materialize(src_join, col_idx, arr.dtype))
        assert common.rows_digest_from_columns(cols) == want_digest


def test_selection_and_partial_rounds(gpu_ctx):
    """a thinned source (upstream filter) and arbitrary slices: counts add up, nothing outside the
    selection is touched"""
    wl = workloads.star_skew(n_fact=50_000)
    paths = scenario_paths(wl, "each_last_once")
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    rng = np.random.default_rng(7)
    sel = np.nonzero(rng.random(50_000) < 0.37)[0].astype(np.uint32)
    ref = orc.run_pipeline(pcols, ojoins, paths, routing="alternate", caching=False, probe_valid=pvalid,
                           collect_output=True, sel=sel)
    pipe, joins, n = gpu_pipeline(gpu_ctx, wl, paths)
    pipe.set_selection(sel)
    m = len(sel)
    cuts = [0, 1, 63, 64, 65, 1000, 1024, 5000, m]
    rounds = [(cuts[i], cuts[i + 1] - cuts[i], 1, 0) for i in range(len(cuts) - 1)]
    counts = pipe.probe_rounds(rounds)
    assert int(counts.sum()) == int(ref["alt_matrix"][:, 1].sum())
    out = capi.Output(pipe, 1024, 4096)
    pipe.probe_rounds([(0, m, 2, 1)], out=out)
    ids = out.fetch_ids()
    assert np.all(np.isin(ids[:, 0], sel))
    assert len(ids) == ref["num_output_rows"]
    with pytest.raises(capi.PolrError):
        pipe.probe_rounds([(m - 5, 10, 0, 0)])  # slice past the end is rejected on the host


def test_output_overflow_is_reported(gpu_ctx):
    wl = workloads.fanout()
    paths = scenario_paths(wl, "each_last_once")
    pipe, joins, n = gpu_pipeline(gpu_ctx, wl, paths)
    out = capi.Output(pipe, 64, 2)
    with pytest.raises(capi.PolrError) as e:
        pipe.probe_rounds([(0, n, 0, 1)], out=out)
    assert e.value.code == capi.E_OVERFLOW


def test_row_format_upload_matches_columns(gpu_ctx):
    """the reference's row-format blob (oracle builds it byte for byte) and the columnar upload give
    the same join"""
    wl = workloads.chain_dep(n_fact=30_000)
    paths = scenario_paths(wl, "each_last_once")
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    ctx = gpu_ctx
    joins_rows = []
    for j, oj in zip(wl["joins"], ojoins):
        ht = oj.ht
        nk, npay = len(j["keys"]), len(j["payload"])
        off = [ht.col_offset(i) for i in range(nk + npay)]
        wid = [a.dtype.itemsize for a in list(j["keys"]) + list(j["payload"].values())]
        sgn = [a.dtype.kind == "i" for a in list(j["keys"]) + list(j["payload"].values())]
        g = capi.HashTable.from_rows(ctx, ht.rows_blob(), ht.count, ht.row_width, off, wid, sgn, nk, npay)
        if j.get("perfect") is None or not g.finalize_perfect(*j["perfect"]):
            g.finalize_hash()
        joins_rows.append((g, j["key_src"]))
    n = len(pcols[0])
    pipe_r = capi.Pipeline(ctx, pcols, n, joins_rows, paths)
    pipe_c, _, _ = gpu_pipeline(ctx, wl, paths)
    rounds = [(0, n, p, 0) for p in range(len(paths))]
    assert np.array_equal(pipe_r.probe_rounds(rounds), pipe_c.probe_rounds(rounds))
