"""BASELINE.json configs[2] -- SSB-skew 4-way star join, 3 alternative probe orders multiplexed -- at reduced
scale (sf 0.2: 1.2 M lineorder rows) against the reference's own run of the SQL
(tests/golden/ssb_skew_q41.json from tests/golden/make_golden.py): ALTERNATE matrix, COUNT(*), and the routing
traces of the six deterministic strategies with join_enumerator = dfs_min_card, max_join_orders = 3.
CPU: the oracle; GPU (-m gpu): the device path through the C ABI -- per-round launches, the resident launch,
and the resident launch with 4 executors (output row set / COUNT(*) only: traces are per executor)."""
import numpy as np
import pytest

import common
from common import orc, workloads

GOLD = common.load_golden("ssb_skew_q41")
ROUTINGS = ["init_once", "opportunistic", "adaptive_reinit", "dynamic", "exponential_backoff", "default_path"]
_wl = {}


def workload():
    """the workload and the reference's three join orders.  dfs_min_card ranks joins by the planner's estimated
    cardinality of each JOIN node (polar_enumeration_algo.cpp:24-25), an input from the out-of-scope optimizer that
    the reference does not log; make_golden.py identified the orders it used from its ALTERNATE matrix (column p
    equals the per-chunk intermediates of exactly one of the 24 permutations) and stored them in the fixture."""
    if "wl" not in _wl:
        wl = workloads.ssb_skew_q41(sf=GOLD["sf"])
        _wl["wl"] = (wl, np.asarray(GOLD["paths"], dtype=np.int32))
    return _wl["wl"]


def budget(routing, n_rows):
    return n_rows / 10240.0 / 10 / 1 if routing == "exponential_backoff" else 0.01  # polar_config.cpp:115-120


@pytest.mark.parametrize("routing", ["alternate"] + ROUTINGS)
def test_oracle_matches_reference(routing):
    wl, paths = workload()
    assert len(paths) == 3
    pcols, pvalid, joins = common.oracle_joins(wl)
    n = len(pcols[0])
    res = orc.run_pipeline(pcols, joins, paths, routing=routing, caching=False, collect_output=False,
                           regret_budget=budget(routing, n))
    assert res["num_output_rows"] == GOLD["count_star"]
    if routing == "alternate":
        assert np.array_equal(res["alt_matrix"], np.asarray(GOLD["alternate"]["matrix"], dtype=np.uint64))
        assert res["num_intermediates"] == GOLD["alternate"]["intms"]
    else:
        g = GOLD["routing"][routing]
        assert list(res["intermediates_per_round"]) == g["rounds"]
        assert res["num_intermediates"] == g["intms"]
        assert res["input_tuple_count_per_path"][:3] == g["tuple_counts"]


@pytest.mark.gpu
@pytest.mark.parametrize("launch", ["rounds", "resident"])
@pytest.mark.parametrize("routing", ["alternate"] + ROUTINGS)
def test_device_matches_reference(gpu_ctx, routing, launch):
    from polr_amd import capi
    wl, paths = workload()
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    mpx = capi.DeviceMultiplexer(pipe, routing, regret_budget=budget(routing, n), max_log_rounds=1 << 16)
    n_chunks = (n + 1023) // 1024
    if launch == "resident":
        capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
    else:
        mpx.run(0, n_chunks)
    st = mpx.finish()
    _, _, inter = mpx.fetch_log()
    k = len(wl["joins"])
    if routing == "alternate":
        assert np.array_equal(inter.reshape(-1, 3), np.asarray(GOLD["alternate"]["matrix"], dtype=np.uint64))
        assert st["num_intermediates"] == GOLD["alternate"]["intms"]
        assert st["stage_out"][0][k - 1] == GOLD["count_star"]  # only path 0 forwards its output
    else:
        g = GOLD["routing"][routing]
        assert list(inter) == g["rounds"]
        assert st["num_intermediates"] == g["intms"]
        assert st["input_tuple_count_per_path"] == g["tuple_counts"]
        assert sum(st["stage_out"][p][k - 1] for p in range(3)) == GOLD["count_star"]
    mpx.close()
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("routing", ["adaptive_reinit", "exponential_backoff", "dynamic"])
def test_device_executors_count_star(gpu_ctx, routing):
    """4 executors in one resident launch: every tuple is probed exactly once whatever the routing does"""
    from polr_amd import capi
    wl, paths = workload()
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    E = 4
    mpxs = [capi.DeviceMultiplexer(pipe, routing, regret_budget=budget(routing, n)) for _ in range(E)]
    ranges = [((e * n_chunks) // E, ((e + 1) * n_chunks) // E) for e in range(E)]
    capi.run_resident(mpxs, ranges, reset=True, finish=True)
    stats = capi.finish_many(mpxs)
    k = len(wl["joins"])
    assert sum(sum(st["stage_out"][p][k - 1] for p in range(3)) for st in stats) == GOLD["count_star"]
    assert sum(sum(st["input_tuple_count_per_path"]) for st in stats) == n
    for m in mpxs:
        m.close()
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("routing", ["alternate", "exponential_backoff", "init_once"])
def test_many_rounds_published_ahead(gpu_ctx, routing):
    """stress for the resident router's two round slots: 6 M tuples (sf 1.0), one executor -- ALTERNATE publishes
    every one of its 17 580 rounds ahead of the previous round's counters, EXPONENTIAL_BACKOFF re-enters its init
    phase hundreds of times; traces, totals and COUNT(*) against the oracle"""
    from polr_amd import capi
    wl = workloads.ssb_skew_q41(sf=1.0)
    paths = np.asarray(GOLD["paths"], dtype=np.int32)
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    n = len(pcols[0])
    ref = orc.run_pipeline(pcols, ojoins, paths, routing=routing, caching=False, collect_output=False,
                           regret_budget=budget(routing, n))
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    pipe = capi.Pipeline(gpu_ctx, pcols, n, joins, paths)
    mpx = capi.DeviceMultiplexer(pipe, routing, regret_budget=budget(routing, n), max_log_rounds=1 << 16)
    capi.run_resident([mpx], [(0, (n + 1023) // 1024)], reset=True, finish=True)
    st = mpx.finish()
    _, _, inter = mpx.fetch_log()
    k = len(wl["joins"])
    if routing == "alternate":
        assert np.array_equal(inter.reshape(-1, 3), ref["alt_matrix"])
        assert st["stage_out"][0][k - 1] == ref["num_output_rows"]
    else:
        assert list(inter) == list(ref["intermediates_per_round"])
        assert st["input_tuple_count_per_path"] == ref["input_tuple_count_per_path"][:3]
        assert sum(st["stage_out"][p][k - 1] for p in range(3)) == ref["num_output_rows"]
    assert st["num_intermediates"] == ref["num_intermediates"]
    mpx.close()
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("routing", ["adaptive_reinit", "default_path", "opportunistic"])
def test_morsel_driven_executors(gpu_ctx, routing):
    """executors pulling 120-chunk morsels from one cursor (the reference's morsel-driven worker threads): every
    chunk is routed exactly once -- COUNT(*), routed tuples and the output row set equal the single-executor run"""
    from polr_amd import capi
    wl, paths = workload()
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    k = len(wl["joins"])
    ref_out = capi.Output(pipe, 1024, 16384)
    one = capi.DeviceMultiplexer(pipe, routing, regret_budget=budget(routing, n))
    capi.run_resident([one], [(0, n_chunks)], out=ref_out, reset=True, finish=True)
    one.finish()
    want = ref_out.fetch_ids()
    want = want[np.lexsort(want.T[::-1])]
    for n_exec, morsel in ((4, 120), (7, 33), (3, 5000)):
        mpxs = [capi.DeviceMultiplexer(pipe, routing, regret_budget=budget(routing, n)) for _ in range(n_exec)]
        out = capi.Output(pipe, 1024, 16384)
        capi.run_resident_morsels(mpxs, 0, n_chunks, morsel, out=out, reset=True, finish=True)
        stats = capi.finish_many(mpxs)
        assert sum(sum(st["input_tuple_count_per_path"]) for st in stats) == n
        assert sum(sum(st["stage_out"][p][k - 1] for p in range(3)) for st in stats) == GOLD["count_star"]
        got = out.fetch_ids()
        assert np.array_equal(got[np.lexsort(got.T[::-1])], want)
        for m in mpxs:
            m.close()
    one.close()
    pipe.close()
