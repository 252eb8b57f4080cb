"""The SSB-skew queries on the load.sql-transformed data (polr_amd.ssb_skew) with the reference's DEFAULT enumerator
(`sample`) -- against the reference's own runs (tests/golden/ssb_skew_sample.json, tests/golden/make_golden_ssb_skew.py):
ALTERNATE matrix, COUNT(*), and the routing traces of the six deterministic strategies, for Q4.1 / Q4.2 / Q4.3 / Q3.1 /
Q2.1 and two bank sizes.  CPU: the oracle.  GPU (-m gpu): the device through the C ABI -- the FLAT pipeline with
LDS-resident bit tables inside the pool launch (one executor = the reference's single-threaded trace), per-round
launches of the generic kernel, and many executors sharing the pool (row counts only: traces are per executor)."""
import numpy as np
import pytest

import common
from common import orc
from polr_amd import host as phost
from polr_amd import ssb_skew

GOLD = common.load_golden("ssb_skew_sample")
ROUTINGS = ["init_once", "opportunistic", "adaptive_reinit", "dynamic", "exponential_backoff", "default_path"]
CASES = sorted(c for c in GOLD["cases"] if GOLD["cases"][c]["paths"] is not None)
_wl = {}


def workload(case):
    c = GOLD["cases"][case]
    if c["query"] not in _wl:
        _wl[c["query"]] = ssb_skew.workload(c["query"], **GOLD["shape"])
    return _wl[c["query"]], np.asarray(c["paths"], dtype=np.int32), c


def budget(routing, n_rows):
    return n_rows / 10240.0 / 10 / 1 if routing == "exponential_backoff" else 0.01  # polar_config.cpp:115-120


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference(case):
    wl, paths, c = workload(case)
    pcols, pvalid, joins = common.oracle_joins(wl)
    n = len(pcols[0])
    res = orc.run_pipeline(pcols, joins, paths, routing="alternate", caching=False, collect_output=False)
    assert np.array_equal(res["alt_matrix"], np.asarray(c["alternate"]["matrix"], dtype=np.uint64))
    assert res["num_intermediates"] == c["alternate"]["intms"]
    for routing in ROUTINGS:
        res = orc.run_pipeline(pcols, joins, paths, routing=routing, caching=False, collect_output=False,
                               regret_budget=budget(routing, n))
        g = c["routing"][routing]
        assert res["num_output_rows"] == c["count_star"], routing
        assert list(res["intermediates_per_round"]) == g["rounds"], routing
        assert res["num_intermediates"] == g["intms"], routing
        assert res["input_tuple_count_per_path"][:len(paths)] == g["tuple_counts"], routing


@pytest.mark.gpu
@pytest.mark.parametrize("launch", ["pool", "rounds"])
@pytest.mark.parametrize("case", CASES)
def test_device_matches_reference(gpu_ctx, case, launch):
    from polr_amd import capi
    wl, paths, c = workload(case)
    # the bank the host mirror enumerates IS the reference's (also checked on the CPU by test_sample_enumerator.py)
    k = len(wl["joins"])
    gen = phost.generate_join_orders("sample", 4, [0] * k, [[j["key_src"][0][1]] for j in wl["joins"]], [1] * k,
                                     max_join_orders=c["max_join_orders"], node_info=c["node_info"])
    assert gen[0].tolist() == paths.tolist()
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    info = pipe.launch_info(False)
    assert info["flat"] == 1 and info["lds_tables"] >= 1  # single-key unique-match joins on probe columns
    n_chunks = (n + 1023) // 1024
    P = len(paths)
    for routing in ["alternate"] + ROUTINGS:
        mpx = capi.DeviceMultiplexer(pipe, routing, regret_budget=budget(routing, n), max_log_rounds=1 << 16)
        if launch == "pool":
            capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
        else:
            mpx.run(0, n_chunks)
        st = mpx.finish()
        _, _, inter = mpx.fetch_log()
        if routing == "alternate":
            assert np.array_equal(inter.reshape(-1, P), np.asarray(c["alternate"]["matrix"], dtype=np.uint64))
            assert st["num_intermediates"] == c["alternate"]["intms"]
            assert st["stage_out"][0][k - 1] == c["count_star"]  # only path 0 forwards its output
        else:
            g = c["routing"][routing]
            assert list(inter) == g["rounds"], routing
            assert st["num_intermediates"] == g["intms"], routing
            assert st["input_tuple_count_per_path"] == g["tuple_counts"], routing
            assert sum(st["stage_out"][p][k - 1] for p in range(P)) == c["count_star"], routing
        mpx.close()
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_exec", [3, 16, 97])
@pytest.mark.parametrize("routing", ["adaptive_reinit", "dynamic", "opportunistic", "init_once"])
def test_pool_executors_cover_every_tuple_once(gpu_ctx, routing, n_exec):
    """many executors share the pool of probe waves: whatever the routers decide and whichever wave takes which unit,
    every tuple is probed exactly once and COUNT(*) is the reference's; per-executor traces equal single-executor
    runs over the same chunk ranges (an executor's decisions depend on its own counters only)"""
    from polr_amd import capi
    wl, paths, c = workload("q4.1/3")
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    k, P = len(wl["joins"]), len(paths)
    mpxs = [capi.DeviceMultiplexer(pipe, routing, regret_budget=budget(routing, n), max_log_rounds=1 << 14)
            for _ in range(n_exec)]
    ranges = [((e * n_chunks) // n_exec, ((e + 1) * n_chunks) // n_exec) for e in range(n_exec)]
    capi.run_resident(mpxs, ranges, reset=True, finish=True)
    stats = capi.finish_many(mpxs)
    assert sum(sum(st["stage_out"][p][k - 1] for p in range(P)) for st in stats) == c["count_star"]
    assert sum(sum(st["input_tuple_count_per_path"]) for st in stats) == n
    # spot-check three executors against single-executor runs of their ranges
    solo = capi.DeviceMultiplexer(pipe, routing, regret_budget=budget(routing, n), max_log_rounds=1 << 14)
    for e in sorted({0, n_exec // 2, n_exec - 1}):
        capi.run_resident([solo], [ranges[e]], reset=True, finish=True)
        want = solo.finish()
        _, _, want_log = solo.fetch_log()
        _, _, got_log = mpxs[e].fetch_log()
        assert list(got_log) == list(want_log), (routing, e)
        assert stats[e]["num_intermediates"] == want["num_intermediates"]
        assert stats[e]["path_resistances"] == want["path_resistances"]
    solo.close()
    for m in mpxs:
        m.close()
    pipe.close()


@pytest.mark.gpu
def test_two_pool_runs_side_by_side(gpu_ctx):
    """POLR_RUN_SHARE: two runs on two streams, half the device each, in flight together -- each a pool of its own
    (own rings, own routers); both traces equal the single run's"""
    import ctypes as C
    from polr_amd import capi
    wl, paths, c = workload("q4.1/3")
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    ref = capi.DeviceMultiplexer(pipe, "adaptive_reinit", max_log_rounds=1 << 14)
    capi.run_resident([ref], [(0, n_chunks)], reset=True, finish=True)
    want = ref.finish()
    _, _, want_log = ref.fetch_log()
    a = capi.DeviceMultiplexer(pipe, "adaptive_reinit", max_log_rounds=1 << 14)
    b = capi.DeviceMultiplexer(pipe, "adaptive_reinit", max_log_rounds=1 << 14)
    for _ in range(3):
        capi.run_resident([a], [(0, n_chunks)], reset=True, finish=True, share=2)
        capi.run_resident([b], [(0, n_chunks)], reset=True, finish=True, share=2)
    for m in (a, b):
        st = m.finish()
        _, _, log = m.fetch_log()
        assert list(log) == list(want_log)
        assert st["num_intermediates"] == want["num_intermediates"]
    for m in (ref, a, b):
        m.close()
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("morsel_chunks", [8, 120])
def test_backpressure_join_orders_race_for_morsels(gpu_ctx, morsel_chunks):
    """MultiplexerRouting::BACKPRESSURE: one executor per join order over one shared morsel cursor (pipeline.cpp:147-156).
    Which order gets which morsel depends on timing; every tuple is probed exactly once, COUNT(*) is the reference's, and
    an executor only ever uses ITS join order."""
    from polr_amd import capi
    wl, paths, c = workload("q4.1/3")
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    k, P = len(wl["joins"]), len(paths)
    mpxs = [capi.DeviceMultiplexer(pipe, "backpressure") for _ in range(P)]
    capi.run_backpressure(mpxs, 0, n_chunks, morsel_chunks)
    stats = capi.finish_many(mpxs)
    assert sum(sum(st["input_tuple_count_per_path"]) for st in stats) == n
    total = 0
    for p, st in enumerate(stats):
        for q in range(P):
            if q != p:
                assert not any(st["stage_out"][q]), (p, q)  # executor p only ever ran join order p
        total += st["stage_out"][p][k - 1]
    assert total == c["count_star"]
    for m in mpxs:
        m.close()
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mask", [0b0010, 0b1111, 0b0101])
def test_lip_prefilter_keeps_the_result(gpu_ctx, mask):
    """`PRAGMA enable_lip` on the device: the filters of the joins named in the mask thin the source chunks before the
    multiplexer sees them.  The survivors are exactly the rows whose key is on the build side of every such join (the
    device's filters are the joins' own indexes: no false positives), the thinned chunks keep the scan's boundaries, and
    the pipeline's COUNT(*) is unchanged -- what LIP must never change."""
    from polr_amd import capi
    wl, paths, c = workload("q4.1/3")
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    names = list(wl["probe"]["cols"].keys())
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    n_sel, n_chunks = pipe.scan_filter([], lip_joins=mask)
    keep = np.ones(n, dtype=bool)
    for j, jn in enumerate(wl["joins"]):
        if (mask >> j) & 1:
            keep &= np.isin(cols[jn["key_src"][0][1]], jn["keys"][0])
    want = np.nonzero(keep)[0].astype(np.uint32)
    sel, offs = pipe.fetch_scan()
    assert n_sel == len(want) and np.array_equal(sel, want)
    import bench
    assert np.array_equal(offs, bench.chunk_offsets_for(want, n, 1024))
    k, P = len(wl["joins"]), len(paths)
    for routing in ("adaptive_reinit", "default_path"):
        mpx = capi.DeviceMultiplexer(pipe, routing)
        mpx.use_scan_chunks()
        capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
        st = mpx.finish()
        assert sum(st["stage_out"][p][k - 1] for p in range(P)) == c["count_star"]
        assert sum(st["input_tuple_count_per_path"]) == n_sel
        if mask == 0b1111 and routing == "default_path":
            # every join's filter was applied at the source: nothing is dropped inside the pipeline any more
            assert st["num_intermediates"] == n_sel * k and n_sel == c["count_star"]
        mpx.close()
    pipe.close()


@pytest.mark.gpu
def test_lip_rejects_a_dependent_join(gpu_ctx):
    from polr_amd import capi, workloads
    wl = workloads.chain_dep(n_fact=20_000)
    joins = capi.build_joins(gpu_ctx, wl)
    cols = list(wl["probe"]["cols"].values())
    pipe = capi.Pipeline(gpu_ctx, cols, len(cols[0]), joins, [[0, 1, 2]])
    with pytest.raises(capi.PolrError):
        pipe.scan_filter([], lip_joins=0b010)  # join 1 is keyed by a build column of join 0
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R", [2, 5])
def test_executors_with_a_list_of_ranges(gpu_ctx, R):
    """polr_mpx_run_resident_ranges: every executor routes R ranges (one out of each part of the source) one after the
    other with ONE multiplexer state: COUNT(*) and the tuple total are the reference's, and an executor's trace is the
    trace of a single executor handed the same list"""
    from polr_amd import capi
    wl, paths, c = workload("q4.1/3")
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    k, P, E = len(wl["joins"]), len(paths), 7
    lists = []
    for e in range(E):
        lst = []
        for r in range(R):
            lo, hi = (r * n_chunks) // R, ((r + 1) * n_chunks) // R
            lst.append((lo + (e * (hi - lo)) // E, lo + ((e + 1) * (hi - lo)) // E))
        lists.append(lst)
    lists[3][0] = (lists[3][0][0], lists[3][0][0])  # an empty first range: its chunks go to the neighbour
    covered = sum(b - a for lst in lists for a, b in lst)
    mpxs = [capi.DeviceMultiplexer(pipe, "adaptive_reinit", max_log_rounds=1 << 14) for _ in range(E)]
    capi.run_resident_ranges(mpxs, lists, reset=True, finish=True)
    stats = capi.finish_many(mpxs)
    routed = sum(sum(st["input_tuple_count_per_path"]) for st in stats)
    # (executor 3 skipped its first range: count what was covered)
    want_tuples = sum(min(b * 1024, n) - min(a * 1024, n) for lst in lists for a, b in lst)
    assert routed == want_tuples and covered <= n_chunks
    solo = capi.DeviceMultiplexer(pipe, "adaptive_reinit", max_log_rounds=1 << 14)
    for e in (0, 3, E - 1):
        capi.run_resident_ranges([solo], [lists[e]], reset=True, finish=True)
        want = solo.finish()
        _, _, want_log = solo.fetch_log()
        _, _, got_log = mpxs[e].fetch_log()
        assert list(got_log) == list(want_log) and stats[e]["num_intermediates"] == want["num_intermediates"]
    # all ranges together (no gap): COUNT(*) is the reference's
    full = []
    for e in range(E):
        lst = []
        for r in range(R):
            lo, hi = (r * n_chunks) // R, ((r + 1) * n_chunks) // R
            lst.append((lo + (e * (hi - lo)) // E, lo + ((e + 1) * (hi - lo)) // E))
        full.append(lst)
    capi.run_resident_ranges(mpxs, full, reset=True, finish=True)
    stats = capi.finish_many(mpxs)
    assert sum(sum(st["stage_out"][p][k - 1] for p in range(P)) for st in stats) == c["count_star"]
    assert sum(sum(st["input_tuple_count_per_path"]) for st in stats) == n
    solo.close()
    for m in mpxs:
        m.close()
    pipe.close()


def test_reference_runs_of_lip_and_backpressure_are_what_the_oracle_computes():
    """tests/golden/lip_backpressure.json (tests/golden/make_golden_lip_bp.py): the reference with `PRAGMA enable_lip` (its
    bloom pre-filter must not change COUNT(*); under enable_polr the reference itself dies with SIGSEGV on this pipeline,
    so no run of LIP under the multiplexer exists to pin against) and with BACKPRESSURE routing at one thread -- where the
    single task that gets to run takes the whole source down ITS join order, i.e. DEFAULT_PATH: same COUNT(*), same
    intermediates"""
    gold = common.load_golden("lip_backpressure")
    wl, paths, c = workload("q4.1/3")
    assert gold["lip_without_polar"]["count_star"] == c["count_star"] == gold["backpressure_threads_1"]["count_star"]
    assert gold["lip_with_polar"]["returncode"] == -11  # (SIGSEGV: a defect of the reference, recorded as data)
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    res = orc.run_pipeline(pcols, ojoins, paths, routing="default_path", caching=False, collect_output=False)
    assert res["num_output_rows"] == gold["backpressure_threads_1"]["count_star"]
    assert [res["num_intermediates"]] == gold["backpressure_threads_1"]["intms_per_task"] == \
        gold["default_path_threads_1"]["intms_per_task"]
    assert gold["backpressure_threads_1"]["tuple_counts_printed"][0] == gold["source_rows"] == len(pcols[0])
    # four threads: three tasks (one per join order that got a morsel), each reporting under ITS path 0; together the source
    t4 = gold["backpressure_threads_4"]
    assert t4["count_star"] == c["count_star"] and sum(t4["tuple_counts_printed"]) == gold["source_rows"]


@pytest.mark.gpu
def test_backpressure_matches_the_reference_run(gpu_ctx):
    """BACKPRESSURE on the device against the reference's own runs: COUNT(*), every tuple taken exactly once, and -- with
    row-group morsels (120 chunks = 122 880 rows, the reference's scan unit) -- the same morsel sizes dealt to the join
    orders as the reference's four-thread run dealt to its tasks (which order gets which is timing on both sides)"""
    from polr_amd import capi
    gold = common.load_golden("lip_backpressure")
    wl, paths, c = workload("q4.1/3")
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    k, P = len(wl["joins"]), len(paths)
    mpxs = [capi.DeviceMultiplexer(pipe, "backpressure") for _ in range(P)]
    capi.run_backpressure(mpxs, 0, n_chunks, 120)
    stats = capi.finish_many(mpxs)
    took = [sum(st["input_tuple_count_per_path"]) for st in stats]
    assert sum(took) == n == gold["source_rows"]
    assert sum(st["stage_out"][p][k - 1] for p, st in enumerate(stats)) == gold["backpressure_threads_4"]["count_star"]
    morsels = sorted(x for x in gold["backpressure_threads_4"]["tuple_counts_printed"] if x)  # 54 240, 122 880, 122 880
    assert sum(morsels) == n
    # every executor's share is a sum of whole morsels of the reference's sizes
    sums = {0}
    for m_ in morsels:
        sums |= {s_ + m_ for s_ in sums}
    assert all(t_ in sums for t_ in took), (took, morsels)
    for m in mpxs:
        m.close()
    pipe.close()
