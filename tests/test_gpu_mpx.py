"""Device-resident multiplexer (router kernel + path kernel, zero host round trips per decision)
against the routing traces the reference itself logged (tests/golden/*.json): intermediates of
every routing round, per-path tuple counts, total intermediates and the result digest.  Bit-exact."""
import numpy as np
import pytest

import common
from common import orc, workloads
from polr_amd import capi
from test_gpu_probe import SCENARIOS, gpu_pipeline, scenario_paths

pytestmark = pytest.mark.gpu


def _cases():
    cases = []
    for name in SCENARIOS:
        gold = common.load_golden(name)
        for key in gold["routing"]:
            parts = key.split("/")
            if len(parts) == 3 and parts[2] == "cache":
                continue  # chunk caching changes chunk boundaries only; the rounds are identical
            cases.append((name, key))
    return cases


_pipes = {}


def pipeline_for(ctx, name, enumerator):
    key = (name, enumerator)
    if key not in _pipes:
        wl = SCENARIOS[name]()
        paths = scenario_paths(wl, enumerator)
        pipe, joins, n = gpu_pipeline(ctx, wl, paths)
        _pipes[key] = (wl, paths, pipe, joins, n)
    return _pipes[key]


def _run(mpx, launch, a, b, out=None):
    """per-round self-routing launches, or the whole run as one resident launch"""
    if launch == "resident":
        mpx.run_resident(a, b, out=out)
    else:
        mpx.run(a, b, out=out)


LAUNCHES = ["rounds", "resident"]


@pytest.mark.parametrize("launch", LAUNCHES)
@pytest.mark.parametrize("name,key", _cases())
def test_device_routing_matches_reference(gpu_ctx, name, key, launch):
    gold = common.load_golden(name)
    g = gold["routing"][key]
    parts = key.split("/")
    enumerator, tag = parts[0], parts[1]
    kw = {"regret_budget": 0.01, "init_tuple_count": 1024, "atc_multiplier": 1}
    routing = tag
    if len(parts) < 3:
        routing = tag.split("_b")[0].split("_i")[0]
        for s in g["settings"]:
            var, val = s.replace("SET ", "").split(" TO ")
            kw[var] = float(val) if var == "regret_budget" else int(val)
    wl, paths, pipe, joins, n = pipeline_for(gpu_ctx, name, enumerator)
    if routing == "exponential_backoff":
        kw["regret_budget"] = n / 10240.0 / 10 / 1  # polar_config.cpp:115-120
    mpx = capi.DeviceMultiplexer(pipe, routing, chunk_size=1024, **kw)
    out = capi.Output(pipe, 1024, 8192) if "rows_sha256" in g else None
    n_chunks = (n + 1023) // 1024
    # route the source in three morsels: state (incl. an open routing window) carries across calls
    cuts = [0, n_chunks // 3, n_chunks // 3 + 1, n_chunks]
    for a, b in zip(cuts[:-1], cuts[1:]):
        _run(mpx, launch, a, b, out=out)
    st = mpx.finish()
    path, tuples, inter = mpx.fetch_log()
    assert list(inter) == g["rounds"]
    assert st["num_intermediates"] == g["intms"]
    assert st["input_tuple_count_per_path"] == g["tuple_counts"]
    assert int(tuples.sum()) == n
    if out is not None:
        cols = []
        for src_join, arr, valid in common.output_columns(wl):
            names = list(wl["probe"]["cols"].keys()) if src_join < 0 else list(wl["joins"][src_join]["payload"].keys())
            src = wl["probe"]["cols"] if src_join < 0 else wl["joins"][src_join]["payload"]
            col_idx = [i for i, nme in enumerate(names) if src[nme] is arr][0]
            cols.append(out.materialize(src_join, col_idx, arr.dtype))
        assert common.rows_digest_from_columns(cols) == (g["rows_sha256"], g["n_rows"])


@pytest.mark.parametrize("launch", LAUNCHES)
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_device_alternate_matches_reference(gpu_ctx, name, launch):
    gold = common.load_golden(name)
    g = gold["alternate"]["each_last_once"]
    wl, paths, pipe, joins, n = pipeline_for(gpu_ctx, name, "each_last_once")
    mpx = capi.DeviceMultiplexer(pipe, "alternate", chunk_size=1024)
    out = capi.Output(pipe, 1024, 8192)
    _run(mpx, launch, 0, (n + 1023) // 1024, out=out)
    st = mpx.finish()
    path, tuples, inter = mpx.fetch_log()
    P = len(paths)
    assert np.array_equal(inter.reshape(-1, P), np.asarray(g["matrix"], dtype=np.uint64))
    assert st["num_intermediates"] == g["intms"]
    n_rows, _, overflow = out.stats()
    assert not overflow and n_rows == g["n_rows"]  # only path 0 forwards its output


@pytest.mark.parametrize("launch", LAUNCHES)
@pytest.mark.parametrize("routing", ["init_once", "opportunistic", "adaptive_reinit", "dynamic",
                                     "exponential_backoff", "default_path"])
def test_bench_workload_matches_reference(gpu_ctx, routing, launch):
    """bench.py's pipeline (selection list + thinned chunk offsets + COUNT(*) sink) at scale 0.1 against the
    reference's own run of the same SQL"""
    from test_oracle_golden import _job_light, job_light_budget
    gold = common.load_golden("job_light_01")
    wl, sel, n_rows, offs = _job_light()
    joins = capi.build_joins(gpu_ctx, wl)
    pipe = capi.Pipeline(gpu_ctx, list(wl["probe"]["cols"].values()), n_rows, joins, [[0, 1], [1, 0]])
    pipe.set_selection(sel)
    mpx = capi.DeviceMultiplexer(pipe, routing, regret_budget=job_light_budget(routing, n_rows))
    mpx.set_chunk_offsets(offs)
    _run(mpx, launch, 0, len(offs) - 1)
    st = mpx.finish()
    path, tuples, inter = mpx.fetch_log()
    g = gold["routing"][routing]
    assert list(inter) == g["rounds"]
    assert st["num_intermediates"] == g["intms"]
    assert st["input_tuple_count_per_path"] == g["tuple_counts"]
    # COUNT(*): the last join's output over all paths
    k = 2
    assert sum(st["stage_out"][p][k - 1] for p in range(2)) == gold["count_star"]


def _sorted_rows(ids):
    return ids[np.lexsort(ids.T[::-1])]


@pytest.mark.parametrize("launch", ["many", "resident"])
@pytest.mark.parametrize("n_exec", [2, 3, 8])
def test_executors_match_single_executor_runs(gpu_ctx, launch, n_exec):
    """E executors over disjoint chunk ranges (polr_mpx_run_many / _run_resident) = E independent
    single-executor runs over the same ranges: same round logs, same statistics, same output row set"""
    wl, paths, pipe, joins, n = pipeline_for(gpu_ctx, "star_skew", "each_last_once")
    n_chunks = (n + 1023) // 1024
    ranges = [((e * n_chunks) // n_exec, ((e + 1) * n_chunks) // n_exec) for e in range(n_exec)]
    want_logs, want_stats = [], []
    ref_out = capi.Output(pipe, 1024, 8192)
    for a, b in ranges:
        m = capi.DeviceMultiplexer(pipe, "adaptive_reinit", chunk_size=1024)
        m.run(a, b, out=ref_out)
        want_stats.append(m.finish())
        want_logs.append(m.fetch_log())
        m.close()
    want_rows = _sorted_rows(ref_out.fetch_ids())
    mpxs = [capi.DeviceMultiplexer(pipe, "adaptive_reinit", chunk_size=1024) for _ in range(n_exec)]
    out = capi.Output(pipe, 1024, 8192)
    for rep in range(2):  # a second pass after reset: sync words and epochs carry over correctly
        out.reset()
        gpu_ctx.sync()  # (the output object is reset on the context's stream, the runs use their own)
        if launch == "many" or rep == 0:
            for m in mpxs:
                m.reset()
        if launch == "many":
            capi.run_many(mpxs, ranges, out=out)
        elif rep == 0:
            capi.run_resident(mpxs, ranges, out=out)
        else:  # reset and closing FinalizePathRun folded into the same launch
            capi.run_resident(mpxs, ranges, out=out, reset=True, finish=True)
        stats = capi.finish_many(mpxs)
        for e in range(n_exec):
            for key in ("num_intermediates", "num_rounds", "input_tuple_count_per_path", "path_resistances",
                        "stage_out"):
                assert stats[e][key] == want_stats[e][key], (rep, e, key)
            got = mpxs[e].fetch_log()
            for a_, b_ in zip(got, want_logs[e]):
                assert np.array_equal(a_, b_)
        n_rows, _, overflow = out.stats()
        assert not overflow
        assert np.array_equal(_sorted_rows(out.fetch_ids()), want_rows)
    for m in mpxs:
        m.close()


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_auto_indexed_tables_match_reference(gpu_ctx, name):
    """polr_ht_finalize_auto (dense unique integer keys of ANY range become perfect tables, beyond the
    reference's 1 M-value cap): same routing trace, same totals, same result digest as the reference"""
    gold = common.load_golden(name)
    g = gold["routing"]["each_last_once/adaptive_reinit/nocache"]
    wl = SCENARIOS[name]()
    paths = scenario_paths(wl, "each_last_once")
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    kinds = [j[0].info()["kind"] for j in joins]
    probe = wl["probe"]
    names = list(probe["cols"].keys())
    pv = [probe.get("valid", {}).get(n_) for n_ in names]
    n = len(probe["cols"][names[0]])
    pipe = capi.Pipeline(gpu_ctx, list(probe["cols"].values()), n, joins, paths, probe_valid=pv)
    mpx = capi.DeviceMultiplexer(pipe, "adaptive_reinit", chunk_size=1024)
    out = capi.Output(pipe, 1024, 8192)
    mpx.run_resident(0, (n + 1023) // 1024, out=out)
    st = mpx.finish()
    _, tuples, inter = mpx.fetch_log()
    assert list(inter) == g["rounds"], kinds
    assert st["num_intermediates"] == g["intms"]
    assert st["input_tuple_count_per_path"] == g["tuple_counts"]
    cols = []
    for src_join, arr, valid in common.output_columns(wl):
        src = wl["probe"]["cols"] if src_join < 0 else wl["joins"][src_join]["payload"]
        col_idx = [i for i, a in enumerate(src.values()) if a is arr][0]
        cols.append(out.materialize(src_join, col_idx, arr.dtype))
    assert common.rows_digest_from_columns(cols) == (g["rows_sha256"], g["n_rows"])
    mpx.close()
    pipe.close()


@pytest.mark.parametrize("routing", ["adaptive_reinit", "dynamic", "init_once"])
def test_job_q18_shape_with_dependent_joins(gpu_ctx, routing):
    """JOB 18a shape at scale 0.02: six multiplexed joins, two keyed by a build column of an earlier join, join
    orders from the host mirror of GenerateJoinOrders, filtered source -- device (resident launch) vs oracle:
    same routing trace, same totals, same COUNT(*)"""
    from polr_amd import host
    wl = workloads.job_q18(scale=0.02)
    gen = host.generate_join_orders("each_last_once", len(wl["probe"]["cols"]),
                                    [len(j["payload"]) for j in wl["joins"]], wl["cond_left_index"],
                                    [len(j["keys"][0]) for j in wl["joins"]])
    paths = gen[0]
    assert len(paths) == 4 and gen[2][3][2] == 1 and gen[2][5][4] == 1  # it1 after mi, it2 after mi_idx
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    names = list(wl["probe"]["cols"].keys())
    osel, ooffs = orc.scan_filter(pcols, [(names.index(c), op, v) for c, op, v in wl["probe"]["filter"]])
    ref = orc.run_pipeline(pcols, ojoins, paths, routing=routing, caching=False, collect_output=False, sel=osel,
                           chunk_offsets=ooffs)
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    pipe = capi.Pipeline(gpu_ctx, pcols, len(pcols[0]), joins, paths)
    n_sel, n_chunks = pipe.scan_filter([(names.index(c), op, v) for c, op, v in wl["probe"]["filter"]])
    assert n_sel == len(osel) and n_chunks == len(ooffs) - 1
    mpx = capi.DeviceMultiplexer(pipe, routing)
    mpx.use_scan_chunks()
    capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
    st = mpx.finish()
    _, _, inter = mpx.fetch_log()
    assert list(inter) == list(ref["intermediates_per_round"])
    assert st["num_intermediates"] == ref["num_intermediates"]
    assert st["input_tuple_count_per_path"] == ref["input_tuple_count_per_path"][:len(paths)]
    k = len(wl["joins"])
    assert sum(st["stage_out"][p][k - 1] for p in range(len(paths))) == ref["num_output_rows"]
    mpx.close()
    pipe.close()


def test_round_counter_wraps(gpu_ctx):
    """more routing rounds in one resident run than the 20-bit round number of the device protocol holds:
    ALTERNATE over 2-tuple chunks, 3 join orders, 350 000 chunks = 1 050 000 rounds (> 2^20); trace vs oracle"""
    wl = workloads.star_skew(n_fact=700_000)
    k = len(wl["joins"])
    paths = workloads.default_paths(k, "each_last_once")
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    n = len(pcols[0])
    ref = orc.run_pipeline(pcols, ojoins, paths, routing="alternate", caching=False, collect_output=False,
                           vector_size=2)
    joins = capi.build_joins(gpu_ctx, wl)
    pipe = capi.Pipeline(gpu_ctx, pcols, n, joins, paths)
    n_chunks = (n + 1) // 2
    assert n_chunks * len(paths) > (1 << 20)
    mpx = capi.DeviceMultiplexer(pipe, "alternate", chunk_size=2, max_log_rounds=n_chunks * len(paths) + 8)
    capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
    st = mpx.finish()
    _, tuples, inter = mpx.fetch_log()
    assert len(inter) == n_chunks * len(paths)
    assert np.array_equal(inter.reshape(-1, len(paths)), ref["alt_matrix"])
    assert st["num_intermediates"] == ref["num_intermediates"]
    mpx.close()
    pipe.close()


def test_a_given_up_run_leaves_the_rings_usable(gpu_ctx):
    """the device-side watchdog (polr_pool_tuning.watchdog_us) gives a run up: polr_mpx_finish_many reports POLR_E_HIP, and
    the next run on the same rings -- owned by the run's FIRST multiplexer, whose own router had finished long before the
    other executor's watchdog fired -- starts from re-initialised rings and is correct"""
    wl = workloads.star_skew(n_fact=12_000_000)
    k = len(wl["joins"])
    paths = workloads.default_paths(k, "each_last_once")
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    n = len(pcols[0])
    joins = capi.build_joins(gpu_ctx, wl)
    pipe = capi.Pipeline(gpu_ctx, pcols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    ms = [capi.DeviceMultiplexer(pipe, "default_path") for _ in range(2)]
    ranges = [(0, 1), (1, n_chunks)]  # executor 0 (the leader) routes one chunk and is done
    try:
        gpu_ctx.set_pool_tuning(watchdog_us=1)
        capi.run_resident(ms, ranges, reset=True, finish=True)
        with pytest.raises(capi.PolrError) as e:
            capi.finish_many(ms)
        assert e.value.code == capi.E_HIP and "timed out" in str(e.value)
    finally:
        gpu_ctx.set_pool_tuning()
    ref = orc.run_pipeline(pcols, ojoins, paths, routing="default_path", caching=False, collect_output=False)
    for _ in range(2):
        capi.run_resident(ms, ranges, reset=True, finish=True)
        sts = capi.finish_many(ms)
        assert sum(st["num_intermediates"] for st in sts) == ref["num_intermediates"]
        assert sum(sum(st["input_tuple_count_per_path"]) for st in sts) == n
    for m in ms:
        m.close()
    pipe.close()


def _product_fanout(n_fact=30_000, n_hot=4, reps=48, seed=11):
    """three joins with repeated build keys on the same probe column -- a hot key meets reps^3 build-row triples -- each
    feeding a small dimension keyed by ITS payload, so that every build id is needed downstream (nothing folds into a
    multiplicity): the JOB 25c situation, where one source tuple is hundreds of thousands of pairs"""
    rng = np.random.default_rng(seed)
    n_m = 4_000
    m_keys = (np.arange(n_m, dtype=np.int64) * 7 + 3).astype(np.int32)
    hot = m_keys[rng.choice(n_m, n_hot, replace=False)]
    fact_keys = m_keys[rng.integers(0, n_m, n_fact)]
    fact_keys[rng.choice(n_fact, 20, replace=False)] = hot[rng.integers(0, n_hot, 20)]
    fact = {"id": np.arange(n_fact, dtype=np.int32), "mk": fact_keys.astype(np.int32)}
    joins = []
    for x, name in enumerate(("mi", "mi_idx", "mk")):
        r = rng.integers(0, 3, size=n_m)
        r[np.isin(m_keys, hot)] = reps - 8 * x
        b = np.repeat(m_keys, r)
        b = b[rng.permutation(len(b))]
        joins.append({"name": name, "keys": [b.astype(np.int32)], "key_names": ["movie_id"],
                      "payload": {"t": (rng.integers(0, 5, len(b)) + 10 * x).astype(np.int32)}, "key_src": [(-1, 1)],
                      "perfect": None})
    for x, name in enumerate(("it1", "it2", "k")):
        keys = (np.arange(4, dtype=np.int32) + 10 * x).astype(np.int32)  # 4 of the 5 payload values survive
        joins.append({"name": name, "keys": [keys], "key_names": ["id"], "payload": {"v": keys * 3},
                      "key_src": [(x, 0)], "perfect": (int(keys.min()), int(keys.max()))})
    return {"name": "product_fanout", "probe": {"name": "fact", "cols": fact}, "joins": joins}


@pytest.mark.parametrize("share_after", [16, 0xFFFFFFFF])
def test_work_sharing_keeps_every_count(gpu_ctx, share_after):
    """work sharing (polr_pool_tuning.share_after): a probe wave that is long on one unit hands parts of it to the pool --
    the back of its source range, half of a run of build rows, half of the tuples waiting in front of a stage.  Whatever is
    cut where, the routing trace (intermediates per round), every per-position counter, COUNT(*) and the materialised row
    set are those of the oracle -- with sharing after 16 steps (it happens hundreds of times here) and with sharing off"""
    from polr_amd import host
    wl = _product_fanout()
    k = len(wl["joins"])
    # (BoundReference index of every join's probe-side key: 2 probe columns, then one payload column per join)
    gen = host.generate_join_orders("each_last_once", len(wl["probe"]["cols"]), [len(j["payload"]) for j in wl["joins"]],
                                    [[1], [1], [1], [2], [3], [4]], [len(j["keys"][0]) for j in wl["joins"]],
                                    max_join_orders=4)
    paths = gen[0]
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    n = len(pcols[0])
    ref = orc.run_pipeline(pcols, ojoins, paths, routing="adaptive_reinit", caching=False, collect_output=True)
    assert ref["num_intermediates"] > 2_000_000
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    pipe = capi.Pipeline(gpu_ctx, pcols, n, joins, paths)
    n_chunks = (n + 1023) // 1024
    try:
        gpu_ctx.set_pool_tuning(share_after=share_after)
        # counting run
        mpx = capi.DeviceMultiplexer(pipe, "adaptive_reinit")
        capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
        st = mpx.finish()
        _, _, inter = mpx.fetch_log()
        assert list(inter) == list(ref["intermediates_per_round"])
        assert st["num_intermediates"] == ref["num_intermediates"]
        assert st["input_tuple_count_per_path"] == ref["input_tuple_count_per_path"][:len(paths)]
        assert sum(st["stage_out"][p][k - 1] for p in range(len(paths))) == ref["num_output_rows"]
        mpx.close()
        # emitting run: the row set
        mpx = capi.DeviceMultiplexer(pipe, "adaptive_reinit")
        out = capi.Output(pipe, 1024, ref["num_output_rows"] // 1024 + 4096)
        mpx.run_resident(0, n_chunks, out=out)
        st = mpx.finish()
        assert st["num_intermediates"] == ref["num_intermediates"]
        cols = []
        for src_join, arr, valid in common.output_columns(wl):
            src = wl["probe"]["cols"] if src_join < 0 else wl["joins"][src_join]["payload"]
            col_idx = [i for i, a in enumerate(src.values()) if a is arr][0]
            cols.append(out.materialize(src_join, col_idx, arr.dtype))
        assert common.rows_digest_from_columns(cols) == common.oracle_output_digest(wl, ref["out_rows"])
        mpx.close()
    finally:
        gpu_ctx.set_pool_tuning()
    pipe.close()


def test_full_size_runs_on_streams_of_their_own(gpu_ctx):
    """pool launches of MANY pipelines, each sized for the whole device, enqueued without a synchronisation in between, every
    pipeline's on the stream of its own leader.  Side by side each launch would hold part of the device with waves that wait
    for work, and the watchdog of whichever router waited longest gave its run up: `python bench.py --workload job_full
    --scale 0.2 --executors 4 --own-streams --job-limit 48 --steps 12 --no-kernel-events` ended in `run timed out waiting for
    its probe waves` 8 times out of 8 (an executor's first exploration round: 0 of its 4 units in 4 s; a bulk round: 62 of
    2 472 units missing) until the library ordered such launches itself (order_pool_launch, polr_mpx.hip: 2 of 2 clean).
    This test is the small version of that command -- it keeps the path exercised; the bench command is the one that
    showed the failure.  Every pipeline must finish, every pass, with the counts of a pass on one stream"""
    from polr_amd import job_family as jf, host
    shapes = jf.shapes()
    tables = jf.Tables(scale=0.05)
    cases = []
    for name in sorted(shapes)[:32]:
        wl = jf.workload(name, tables, shapes[name])
        pn = list(wl["probe"]["cols"].keys())
        paths = host.generate_join_orders("each_last_once", len(pn), [len(j["payload"]) for j in wl["joins"]],
                                          wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8)[0]
        joins = capi.build_joins(gpu_ctx, wl, auto=True)
        n_rows = len(wl["probe"]["cols"][pn[0]])
        pipe = capi.Pipeline(gpu_ctx, list(wl["probe"]["cols"].values()), n_rows, joins, paths)
        flt = wl["probe"].get("filter")
        if flt:
            _, n_chunks = pipe.scan_filter([(pn.index(c), op, const) for c, op, const in flt], vector_size=1024)
        else:
            n_chunks = (n_rows + 1023) // 1024
        e_n = max(1, min(4, n_chunks))
        ms = []
        for _e in range(e_n):
            m = capi.DeviceMultiplexer(pipe, "adaptive_reinit", log_rounds=False)
            if flt:
                m.use_scan_chunks()
            ms.append(m)
        cases.append((pipe, ms, [((e * n_chunks) // e_n, ((e + 1) * n_chunks) // e_n) for e in range(e_n)], len(wl["joins"]),
                      len(paths)))
    def finish_all():
        got = []
        for _p, ms, _r, k, P in cases:
            sts = capi.finish_many(ms)
            got.append((sum(st["num_intermediates"] for st in sts), sum(st["stage_out"][p][k - 1] for st in sts for p in range(P))))
        return got

    try:
        gpu_ctx.set_pool_tuning(watchdog_us=2_000_000)
        for _p, ms, ranges, _k, _P in cases:
            capi.run_resident(ms, ranges, reset=True, finish=True, stream=gpu_ctx.stream())  # (one stream: one after the other)
        want = finish_all()
        for _round in range(2):
            for _pass in range(8):  # (nothing waits in between: hundreds of launches in flight on 32 streams)
                for _p, ms, ranges, _k, _P in cases:
                    capi.run_resident(ms, ranges, reset=True, finish=True)  # (no stream argument: the leader's own)
            assert finish_all() == want
    finally:
        gpu_ctx.set_pool_tuning()
    for pipe, ms, _r, _k, _P in cases:
        for m in ms:
            m.close()
        pipe.close()
