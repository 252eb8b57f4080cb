"""BASELINE.json configs[3] -- the 113 JOB-shaped pipelines (polr_amd/job_family.py): every query of
benchmark/imdb_plan_cost/queries yields a multiplexed pipeline; sampled ones run on the device (-m gpu) against the
oracle: ALTERNATE matrix over the enumerated join orders, an adaptive trace and COUNT(*)."""
import numpy as np
import pytest

import common
from common import orc
from polr_amd import host as phost
from polr_amd import job_family as jf

SHAPES = jf.shapes()
SAMPLE = ["01a", "06d", "10c", "13b", "16b", "18a", "22c", "25a", "29a", "33c"]
_TABLES = {}


def _tables(scale):
    if scale not in _TABLES:
        _TABLES[scale] = jf.Tables(scale=scale)
    return _TABLES[scale]


def _matrix_sha(m):
    import hashlib
    return hashlib.sha1(np.ascontiguousarray(m, dtype=np.uint64).tobytes()).hexdigest()


def _each_last_once(wl):
    pn = list(wl["probe"]["cols"].keys())
    return phost.generate_join_orders("each_last_once", len(pn), [len(j["payload"]) for j in wl["joins"]],
                                      wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8)[0]


def test_all_113_queries_give_a_pipeline():
    t = jf.Tables(scale=0.002)
    assert len(SHAPES) == 113
    n_dep = 0
    for name in sorted(SHAPES):
        wl = jf.workload(name, t, SHAPES[name])
        assert wl is not None and 2 <= len(wl["joins"]) <= 8, name
        for x, j in enumerate(wl["joins"]):
            sj, sc = j["key_src"][0]
            assert sj < x  # a join is keyed by the probe side or by an EARLIER join's build column
            n_dep += sj >= 0
        # the host mirror of POLARConfig::GenerateJoinOrders accepts it and respects the dependencies
        gen = phost.generate_join_orders("each_last_once", len(wl["probe"]["cols"]),
                                         [len(j["payload"]) for j in wl["joins"]], wl["cond_left_index"],
                                         [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8)
        assert gen is not None and gen[0][0].tolist() == list(range(len(wl["joins"]))), name
    assert n_dep > 100


def test_generation_is_deterministic():
    a = jf.workload("18a", jf.Tables(scale=0.002), SHAPES["18a"])
    b = jf.workload("18a", jf.Tables(scale=0.002), SHAPES["18a"])
    assert all(np.array_equal(x["keys"][0], y["keys"][0]) for x, y in zip(a["joins"], b["joins"]))
    assert all(np.array_equal(a["probe"]["cols"][c], b["probe"]["cols"][c]) for c in a["probe"]["cols"])


def _oracle(wl, paths, routing):
    oj = [orc.JoinSpec(orc.HashTable(j["keys"], list(j["payload"].values())), j["key_src"]) for j in wl["joins"]]
    sel = wl["probe"].get("filter_sel")
    offs = None
    if sel is not None:
        import bench
        offs = bench.chunk_offsets_for(sel, len(next(iter(wl["probe"]["cols"].values()))), 1024)
    return orc.run_pipeline(list(wl["probe"]["cols"].values()), oj, paths, routing=routing, caching=False,
                            collect_output=False, sel=sel, chunk_offsets=offs)


def _oracle_rows(wl, paths):
    oj = [orc.JoinSpec(orc.HashTable(j["keys"], list(j["payload"].values())), j["key_src"]) for j in wl["joins"]]
    sel = wl["probe"].get("filter_sel")
    offs = None
    if sel is not None:
        import bench
        offs = bench.chunk_offsets_for(sel, len(next(iter(wl["probe"]["cols"].values()))), 1024)
    return orc.run_pipeline(list(wl["probe"]["cols"].values()), oj, paths[:1], routing="default_path", caching=False,
                            collect_output=True, sel=sel, chunk_offsets=offs)["out_rows"]


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_oracle_matches_the_reference_on_every_query(name):
    """tests/golden/job_family.json (tests/golden/make_golden_job.py): the reference itself ran every one of the 113
    pipelines as SQL over the synthetic tables -- join order pinned, so that its one POLAR pipeline multiplexes exactly the
    joins of ours (src/parallel/polar_config.cpp:19-71) -- with ALTERNATE and with ADAPTIVE_REINIT routing; the oracle
    reproduces the per-chunk x per-order intermediates (SHA-1 of the matrix, column sums), COUNT(*), the totals and the
    adaptive trace"""
    gold = common.load_golden("job_family")
    g = gold["queries"][name]
    wl = jf.workload(name, _tables(gold["scale"]), SHAPES[name])
    assert [j["name"] for j in wl["joins"]] == g["joins"] and wl["ref"]["query"] == g["sql"]
    paths = _each_last_once(wl)
    assert len(paths) == g["n_join_orders"]
    res = _oracle(wl, paths, "alternate")
    assert res["alt_matrix"].shape == (g["n_chunks"], g["n_join_orders"])
    assert res["alt_matrix"].sum(axis=0).tolist() == g["alternate_sums"]
    assert _matrix_sha(res["alt_matrix"]) == g["alternate_sha1"]
    assert res["num_output_rows"] == g["count_star"] and res["num_intermediates"] == g["alternate_intms"]
    res = _oracle(wl, paths, "adaptive_reinit")
    assert list(res["intermediates_per_round"]) == g["adaptive_rounds"] and res["num_intermediates"] == g["adaptive_intms"]
    assert res["num_output_rows"] == g["count_star"]
    # the query's own sink: MIN(<varchar>) per select-list column, against the reference's answer
    assert [[sj, c] for sj, c in wl["select"]] == g["select"]
    if wl["select"]:
        rows = _oracle_rows(wl, paths)
        got = []
        for sj, col in wl["select"]:
            vals = wl["probe"]["strings"][col] if sj < 0 else wl["joins"][sj]["strings"][col]
            ids = rows[:, 0] if sj < 0 else rows[:, 1 + sj]
            m = min((vals[i] for i in ids.tolist()), default=None)
            got.append(None if m is None else m.decode())
        assert got == g["select_min"]


@pytest.mark.gpu
def test_device_matches_the_reference_on_every_query(gpu_ctx):
    """the same fixture on the device, all 113 pipelines through the C ABI: pool launch, one executor (the reference ran
    single-threaded) -- ALTERNATE matrix bit-equal (SHA-1), COUNT(*), totals, and the ADAPTIVE_REINIT trace round by round"""
    from polr_amd import capi
    gold = common.load_golden("job_family")
    t = _tables(gold["scale"])
    for name in sorted(SHAPES):
        g = gold["queries"][name]
        wl = jf.workload(name, t, SHAPES[name])
        pn = list(wl["probe"]["cols"].keys())
        paths = _each_last_once(wl)
        joins = capi.build_joins(gpu_ctx, wl, auto=True)
        cols = list(wl["probe"]["cols"].values())
        # VARCHAR columns of the probe table that the select list names: string_t cells behind the key columns + their heap
        pstr = [capi.string_cells(v) for v in wl["probe"].get("strings", {}).values()]
        pipe = capi.Pipeline(gpu_ctx, cols + [c for c, _h in pstr], len(cols[0]), joins, paths)
        for i, (_c, heap) in enumerate(pstr):
            pipe.set_probe_heap(len(cols) + i, heap)
        flt = wl["probe"].get("filter")
        if flt:
            _n, n_chunks = pipe.scan_filter([(pn.index(c), op, const) for c, op, const in flt])
        else:
            n_chunks = (len(cols[0]) + 1023) // 1024
        k, P = len(wl["joins"]), len(paths)
        assert P == g["n_join_orders"] and n_chunks == g["n_chunks"], name
        for routing in ("alternate", "adaptive_reinit"):
            mpx = capi.DeviceMultiplexer(pipe, routing, max_log_rounds=1 << 16)
            if flt:
                mpx.use_scan_chunks()
            capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
            st = mpx.finish()
            _, _, inter = mpx.fetch_log()
            if routing == "alternate":
                m = np.asarray(inter, dtype=np.uint64).reshape(-1, P)
                assert m.sum(axis=0).tolist() == g["alternate_sums"] and _matrix_sha(m) == g["alternate_sha1"], name
                assert st["num_intermediates"] == g["alternate_intms"], name
                assert st["stage_out"][0][k - 1] == g["count_star"], name
            else:
                assert list(int(x) for x in inter) == g["adaptive_rounds"], name
                assert st["num_intermediates"] == g["adaptive_intms"], name
                assert sum(st["stage_out"][p][k - 1] for p in range(P)) == g["count_star"], name
            mpx.close()
        if wl["select"]:
            # the query's own sink on the device: a materialising adaptive run, then MIN over the VARCHAR column's string_t
            # cells (inline and heap strings) by output row id -- polr_out_aggregate_string
            out = capi.Output(pipe, 1024, 16384)
            mpx = capi.DeviceMultiplexer(pipe, "adaptive_reinit")
            if flt:
                mpx.use_scan_chunks()
            capi.run_resident([mpx], [(0, n_chunks)], out=out, reset=True, finish=True)
            mpx.finish()
            got = []
            for sj, col in wl["select"]:
                if sj < 0:
                    v = out.aggregate_string("min", -1, len(cols) + list(wl["probe"]["strings"].keys()).index(col))
                else:
                    v = out.aggregate_string("min", sj, capi.string_payload_index(wl["joins"][sj], col))
                got.append(None if v is None else v.decode())
            assert got == g["select_min"], (name, got, g["select_min"])
            assert out.stats()[0] == g["count_star"], name
            mpx.close()
            out.close()
        pipe.close()
        for ht, _ in joins:
            ht.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SAMPLE)
def test_device_matches_oracle(gpu_ctx, name):
    from polr_amd import capi
    t = jf.Tables(scale=0.004)
    wl = jf.workload(name, t, SHAPES[name])
    pn = list(wl["probe"]["cols"].keys())
    paths = phost.generate_join_orders("each_last_once", len(pn), [len(j["payload"]) for j in wl["joins"]],
                                       wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]],
                                       max_join_orders=8)[0]
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    pipe = capi.Pipeline(gpu_ctx, cols, len(cols[0]), joins, paths)
    flt = wl["probe"].get("filter")
    if flt:
        n_sel, n_chunks = pipe.scan_filter([(pn.index(c), op, const) for c, op, const in flt])
        assert n_sel == len(wl["probe"]["filter_sel"])
    else:
        n_chunks = (len(cols[0]) + 1023) // 1024
    k, P = len(wl["joins"]), len(paths)
    for routing in ("alternate", "adaptive_reinit"):
        ref = _oracle(wl, paths, routing)
        mpx = capi.DeviceMultiplexer(pipe, routing, max_log_rounds=1 << 16)
        if flt:
            mpx.use_scan_chunks()
        capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
        st = mpx.finish()
        _, _, inter = mpx.fetch_log()
        if routing == "alternate":
            assert np.array_equal(inter.reshape(-1, P), ref["alt_matrix"]), name
        else:
            assert list(inter) == list(ref["intermediates_per_round"]), name
            assert sum(st["stage_out"][p][k - 1] for p in range(P)) == ref["num_output_rows"]
        assert st["num_intermediates"] == ref["num_intermediates"]
        mpx.close()
    pipe.close()


@pytest.mark.gpu
def test_pool_launch_and_path_kernel_agree_on_all_113(gpu_ctx):
    """every join order of every one of the 113 pipelines through the POOL launch (generic pipeline of
    polr_gen_device.h with multiplicity folding, or the flat pipeline) and through the per-round path kernel
    (polr_probe_device.h): two independent implementations of RunPath; the tuples every join produces at every position
    of every join order must be identical (what the multiplexer's reward is computed from), and COUNT(*) the same down
    every join order"""
    from polr_amd import capi
    t = jf.Tables(scale=0.03)
    checked = 0
    for name in sorted(SHAPES):
        wl = jf.workload(name, t, SHAPES[name])
        pn = list(wl["probe"]["cols"].keys())
        paths = phost.generate_join_orders("each_last_once", len(pn), [len(j["payload"]) for j in wl["joins"]],
                                           wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]],
                                           max_join_orders=8)[0]
        joins = capi.build_joins(gpu_ctx, wl, auto=True)
        cols = list(wl["probe"]["cols"].values())
        pipe = capi.Pipeline(gpu_ctx, cols, len(cols[0]), joins, paths)
        flt = wl["probe"].get("filter")
        if flt:
            n, n_chunks = pipe.scan_filter([(pn.index(c), op, const) for c, op, const in flt])
        else:
            n, n_chunks = len(cols[0]), (len(cols[0]) + 1023) // 1024
        k, P = len(wl["joins"]), len(paths)
        want = pipe.probe_rounds([(0, n, p, 0) for p in range(P)])
        mpx = capi.DeviceMultiplexer(pipe, "alternate", log_rounds=False)
        if flt:
            mpx.use_scan_chunks()
        capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
        st = mpx.finish()
        got = np.asarray([[st["stage_out"][p][j] for j in range(k)] for p in range(P)], dtype=np.uint64)
        assert np.array_equal(got, want), name
        assert len(set(int(x) for x in want[:, k - 1])) == 1, name  # COUNT(*) does not depend on the join order
        checked += 1
        mpx.close()
        pipe.close()
        for ht, _ in joins:
            ht.close()
    assert checked == 113


@pytest.mark.gpu
@pytest.mark.parametrize("share_after", [16, 0xFFFFFFFF])
def test_heavy_fanout_shapes_with_work_sharing(gpu_ctx, share_after):
    """the JOB shapes whose fan-out joins multiply (25c: one cast_info row meets hundreds of thousands of build-row triples
    at scale 1; 17e, 31a, 19c) at scale 0.05, COUNTING runs -- multiplicities folded where nobody reads the build rows,
    expansions where they are read -- with work sharing after 16 steps and with sharing off: the ADAPTIVE_REINIT trace, the
    totals and COUNT(*) are the oracle's, whatever was cut and handed to other waves"""
    from polr_amd import capi
    t = _tables(0.05)
    for name in ("25c", "17e", "31a", "19c"):
        wl = jf.workload(name, t, SHAPES[name])
        pn = list(wl["probe"]["cols"].keys())
        paths = _each_last_once(wl)
        oj = [orc.JoinSpec(orc.HashTable(j["keys"], list(j["payload"].values())), j["key_src"]) for j in wl["joins"]]
        sel = wl["probe"].get("filter_sel")
        offs = None
        if sel is not None:
            bounds = np.searchsorted(sel, np.arange(0, len(wl["probe"]["cols"][pn[0]]) + 1024, 1024, dtype=np.int64)).astype(np.uint64)
            offs = bounds[np.concatenate([[True], bounds[1:] != bounds[:-1]])]
        ref = orc.run_pipeline(list(wl["probe"]["cols"].values()), oj, paths, routing="adaptive_reinit", caching=False,
                               collect_output=False, sel=sel, chunk_offsets=offs)
        joins = capi.build_joins(gpu_ctx, wl, auto=True)
        cols = list(wl["probe"]["cols"].values())
        pipe = capi.Pipeline(gpu_ctx, cols, len(cols[0]), joins, paths)
        flt = wl["probe"].get("filter")
        if flt:
            _n, n_chunks = pipe.scan_filter([(pn.index(c), op, const) for c, op, const in flt])
        else:
            n_chunks = (len(cols[0]) + 1023) // 1024
        try:
            gpu_ctx.set_pool_tuning(share_after=share_after)
            mpx = capi.DeviceMultiplexer(pipe, "adaptive_reinit", max_log_rounds=1 << 16)
            if flt:
                mpx.use_scan_chunks()
            capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
            st = mpx.finish()
            _, _, inter = mpx.fetch_log()
        finally:
            gpu_ctx.set_pool_tuning()
        k = len(wl["joins"])
        assert list(inter) == list(ref["intermediates_per_round"]), name
        assert st["num_intermediates"] == ref["num_intermediates"], name
        assert sum(st["stage_out"][p][k - 1] for p in range(len(paths))) == ref["num_output_rows"], name
        mpx.close()
        pipe.close()
