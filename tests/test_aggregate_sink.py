"""Sink side of the pipeline: ungrouped COUNT(*) / COUNT / SUM / MIN / MAX.

tests/golden/aggregates.json holds the REFERENCE's answers (tests/golden/make_golden_agg.py: its own SQL
with POLAR enabled) for every output column of the four parity scenarios, NULLs included.
CPU part: the oracle's pipeline + column materialisation reproduce those answers (exact Python integers).
GPU part (-m gpu): polr_out_aggregate over the device pipeline's row-id output gives the same numbers,
for every routing strategy's output, plus 128-bit / negative / all-NULL / empty cases against Python ints."""
import json
import os

import numpy as np
import pytest

import common
from common import orc, workloads

GOLD = json.load(open(os.path.join(common.GOLDEN, "aggregates.json")))
SCENARIOS = {
    "star_skew": lambda: workloads.star_skew(),
    "star_skew_nulls": lambda: workloads.star_skew(n_fact=60_000, with_nulls=True),
    "chain_dep": lambda: workloads.chain_dep(),
    "fanout": lambda: workloads.fanout(),
}


def exact(data, valid):
    """python-int aggregates of a materialised column (NULLs take no part)"""
    vals = [int(v) for v, ok in zip(data.tolist(), (valid.tolist() if valid is not None else [1] * len(data))) if ok]
    return {"count": len(vals), "sum": sum(vals) if vals else None, "min": min(vals) if vals else None,
            "max": max(vals) if vals else None}


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_oracle_reproduces_reference_aggregates(name):
    wl = SCENARIOS[name]()
    g = GOLD[name]
    pcols, pvalid, joins = common.oracle_joins(wl)
    k = len(joins)
    res = orc.run_pipeline(pcols, joins, [list(range(k))], routing="default_path", probe_valid=pvalid)
    rows = res["out_rows"]
    assert len(rows) == g["count_star"]
    for (src_join, arr, valid), cname in zip(common.output_columns(wl), g["out_cols"]):
        data, v = orc.materialize_column(rows, k, src_join, arr, valid)
        assert exact(data, v) == g["columns"][cname], cname


# ---- GPU -------------------------------------------------------------------------------------------
def _col_index(wl, src_join, arr):
    src = wl["probe"]["cols"] if src_join < 0 else wl["joins"][src_join]["payload"]
    return [i for i, a in enumerate(src.values()) if a is arr][0]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENARIOS))
@pytest.mark.parametrize("routing", ["adaptive_reinit", "default_path", "dynamic"])
def test_device_aggregates_match_reference(gpu_ctx, name, routing):
    from polr_amd import capi
    from test_gpu_probe import gpu_pipeline, scenario_paths
    wl = SCENARIOS[name]()
    g = GOLD[name]
    paths = scenario_paths(wl, "each_last_once")
    pipe, joins, n = gpu_pipeline(gpu_ctx, wl, paths)
    out = capi.Output(pipe, 1024, 8192)
    mpx = capi.DeviceMultiplexer(pipe, routing)
    mpx.run_resident(0, (n + 1023) // 1024, out=out)
    mpx.finish()
    specs, names = [("count_star", -1, 0)], []
    for (src_join, arr, valid), cname in zip(common.output_columns(wl), g["out_cols"]):
        ci = _col_index(wl, src_join, arr)
        for fn in ("count", "sum", "min", "max"):
            specs.append((fn, src_join, ci))
        names.append(cname)
    got = []
    for i in range(0, len(specs), 8):  # at most 8 aggregates per call
        got += out.aggregate(specs[i:i + 8])
    assert got[0] == g["count_star"]
    for i, cname in enumerate(names):
        cnt, s, mn, mx = got[1 + 4 * i:5 + 4 * i]
        assert {"count": cnt, "sum": s, "min": mn, "max": mx} == g["columns"][cname], cname
    mpx.close()
    pipe.close()


@pytest.mark.gpu
def test_device_aggregates_wide_values_and_nulls(gpu_ctx):
    """sums beyond 64 bits, negative values, every integer width, all-NULL and empty outputs"""
    from polr_amd import capi
    rng = np.random.default_rng(5)
    n = 300_000
    keys = np.arange(1000, dtype=np.int32)
    big = rng.integers(-2 ** 62, 2 ** 62, size=1000, dtype=np.int64)  # build payload: sums overflow int64
    small = rng.integers(-100, 100, size=1000).astype(np.int8)
    u16 = rng.integers(0, 65535, size=1000).astype(np.uint16)
    allnull = np.zeros(1000, dtype=np.int32)
    pv = [None, None, None, np.zeros(1000, dtype=np.uint8)]
    ht = capi.HashTable.from_columns(gpu_ctx, [keys], [big, small, u16, allnull], payload_valid=pv)
    ht.finalize_hash()
    fk = rng.integers(0, 1200, size=n).astype(np.int32)  # ~1/6 without a partner
    pcol_valid = (rng.random(n) > 0.1).astype(np.uint8)
    val = rng.integers(-2 ** 31, 2 ** 31 - 1, size=n).astype(np.int32)
    pipe = capi.Pipeline(gpu_ctx, [fk, val], n, [(ht, [(-1, 0)])], [[0]], probe_valid=[None, pcol_valid])
    out = capi.Output(pipe, 1024, 8192)  # (every emitting wave owns a partially filled chunk)
    pipe.probe_rounds([(0, n, 0, 1)], out=out)
    ids = out.fetch_ids()
    prow, brow = ids[:, 0].astype(np.int64), ids[:, 1].astype(np.int64)
    want = [len(ids)]
    specs = [("count_star", -1, 0)]
    for src_join, ci, data, valid in [(-1, 1, val[prow], pcol_valid[prow]), (0, 0, big[brow], None),
                                      (0, 1, small[brow], None), (0, 2, u16[brow], None),
                                      (0, 3, allnull[brow], np.zeros(len(brow), dtype=np.uint8))]:
        e = exact(data, valid)
        for fn in ("count", "sum", "min", "max"):
            specs.append((fn, src_join, ci))
            want.append(e[fn])
    got = []
    for i in range(0, len(specs), 8):
        got += out.aggregate(specs[i:i + 8])
    assert got == want
    assert abs(want[6]) > 2 ** 63  # the 128-bit path was exercised
    # empty output: COUNT = 0, everything else NULL
    out.reset()
    gpu_ctx.sync()
    assert out.aggregate([("count_star", -1, 0), ("sum", -1, 1), ("min", 0, 0), ("count", 0, 1)]) == [0, None, None, 0]
    with pytest.raises(capi.PolrError):
        out.aggregate([("sum", 0, 9)])
    pipe.close()
    ht.close()


# ---- BASELINE configs[0]: SSB Q1.1 shape, end to end (source scan filter -> join -> aggregate sink) -------------
Q11 = json.load(open(os.path.join(common.GOLDEN, "ssb_q11.json")))
Q11_AGGS = [("count_star", None), ("sum", "lo_extendedprice"), ("min", "lo_extendedprice"), ("max", "lo_extendedprice"),
            ("max", "lo_quantity"), ("min", "lo_discount"), ("sum", "lo_discount"), ("sum", "d_year"), ("count", "d_year")]


def test_oracle_ssb_q11_matches_reference():
    """filtered scan (oracle scan_filter) -> join (oracle pipeline) -> aggregates = the reference's answers"""
    wl = workloads.ssb_q11()
    names = list(wl["probe"]["cols"].keys())
    cols = list(wl["probe"]["cols"].values())
    flt = [(names.index(c), op, v) for c, op, v in wl["probe"]["filter"]]
    sel, offs = orc.scan_filter(cols, flt)
    assert len(sel) == Q11["filtered_rows"]
    pcols, pvalid, joins = common.oracle_joins(wl)
    res = orc.run_pipeline(pcols, joins, [[0]], routing="default_path", sel=sel, chunk_offsets=offs)
    rows = res["out_rows"]
    got = []
    for fn, cname in Q11_AGGS:
        if fn == "count_star":
            got.append(len(rows))
            continue
        if cname in wl["probe"]["cols"]:
            data, v = orc.materialize_column(rows, 1, -1, wl["probe"]["cols"][cname], None)
        else:
            data, v = orc.materialize_column(rows, 1, 0, wl["joins"][0]["payload"][cname], None)
        got.append(exact(data, v)[fn])
    assert got == Q11["values"]


@pytest.mark.gpu
def test_device_ssb_q11_matches_reference(gpu_ctx):
    """the same query on the device: polr_pipeline_scan_filter -> resident run -> polr_out_aggregate; only the
    nine aggregate values leave the GPU"""
    from polr_amd import capi
    wl = workloads.ssb_q11()
    names = list(wl["probe"]["cols"].keys())
    cols = list(wl["probe"]["cols"].values())
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    pipe = capi.Pipeline(gpu_ctx, cols, len(cols[0]), joins, [[0]])
    n_sel, n_chunks = pipe.scan_filter([(names.index(c), op, v) for c, op, v in wl["probe"]["filter"]])
    assert n_sel == Q11["filtered_rows"]
    out = capi.Output(pipe, 1024, 8192)
    mpx = capi.DeviceMultiplexer(pipe, "default_path")
    mpx.use_scan_chunks()
    capi.run_resident([mpx], [(0, n_chunks)], out=out, reset=True, finish=True)
    st = mpx.finish()
    assert st["stage_out"][0][0] == Q11["values"][0]
    specs = []
    for fn, cname in Q11_AGGS:
        if fn == "count_star":
            specs.append((fn, -1, 0))
        elif cname in wl["probe"]["cols"]:
            specs.append((fn, -1, names.index(cname)))
        else:
            specs.append((fn, 0, list(wl["joins"][0]["payload"].keys()).index(cname)))
    got = out.aggregate(specs[:8]) + out.aggregate(specs[8:])
    assert got == Q11["values"]
    mpx.close()
    pipe.close()


# ---- grouped sink: SSB Q4.1's GROUP BY d_year, c_nation ------------------------------------------------------
Q41 = json.load(open(os.path.join(common.GOLDEN, "ssb_q41_groups.json")))


def _q41_want():
    return {(r[0], r[1]): r[2:] for r in Q41["rows"]}


def test_oracle_ssb_q41_groups_match_reference():
    """oracle pipeline -> materialised columns -> python group-by = the reference's GROUP BY answer (175 groups)"""
    wl = workloads.ssb_skew_q41(sf=0.2)
    pcols, pvalid, joins = common.oracle_joins(wl)
    k = len(joins)
    res = orc.run_pipeline(pcols, joins, [list(range(k))], routing="default_path")
    rows = res["out_rows"]
    rev, _ = orc.materialize_column(rows, k, -1, wl["probe"]["cols"]["lo_revenue"], None)
    sup, _ = orc.materialize_column(rows, k, -1, wl["probe"]["cols"]["lo_supplycost"], None)
    nat, _ = orc.materialize_column(rows, k, 0, wl["joins"][0]["payload"]["c_nation"], None)
    yr, _ = orc.materialize_column(rows, k, 3, wl["joins"][3]["payload"]["d_year"], None)
    got = {}
    for y, c, r, s in zip(yr.tolist(), nat.tolist(), rev.tolist(), sup.tolist()):
        g = got.setdefault((y, c), [0, 0, 0, r, s, 0])
        g[0] += 1
        g[1] += r
        g[2] += s
        g[3] = min(g[3], r)
        g[4] = max(g[4], s)
        g[5] += r - s
    assert got == _q41_want()


@pytest.mark.gpu
@pytest.mark.parametrize("routing", ["adaptive_reinit", "default_path"])
def test_device_ssb_q41_groups_match_reference(gpu_ctx, routing):
    """the SSB Q4.1 sink on the device: GROUP BY d_year, c_nation over the multiplexed star join's row-id output;
    sum(lo_revenue - lo_supplycost) = sum(lo_revenue) - sum(lo_supplycost)"""
    from polr_amd import capi
    wl = workloads.ssb_skew_q41(sf=0.2)
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    names = list(wl["probe"]["cols"].keys())
    n = len(cols[0])
    paths = np.asarray(common.load_golden("ssb_skew_q41")["paths"], dtype=np.int32)
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    out = capi.Output(pipe, 1024, 8192)
    mpx = capi.DeviceMultiplexer(pipe, routing)
    capi.run_resident([mpx], [(0, (n + 1023) // 1024)], out=out, reset=True, finish=True)
    mpx.finish()
    years = sorted({r[0] for r in Q41["rows"]})
    y0, ny = years[0], years[-1] - years[0] + 1
    keys = [(3, 0, y0, ny), (0, 0, 0, 25)]  # d_year (payload 0 of join 3), c_nation (payload 0 of join 0)
    specs = [("count_star", -1, 0), ("sum", -1, names.index("lo_revenue")), ("sum", -1, names.index("lo_supplycost")),
             ("min", -1, names.index("lo_revenue")), ("max", -1, names.index("lo_supplycost"))]
    vals, counts, dropped = out.aggregate_grouped(keys, specs)
    assert dropped == 0
    want = _q41_want()
    seen = 0
    for g, v in enumerate(vals):
        key = (y0 + g // 25, g % 25)
        if key in want:
            w = want[key]
            assert v == w[:5], key
            assert v[1] - v[2] == w[5]  # the profit column of Q4.1
            seen += 1
        else:
            assert v[0] == 0 and v[1] is None and v[3] is None  # an empty group: absent from the reference's result
    assert seen == len(want)
    # a domain that is too narrow drops rows instead of writing outside the table
    vals2, _, dropped2 = out.aggregate_grouped([(3, 0, y0, 1), (0, 0, 0, 25)], specs[:1])
    assert dropped2 == sum(r[2] for r in Q41["rows"] if r[0] != y0)
    assert sum(v[0] for v in vals2) == sum(r[2] for r in Q41["rows"] if r[0] == y0)
    mpx.close()
    pipe.close()


def _q41_shipped_want():
    gold = common.load_golden("ssb_q41_groupby")
    return gold, {(r[0], r[1]): r[2] for r in gold["rows"]}


def test_oracle_q41_as_shipped_matches_reference():
    """benchmark/ssb-skew/queries/q4-1.sql with its own select list on the load.sql-exact instance (polr_amd/ssb_skew.py):
    the reference's GROUP BY d_year, c_nation answer (tests/golden/ssb_q41_groupby.json) from the oracle's pipeline output"""
    from polr_amd import ssb_skew
    gold, want = _q41_shipped_want()
    wl = ssb_skew.workload("q4.1", **gold["shape"])
    inst = wl["instance"]
    m = inst.lineorder(0, inst.n_lo, cols=["lo_revenue", "lo_supplycost"])
    pcols, pvalid, joins = common.oracle_joins(wl)
    k = len(joins)
    res = orc.run_pipeline(pcols, joins, [list(range(k))], routing="default_path")
    rows = res["out_rows"]
    rev, _ = orc.materialize_column(rows, k, -1, m["lo_revenue"], None)
    sup, _ = orc.materialize_column(rows, k, -1, m["lo_supplycost"], None)
    nat, _ = orc.materialize_column(rows, k, 0, wl["joins"][0]["payload"]["c_nation"], None)
    yr, _ = orc.materialize_column(rows, k, 3, wl["joins"][3]["payload"]["d_year"], None)
    got = {}
    for y, c, r, s in zip(yr.tolist(), nat.tolist(), rev.tolist(), sup.tolist()):
        got[(y, c)] = got.get((y, c), 0) + r - s
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("n_exec", [1, 16])
def test_device_q41_as_shipped_on_the_flat_pipeline(gpu_ctx, n_exec):
    """Q4.1 with its own sink, on the device: the FLAT pipeline (bit tables in LDS) emits the surviving tuples' row ids
    -- probe row + per join the key's offset in its perfect table -- and the perfect-hash aggregate groups them by
    d_year, c_nation; profit = SUM(lo_revenue) - SUM(lo_supplycost).  Against the reference's answer; and the emitted row
    set against numpy"""
    from polr_amd import capi, ssb_skew
    gold, want = _q41_shipped_want()
    wl = ssb_skew.workload("q4.1", **gold["shape"])
    inst = wl["instance"]
    m = inst.lineorder(0, inst.n_lo, cols=["lo_revenue", "lo_supplycost"])
    names = list(wl["probe"]["cols"].keys()) + ["lo_revenue", "lo_supplycost"]
    cols = list(wl["probe"]["cols"].values()) + [m["lo_revenue"], m["lo_supplycost"]]
    n = len(cols[0])
    paths = np.asarray(common.load_golden("ssb_skew_sample")["cases"]["q4.1/3"]["paths"], dtype=np.int32)
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    assert pipe.launch_info(True)["flat"] == 1, "an emitting run over perfect tables must take the flat pipeline"
    out = capi.Output(pipe, 1024, 16384)
    n_chunks = (n + 1023) // 1024
    mpxs = [capi.DeviceMultiplexer(pipe, "adaptive_reinit") for _ in range(n_exec)]
    capi.run_resident(mpxs, [((e * n_chunks) // n_exec, ((e + 1) * n_chunks) // n_exec) for e in range(n_exec)], out=out,
                      reset=True, finish=True)
    stats = capi.finish_many(mpxs)
    k = len(wl["joins"])
    ids = out.fetch_ids()
    assert len(ids) == sum(sum(st["stage_out"][p][k - 1] for p in range(len(paths))) for st in stats) == sum(
        1 for _ in ids)
    # the emitted row set: exactly the probe rows whose four keys are on their build sides, each once
    keep = np.ones(n, dtype=bool)
    for j in wl["joins"]:
        keep &= np.isin(cols[j["key_src"][0][1]], j["keys"][0])
    assert np.array_equal(np.sort(ids[:, 0]), np.nonzero(keep)[0].astype(np.uint32))
    for x, j in enumerate(wl["joins"]):  # slot 1 + j: the key's offset in join j's perfect table
        assert np.array_equal(ids[:, 1 + x].astype(np.int64), cols[j["key_src"][0][1]][ids[:, 0]].astype(np.int64) - int(j["keys"][0].min()))
    years = sorted({y for y, _c in want})
    y0, ny = years[0], years[-1] - years[0] + 1
    keys = [(3, 0, y0, ny), (0, 0, 0, 50)]  # d_year (payload 0 of join 3), c_nation (payload 0 of join 0)
    specs = [("count_star", -1, 0), ("sum", -1, names.index("lo_revenue")), ("sum", -1, names.index("lo_supplycost"))]
    vals, counts, dropped = out.aggregate_grouped(keys, specs)
    assert dropped == 0
    seen = 0
    for g_, v in enumerate(vals):
        key = (y0 + g_ // 50, g_ % 50)
        if key in want:
            assert v[1] - v[2] == want[key], key
            seen += 1
        else:
            assert v[0] == 0
    assert seen == len(want) and sum(v[0] for v in vals) == len(ids)
    for m_ in mpxs:
        m_.close()
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_exec", [1, 16])
def test_device_q41_as_shipped_with_the_sink_fused_into_the_run(gpu_ctx, n_exec):
    """the same query with the GROUP BY FUSED into the flat pipeline's last join (polr_out_fuse_grouped): no row id is
    written, the group cells are the only output.  Against the reference's answer and against the two-kernel path; passes
    add up until the output object is reset; what a fused sink cannot do is refused"""
    from polr_amd import capi, ssb_skew
    gold, want = _q41_shipped_want()
    wl = ssb_skew.workload("q4.1", **gold["shape"])
    inst = wl["instance"]
    m = inst.lineorder(0, inst.n_lo, cols=["lo_revenue", "lo_supplycost"])
    names = list(wl["probe"]["cols"].keys()) + ["lo_revenue", "lo_supplycost"]
    cols = list(wl["probe"]["cols"].values()) + [m["lo_revenue"], m["lo_supplycost"]]
    n = len(cols[0])
    paths = np.asarray(common.load_golden("ssb_skew_sample")["cases"]["q4.1/3"]["paths"], dtype=np.int32)
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    years = sorted({y for y, _c in want})
    y0, ny = years[0], years[-1] - years[0] + 1
    keys = [(3, 0, y0, ny), (0, 0, 0, 50)]
    specs = [("count_star", -1, 0), ("sum", -1, names.index("lo_revenue")), ("sum", -1, names.index("lo_supplycost"))]
    out = capi.Output(pipe, 1024, 64)  # (no room for the row ids: none are written)
    out.fuse_grouped(keys, specs)
    n_chunks = (n + 1023) // 1024
    mpxs = [capi.DeviceMultiplexer(pipe, "adaptive_reinit") for _ in range(n_exec)]
    ranges = [((e * n_chunks) // n_exec, ((e + 1) * n_chunks) // n_exec) for e in range(n_exec)]
    for passes in (1, 2):
        capi.run_resident(mpxs, ranges, out=out, reset=True, finish=True)
        stats = capi.finish_many(mpxs)
        vals, counts, dropped = out.fused_result()
        assert dropped == 0
        k = len(wl["joins"])
        n_out = sum(sum(st["stage_out"][p][k - 1] for p in range(len(paths))) for st in stats)
        seen = 0
        for g_, v in enumerate(vals):
            key = (y0 + g_ // 50, g_ % 50)
            if key in want:
                assert v[1] - v[2] == passes * want[key], key
                seen += 1
            else:
                assert v[0] == 0 and v[1] is None
        assert seen == len(want) and sum(v[0] for v in vals) == passes * n_out
        assert out.stats()[0] == 0  # nothing was emitted
    out.reset()
    assert sum(v[0] for v in out.fused_result()[0]) == 0
    # MIN / MAX and 8-byte SUMs stay with the two-kernel path; a second sink on the same object is refused
    out2 = capi.Output(pipe, 1024, 64)
    with pytest.raises(capi.PolrError) as e:
        out2.fuse_grouped(keys, [("min", -1, names.index("lo_revenue"))])
    assert e.value.code == capi.E_UNSUPPORTED
    with pytest.raises(capi.PolrError):
        out.fuse_grouped(keys, specs)
    out.fuse_grouped(None, None)
    for m_ in mpxs:
        m_.close()
    pipe.close()


@pytest.mark.gpu
def test_string_min_max_over_inline_and_heap_cells(gpu_ctx):
    """polr_out_aggregate_string over string_t cells: strings of 0 .. 40 bytes (inline up to 12, heap beyond), many sharing
    their first 4 / 12 / 20 bytes, bytes above 0x7F (compared unsigned), proper prefixes (the shorter first), NULL cells (take
    no part), probe-side and build-side columns; MIN and MAX against python's bytes order (the reference's string order:
    memcmp over the common length, then the length -- src/include/duckdb/common/types/string_type.hpp)"""
    from polr_amd import capi
    rng = np.random.default_rng(11)
    stems = [b"", b"a", b"abcd", b"abcdefghijkl", b"abcdefghijklmnopqrst", b"abce", b"\xc3\xa9t\xc3\xa9", b"abcdefghijklm"]
    n_build = 3000

    def rand_str():
        s_ = stems[rng.integers(0, len(stems))]
        return s_ + bytes(rng.integers(97, 100, rng.integers(0, 21)).astype(np.uint8))

    bvals = [rand_str() for _ in range(n_build)]
    bkeys = rng.permutation(np.arange(1, n_build + 1, dtype=np.int32))
    bvalid = (rng.random(n_build) > 0.2).astype(np.uint8)
    cells, heap = capi.string_cells(bvals)
    ht = capi.HashTable.from_columns(gpu_ctx, [bkeys], [cells], payload_valid=[bvalid])
    ht.set_payload_heap(0, heap)
    ht.finalize_hash()
    n = 50_000
    pk = rng.integers(1, 2 * n_build, n).astype(np.int32)
    pvals = [rand_str() for _ in range(n)]
    pcells, pheap = capi.string_cells(pvals)
    pipe = capi.Pipeline(gpu_ctx, [pk, pcells], n, [(ht, [(-1, 0)])], [[0]])
    pipe.set_probe_heap(1, pheap)
    out = capi.Output(pipe, 1024, 8192)
    pipe.probe_rounds([(0, n, 0, 1)], out=out)
    ids = out.fetch_ids()
    assert len(ids) > 10_000
    build_seen = [bvals[i] for i in ids[:, 1].tolist() if bvalid[i]]
    probe_seen = [pvals[i] for i in ids[:, 0].tolist()]
    assert out.aggregate_string("min", 0, 0) == min(build_seen) and out.aggregate_string("max", 0, 0) == max(build_seen)
    assert out.aggregate_string("min", -1, 1) == min(probe_seen) and out.aggregate_string("max", -1, 1) == max(probe_seen)
    # no row at all: SQL NULL
    out2 = capi.Output(pipe, 1024, 64)
    pipe.probe_rounds([(0, 0, 0, 1)], out=out2)
    assert out2.aggregate_string("min", 0, 0) is None
    with pytest.raises(capi.PolrError):
        out.aggregate_string("min", -1, 0)  # not a VARCHAR column
    pipe.close()
    ht.close()


@pytest.mark.gpu
def test_general_hash_aggregate(gpu_ctx):
    """GROUP BY over group columns of any domain (PhysicalHashAggregate): a hash table of groups on the device.  Group
    columns from the probe table and from two build sides, one with NULLs (NULL is a group of its own), wide and negative
    key values, thousands of groups and a handful; COUNT(*), COUNT, SUM, MIN, MAX against a numpy GROUP BY over the oracle's
    join result; more groups than the caller made room for -> POLR_E_OVERFLOW"""
    from polr_amd import capi
    wl = workloads.star_skew(n_fact=150_000, with_nulls=True)  # (NULL keys and NULL payload cells)
    k = len(wl["joins"])
    paths = workloads.default_paths(k, "each_last_once")
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    n = len(pcols[0])
    ref = orc.run_pipeline(pcols, ojoins, paths[:1], routing="default_path", collect_output=True, probe_valid=pvalid)
    rows = ref["out_rows"]
    joins = capi.build_joins(gpu_ctx, wl)
    pipe = capi.Pipeline(gpu_ctx, pcols, n, joins, paths, probe_valid=pvalid)
    out = capi.Output(pipe, 1024, len(rows) // 1024 + 4096)
    mpx = capi.DeviceMultiplexer(pipe, "default_path")
    mpx.run_resident(0, (n + 1023) // 1024, out=out)
    mpx.finish()
    assert out.stats()[0] == len(rows)
    names = list(wl["probe"]["cols"].keys())
    j0 = wl["joins"][0]
    p0 = list(j0["payload"].keys())
    # group columns: probe column 1 (a key column: thousands of values), payload 0 of join 0 (with its NULLs, if any)
    gcols = [(-1, 1), (0, 0)]
    specs = [("count_star", -1, 0), ("sum", -1, 0), ("min", -1, 0), ("max", 0, 0), ("count", 0, 0)]
    got = out.aggregate_hashed(gcols, specs, 1 << 16)

    def col(sj, sc):
        if sj < 0:
            return orc.materialize_column(rows, k, -1, pcols[sc], pvalid[sc] if pvalid else None)
        arr = list(wl["joins"][sj]["payload"].values())[sc]
        val = wl["joins"][sj].get("payload_valid", {}).get(list(wl["joins"][sj]["payload"].keys())[sc])
        return orc.materialize_column(rows, k, sj, arr, val)

    g0, v0 = col(*gcols[0])
    g1, v1 = col(*gcols[1])
    a0, av0 = col(-1, 0)
    a1, av1 = col(0, 0)
    want = {}
    for i in range(len(rows)):
        key = (None if v0 is not None and not v0[i] else int(g0[i]), None if v1 is not None and not v1[i] else int(g1[i]))
        w = want.setdefault(key, [0, 0, None, None, 0])
        w[0] += 1
        w[1] += int(a0[i])
        w[2] = int(a0[i]) if w[2] is None else min(w[2], int(a0[i]))
        if av1 is None or av1[i]:
            w[3] = int(a1[i]) if w[3] is None else max(w[3], int(a1[i]))
            w[4] += 1
    assert len(got) == len(want) and len(want) > 1000
    for key, w in want.items():
        assert got[key] == w, key
    # one group column
    one = out.aggregate_hashed([(0, 0)], [("count_star", -1, 0)], 1 << 16)
    assert sum(v[0] for v in one.values()) == len(rows) and set(one) == {(k2[1],) for k2 in want}
    with pytest.raises(capi.PolrError) as e:
        out.aggregate_hashed(gcols, specs, 16)
    assert e.value.code == capi.E_OVERFLOW
    mpx.close()
    pipe.close()
