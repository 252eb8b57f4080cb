"""Source side of the pipeline: table scan with pushed-down filters.

CPU part: the oracle restatement of RowGroup::TemplatedScan / ColumnSegment::FilterSelection against
(1) the selection + chunk boundaries the JOB-light fixture is built from -- those reproduce the routing
traces the REFERENCE logged for the same filtered scan (tests/golden/job_light_01.json,
test_oracle_golden.test_job_light_bench_pipeline), which pins vector size, empty-vector skipping and row order --
and (2) a plain numpy statement of the predicate semantics (NULL passes no comparison).
GPU part (-m gpu): polr_pipeline_scan_filter == oracle, bit for bit, through the C ABI."""
import numpy as np
import pytest

import common
from common import orc, workloads

OPS = ["=", "!=", "<", ">", "<=", ">="]


def numpy_scan(cols, filters, V, valids=None):
    n = len(cols[0])
    keep = np.ones(n, dtype=bool)
    for col, op, const in filters:
        a = cols[col]
        valid = np.ones(n, dtype=bool) if valids is None or valids[col] is None else valids[col].astype(bool)
        if op == "is null":
            keep &= ~valid
            continue
        if op == "is not null":
            keep &= valid
            continue
        r = {"=": a == const, "!=": a != const, "<": a < const, ">": a > const, "<=": a <= const, ">=": a >= const}[op]
        keep &= valid & r
    sel = np.nonzero(keep)[0].astype(np.uint32)
    bounds = np.searchsorted(sel, np.arange(0, n + V, V, dtype=np.int64)).astype(np.uint64)
    if len(bounds) < 2:
        return sel, np.zeros(1, dtype=np.uint64)
    nonempty = np.concatenate([[True], bounds[1:] != bounds[:-1]])
    offs = bounds[nonempty]
    if len(offs) == 0 or offs[-1] != len(sel):
        offs = np.concatenate([offs, [len(sel)]]).astype(np.uint64)
    return sel, offs


def random_case(seed, n, dtype, with_nulls):
    rng = np.random.default_rng(seed)
    info = np.iinfo(dtype)
    lo, hi = max(info.min, -50), min(info.max, 50)
    a = rng.integers(lo, hi, size=n, endpoint=True).astype(dtype)
    b = rng.integers(0, 5, size=n).astype(np.uint8)
    va = (rng.random(n) > 0.2).astype(np.uint8) if with_nulls else None
    return [a, b], [va, None]


def test_oracle_matches_job_light_fixture_inputs():
    import bench
    wl = workloads.job_light_01(scale=0.1)
    cols = list(wl["probe"]["cols"].values())
    names = list(wl["probe"]["cols"].keys())
    sel, offs = orc.scan_filter(cols, [(names.index("company_type_id"), "=", 2)], vector_size=1024)
    want_sel = wl["probe"]["filter_sel"]
    assert np.array_equal(sel, want_sel)
    assert np.array_equal(offs, bench.chunk_offsets_for(want_sel, len(cols[0]), 1024))


@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64])
@pytest.mark.parametrize("with_nulls", [False, True])
def test_oracle_matches_numpy(dtype, with_nulls):
    for seed, (n, V) in enumerate([(0, 1024), (1, 1024), (1023, 1024), (1025, 1024), (5000, 64), (4097, 1000), (300, 2)]):
        cols, valids = random_case(seed, n, dtype, with_nulls)
        for op in OPS:
            const = 7 if np.iinfo(dtype).min == 0 else -3
            flt = [(0, op, const), (1, "!=", 3)]
            sel, offs = orc.scan_filter(cols, flt, vector_size=V, valids=valids)
            want_sel, want_offs = numpy_scan(cols, flt, V, valids)
            assert np.array_equal(sel, want_sel), (n, V, op)
            assert np.array_equal(offs, want_offs), (n, V, op)
        for op in ("is null", "is not null"):
            sel, offs = orc.scan_filter(cols, [(0, op, 0)], vector_size=V, valids=valids)
            want_sel, want_offs = numpy_scan(cols, [(0, op, 0)], V, valids)
            assert np.array_equal(sel, want_sel) and np.array_equal(offs, want_offs)


# ---- GPU -------------------------------------------------------------------------------------------
def _device_pipeline(ctx, cols, valids):
    from polr_amd import capi
    # a one-join pipeline over the columns (the scan does not care about the joins)
    keys = np.arange(16, dtype=np.int32)
    ht = capi.HashTable.from_columns(ctx, [keys], [])
    ht.finalize_hash()
    probe = [np.ascontiguousarray(c) for c in cols] + [np.zeros(len(cols[0]), dtype=np.int32)]
    pv = list(valids) + [None]
    pipe = capi.Pipeline(ctx, probe, len(cols[0]), [(ht, [(-1, len(cols))])], [[0]], probe_valid=pv)
    return pipe, ht


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64])
@pytest.mark.parametrize("with_nulls", [False, True])
def test_device_scan_matches_oracle(gpu_ctx, dtype, with_nulls):
    for seed, (n, V) in enumerate([(1, 1024), (1023, 1024), (1025, 1024), (70000, 1024), (5000, 64), (4097, 1000),
                                   (300, 2), (200000, 2048)]):
        cols, valids = random_case(seed, n, dtype, with_nulls)
        pipe, ht = _device_pipeline(gpu_ctx, cols, valids)
        const = 7 if np.iinfo(dtype).min == 0 else -3
        cases = [[(0, op, const), (1, "!=", 3)] for op in OPS] + [[(0, "is null", 0)], [(0, "is not null", 0)], [],
                                                                    [(0, "=", 49), (1, "=", 4), (0, ">", 0)]]
        for flt in cases:
            n_sel, n_chunks = pipe.scan_filter(flt, vector_size=V)
            sel, offs = pipe.fetch_scan()
            want_sel, want_offs = orc.scan_filter(cols, flt, vector_size=V, valids=valids)
            assert n_sel == len(want_sel) and n_chunks == len(want_offs) - 1, (n, V, flt)
            assert np.array_equal(sel, want_sel), (n, V, flt)
            assert np.array_equal(offs, want_offs), (n, V, flt)
        pipe.close()
        ht.close()


@pytest.mark.gpu
def test_device_scan_all_filtered_and_errors(gpu_ctx):
    from polr_amd import capi
    cols, valids = random_case(3, 5000, np.int32, False)
    pipe, ht = _device_pipeline(gpu_ctx, cols, valids)
    assert pipe.scan_filter([(0, ">", 1000)]) == (0, 0)  # nothing survives: no chunk at all
    sel, offs = pipe.fetch_scan()
    assert len(sel) == 0 and list(offs) == [0]
    with pytest.raises(capi.PolrError):
        pipe.scan_filter([(9, "=", 1)])  # no such column
    with pytest.raises(capi.PolrError):
        pipe.scan_filter([(1, "=", -1)])  # negative constant against an unsigned column
    pipe.close()
    ht.close()


@pytest.mark.gpu
@pytest.mark.parametrize("routing", ["adaptive_reinit", "dynamic"])
def test_pipeline_over_device_scan_matches_reference(gpu_ctx, routing):
    """bench.py's pipeline with the filter evaluated on the device: same routing trace and COUNT(*) as the
    reference's run of the SQL (golden), i.e. as with the host-computed selection"""
    from polr_amd import capi
    from test_oracle_golden import job_light_budget
    gold = common.load_golden("job_light_01")
    wl = workloads.job_light_01(scale=0.1)
    names = list(wl["probe"]["cols"].keys())
    cols = list(wl["probe"]["cols"].values())
    joins = capi.build_joins(gpu_ctx, wl)
    pipe = capi.Pipeline(gpu_ctx, cols, len(cols[0]), joins, [[0, 1], [1, 0]])
    n_sel, n_chunks = pipe.scan_filter([(names.index("company_type_id"), "=", 2)])
    assert n_sel == len(wl["probe"]["filter_sel"])
    for launch in ("rounds", "resident"):
        mpx = capi.DeviceMultiplexer(pipe, routing, regret_budget=job_light_budget(routing, len(cols[0])))
        mpx.use_scan_chunks()
        if launch == "resident":
            mpx.run_resident(0, n_chunks)
        else:
            mpx.run(0, n_chunks)
        st = mpx.finish()
        path, tuples, inter = mpx.fetch_log()
        g = gold["routing"][routing]
        assert list(inter) == g["rounds"]
        assert st["num_intermediates"] == g["intms"]
        assert st["input_tuple_count_per_path"] == g["tuple_counts"]
        assert sum(st["stage_out"][p][1] for p in range(2)) == gold["count_star"]
        mpx.close()
    pipe.close()


@pytest.mark.gpu
def test_device_scan_full_size_properties(gpu_ctx):
    """BASELINE configs[1] size: 2.6 M rows -- survivors ascending, exactly the rows that pass, chunk boundaries
    at the 1024-row vectors (size-independent properties; no oracle run needed)"""
    wl = workloads.job_light_01(scale=1.0)
    names = list(wl["probe"]["cols"].keys())
    cols = list(wl["probe"]["cols"].values())
    pipe, ht = _device_pipeline(gpu_ctx, cols, [None] * len(cols))
    ci = names.index("company_type_id")
    n_sel, n_chunks = pipe.scan_filter([(ci, "=", 2)])
    sel, offs = pipe.fetch_scan()
    assert n_sel == int((cols[ci] == 2).sum())
    assert np.all(np.diff(sel.astype(np.int64)) > 0)
    assert np.all(cols[ci][sel] == 2)
    assert offs[0] == 0 and offs[-1] == n_sel and np.all(np.diff(offs.astype(np.int64)) > 0)
    # every chunk lies inside one 1024-row vector, consecutive chunks in different vectors
    first = sel[offs[:-1].astype(np.int64)] // 1024
    last = sel[offs[1:].astype(np.int64) - 1] // 1024
    assert np.array_equal(first, last) and np.all(np.diff(first.astype(np.int64)) > 0)
    pipe.close()
    ht.close()
