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


@pytest.mark.parametrize("name,key", _cases())
def test_device_routing_matches_reference(gpu_ctx, name, key):
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
        mpx.run(a, b, out=out)
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


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_device_alternate_matches_reference(gpu_ctx, name):
    gold = common.load_golden(name)
    g = gold["alternate"]["each_last_once"]
    wl, paths, pipe, joins, n = pipeline_for(gpu_ctx, name, "each_last_once")
    mpx = capi.DeviceMultiplexer(pipe, "alternate", chunk_size=1024)
    out = capi.Output(pipe, 1024, 8192)
    mpx.run(0, (n + 1023) // 1024, out=out)
    st = mpx.finish()
    path, tuples, inter = mpx.fetch_log()
    P = len(paths)
    assert np.array_equal(inter.reshape(-1, P), np.asarray(g["matrix"], dtype=np.uint64))
    assert st["num_intermediates"] == g["intms"]
    n_rows, _, overflow = out.stats()
    assert not overflow and n_rows == g["n_rows"]  # only path 0 forwards its output


@pytest.mark.parametrize("routing", ["init_once", "opportunistic", "adaptive_reinit", "dynamic",
                                     "exponential_backoff", "default_path"])
def test_bench_workload_matches_reference(gpu_ctx, routing):
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
    mpx.run(0, len(offs) - 1)
    st = mpx.finish()
    path, tuples, inter = mpx.fetch_log()
    g = gold["routing"][routing]
    assert list(inter) == g["rounds"]
    assert st["num_intermediates"] == g["intms"]
    assert st["input_tuple_count_per_path"] == g["tuple_counts"]
    # COUNT(*): the last join's output over all paths
    k = 2
    assert sum(st["stage_out"][p][k - 1] for p in range(2)) == gold["count_star"]
