"""The C++ host mirror (duckdb-polr_amd/host: PhysicalMultiplexer, RoutingStrategy family,
POLARConfig + enumerators) on the CPU: driven with the intermediates the oracle computes per tuple,
it must reproduce the routing traces the reference logged (tests/golden), decision for decision."""
import numpy as np
import pytest

import common
from common import orc
from polr_amd import host
from test_oracle_golden import SCENARIOS, scenario, scenario_paths

V = 1024
_inter = {}


def per_tuple_intermediates(name, enumerator):
    """inter[p][t] = intermediates tuple t produces on join order p (oracle, ALTERNATE over 1-tuple chunks)
    as prefix sums, so any routed slice is two lookups"""
    key = (name, enumerator)
    if key not in _inter:
        wl, pcols, pvalid, joins, gold = scenario(name)
        paths = scenario_paths(wl, enumerator)
        n = len(pcols[0])
        res = orc.run_pipeline(pcols, joins, paths, routing="alternate", caching=False, probe_valid=pvalid,
                               collect_output=False, chunk_offsets=np.arange(n + 1, dtype=np.uint64))
        m = res["alt_matrix"].astype(np.int64)
        assert m.shape == (n, len(paths))
        _inter[key] = (np.concatenate([np.zeros((1, m.shape[1]), dtype=np.int64), np.cumsum(m, axis=0)]), paths, n)
    return _inter[key]


def _cases():
    out = []
    for name in SCENARIOS:
        gold = common.load_golden(name)
        for key in gold["routing"]:
            parts = key.split("/")
            if len(parts) == 3 and parts[2] == "cache":
                continue
            out.append((name, key))
    return out


@pytest.mark.parametrize("name,key", _cases())
def test_host_multiplexer_replays_reference_trace(name, key):
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
    prefix, paths, n = per_tuple_intermediates(name, enumerator)
    if routing == "exponential_backoff":
        kw["regret_budget"] = n / 10240.0 / 10 / 1
    mpx = host.HostMultiplexer(len(paths), routing, **kw)
    # the routing loop: path of a bypass chunk = the current path of the multiplexer
    n_chunks = (n + V - 1) // V
    c, skips, in_process, cur_path = 0, 0, False, 0
    while c < n_chunks:
        c0, size = c * V, min(V, n - c * V)
        if skips > 0 and not in_process:
            take = min(skips, n_chunks - c)
            begin, end = c0, min(n, (c + take) * V)
            mpx.increase_input(end - begin)
            if skips != 2**64 - 1:
                skips -= take
            mpx.set_skips(skips)
            c += take
        else:
            more, off, cnt, cur_path, skips = mpx.execute(size)
            begin, end = c0 + off, c0 + off + cnt
            in_process = more
            if not more:
                c += 1
        mpx.add_intermediates(int(prefix[end, cur_path] - prefix[begin, cur_path]))
    mpx.finalize_path_run()
    rounds = [int(x) for x in mpx.log_csv().strip().splitlines()[1:]]
    assert rounds == g["rounds"]
    assert mpx.tuple_counts() == g["tuple_counts"]
    assert sum(rounds) == g["intms"]


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_host_alternate_log_matches_reference_matrix(name):
    gold = common.load_golden(name)
    g = gold["alternate"]["each_last_once"]
    prefix, paths, n = per_tuple_intermediates(name, "each_last_once")
    mpx = host.HostMultiplexer(len(paths), "alternate")
    for c in range((n + V - 1) // V):
        c0, size = c * V, min(V, n - c * V)
        more = True
        while more:
            more, off, cnt, path, _ = mpx.execute(size)
            assert (off, cnt) == (0, size)
            mpx.add_intermediates(int(prefix[c0 + size, path] - prefix[c0, path]))
    mpx.finalize_path_run()
    lines = mpx.log_csv().strip().splitlines()
    assert lines[0] == "".join("path_%d," % i for i in range(len(paths)))  # physical_multiplexer.cpp:197-200
    matrix = [[int(x) for x in l.rstrip(",").split(",")] for l in lines[1:]]
    assert matrix == g["matrix"]


def test_join_path_weights_match_oracle():
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(2, 9))
        costs = list(np.round(rng.uniform(0.5, 6.0, n), int(rng.integers(1, 6))))
        b = float(rng.choice([0.01, 0.1, 0.2, 0.5]))
        assert host.join_path_weights(costs, b) == orc.join_path_weights(costs, b)


@pytest.mark.parametrize("name", list(SCENARIOS))
@pytest.mark.parametrize("enumerator", ["each_last_once", "each_first_once", "dfs_min_card", "bfs_min_card"])
def test_polar_config_join_orders_and_bindings(name, enumerator):
    """POLARConfig::GenerateJoinOrders (dependencies, enumerators, left_expression_bindings) against the
    oracle's restatement on every scenario's join shapes"""
    wl, pcols, pvalid, joins, gold = scenario(name)
    k = len(wl["joins"])
    n_probe = len(wl["probe"]["cols"])
    n_build = [len(j["payload"]) for j in wl["joins"]]
    # BoundReference index of every condition in the original layout
    offsets = np.concatenate([[n_probe], n_probe + np.cumsum(n_build)])
    cond = []
    for j in wl["joins"]:
        cond.append([sc if sj < 0 else int(offsets[sj]) + sc for sj, sc in j["key_src"]])
    card = [len(j["keys"][0]) for j in wl["joins"]]
    got = host.generate_join_orders(enumerator, n_probe, n_build, cond, card, 8)
    want_paths = scenario_paths(wl, enumerator)
    if got is None:
        assert len(want_paths) < 2
        return
    paths, bind, deps = got
    assert np.array_equal(paths, want_paths)
    want_bind = orc.bindings(n_probe, n_build, joins, want_paths)
    assert np.array_equal(bind, want_bind[:, :, :2])
    for i, j in enumerate(wl["joins"]):
        for sj, _ in j["key_src"]:
            if sj >= 0:
                assert deps[i, sj] == 1
    if enumerator in ("each_last_once", "each_first_once"):
        g = gold["alternate"][enumerator]
        assert g is None or len(g["matrix"][0]) == len(paths)
