"""IS NOT DISTINCT FROM join keys and CAST'ed join keys (JoinHashTable::null_values_are_equal,
src/execution/join_hashtable.cpp:35-36,170-192; polar_config.cpp:75-82) -- oracle and device against what the reference
itself did with polr_amd.workloads.key_semantics (tests/golden/key_semantics.json, made by tests/golden/make_golden_keysem.py):

  * fact JOIN dim_b ON b IS NOT DISTINCT FROM k JOIN dim_c JOIN dim_d: multiplexed by the reference -- ALTERNATE matrix,
    COUNT(*), totals and three routing traces are the reference's;
  * the same with dim_a ON CAST(fact.a AS BIGINT) = dim_a.k in front: the reference does NOT multiplex such a pipeline
    (recorded in the fixture with the reason); its COUNT(*) is what oracle and device return when they multiplex all four
    joins (device: POLR_KEY_BY_VALUE, no cast copy of the column).  Traces of that form: device vs oracle only
    (parity unpinned against the reference -- it has no such run).
"""
import numpy as np
import pytest

import common
from common import orc, workloads
from polr_amd import host

GOLD = common.load_golden("key_semantics")
TRACED = ["adaptive_reinit", "init_once", "opportunistic"]


def _paths(wl):
    return host.generate_join_orders("each_last_once", len(wl["probe"]["cols"]), [len(j["payload"]) for j in wl["joins"]],
                                     wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8)[0]


def _oracle(wl, paths, routing, collect_output=False):
    pcols, pvalid, ojoins = common.oracle_joins(wl)
    pcols = list(pcols)
    if wl["joins"][0]["name"] == "dim_a":
        pcols[1] = pcols[1].astype(np.int64)  # CAST(fact.a AS BIGINT): the oracle compares two BIGINTs, as the reference would
    return orc.run_pipeline(pcols, ojoins, paths, routing=routing, caching=False, collect_output=collect_output,
                            probe_valid=pvalid)


def test_oracle_not_distinct_from_matches_reference():
    wl = workloads.key_semantics(cast=False)
    paths = _paths(wl)
    assert np.asarray(paths).tolist() == GOLD["paths"]
    res = _oracle(wl, paths, "alternate")
    assert np.array_equal(res["alt_matrix"], np.asarray(GOLD["alternate"], dtype=np.uint64))
    assert res["num_output_rows"] == GOLD["count_star"] and res["num_intermediates"] == GOLD["alternate_intms"]
    for routing in TRACED:
        g = GOLD["traces"][routing]
        r = _oracle(wl, paths, routing)
        assert list(r["intermediates_per_round"]) == g["rounds"] and r["num_intermediates"] == g["intms"], routing
        assert r["input_tuple_count_per_path"][:len(paths)] == g["tuple_counts"], routing


def test_oracle_cast_key_count_matches_reference():
    """the reference runs this query without its multiplexer (fixture: reference_multiplexed false); the oracle multiplexes
    all four joins over the pre-cast column and must count the same rows on every join order"""
    assert GOLD["with_cast"]["reference_multiplexed"] is False
    wl = workloads.key_semantics(cast=True)
    paths = _paths(wl)
    assert np.asarray(paths).tolist() == GOLD["with_cast"]["paths"]
    res = _oracle(wl, paths, "alternate")
    assert res["num_output_rows"] == GOLD["with_cast"]["count_star"]
    for p in range(len(paths)):
        one = _oracle(wl, paths[p:p + 1], "default_path")
        assert one["num_output_rows"] == GOLD["with_cast"]["count_star"]


# ---- device ------------------------------------------------------------------------------------------------------------
def _device(gpu_ctx, wl, paths):
    from polr_amd import capi
    joins = capi.build_joins(gpu_ctx, wl)
    probe = wl["probe"]
    names = list(probe["cols"].keys())
    pv = [probe.get("valid", {}).get(n) for n in names]
    n = len(probe["cols"][names[0]])
    return capi.Pipeline(gpu_ctx, list(probe["cols"].values()), n, joins, paths, probe_valid=pv), n


@pytest.mark.gpu
@pytest.mark.parametrize("launch", ["rounds", "resident"])
def test_device_not_distinct_from_matches_reference(gpu_ctx, launch):
    from polr_amd import capi
    wl = workloads.key_semantics(cast=False)
    paths = _paths(wl)
    pipe, n = _device(gpu_ctx, wl, paths)
    n_chunks = (n + 1023) // 1024
    mpx = capi.DeviceMultiplexer(pipe, "alternate", chunk_size=1024)
    (mpx.run_resident if launch == "resident" else mpx.run)(0, n_chunks)
    st = mpx.finish()
    _, _, inter = mpx.fetch_log()
    assert np.array_equal(inter.reshape(-1, len(paths)), np.asarray(GOLD["alternate"], dtype=np.uint64))
    assert st["num_intermediates"] == GOLD["alternate_intms"]
    mpx.close()
    for routing in TRACED:
        g = GOLD["traces"][routing]
        mpx = capi.DeviceMultiplexer(pipe, routing, chunk_size=1024)
        (mpx.run_resident if launch == "resident" else mpx.run)(0, n_chunks)
        st = mpx.finish()
        _, _, inter = mpx.fetch_log()
        assert list(inter) == g["rounds"] and st["num_intermediates"] == g["intms"], routing
        assert st["input_tuple_count_per_path"] == g["tuple_counts"], routing
        k = len(wl["joins"])
        assert sum(st["stage_out"][p][k - 1] for p in range(len(paths))) == GOLD["count_star"]
        mpx.close()
    pipe.close()


@pytest.mark.gpu
def test_device_not_distinct_from_row_set(gpu_ctx):
    """the materialised join result (NULL keys joined to NULL keys included) against the oracle's, every join order"""
    from polr_amd import capi
    wl = workloads.key_semantics(cast=False)
    paths = np.asarray(_paths(wl))
    want = common.oracle_output_digest(wl, _oracle(wl, paths, "default_path", collect_output=True)["out_rows"])
    assert want[1] == GOLD["count_star"]
    for p in range(len(paths)):
        # DEFAULT_PATH runs path 0 of the bank: the bank with join order p in front
        bank = np.concatenate([paths[p:p + 1], np.delete(paths, p, axis=0)])
        pipe, n = _device(gpu_ctx, wl, bank)
        mpx = capi.DeviceMultiplexer(pipe, "default_path", chunk_size=1024)
        out = capi.Output(pipe, 1024, 4096)
        mpx.run_resident(0, (n + 1023) // 1024, out=out)
        mpx.finish()
        cols = []
        for src_join, arr, valid in common.output_columns(wl):
            src = wl["probe"]["cols"] if src_join < 0 else wl["joins"][src_join]["payload"]
            col_idx = [i for i, a in enumerate(src.values()) if a is arr][0]
            cols.append(out.materialize(src_join, col_idx, arr.dtype))
        assert common.rows_digest_from_columns(cols) == want, p
        mpx.close()
        pipe.close()


@pytest.mark.gpu
def test_device_cast_key_by_value(gpu_ctx):
    """POLR_KEY_BY_VALUE: an INTEGER probe column against a BIGINT build key (values no INTEGER holds, negative values, NULLs
    on the probe side), multiplexed with the other three joins: COUNT(*) is the reference's, traces and per-path counts are
    the oracle's (which reads a pre-cast BIGINT column)"""
    from polr_amd import capi
    wl = workloads.key_semantics(cast=True)
    paths = _paths(wl)
    pipe, n = _device(gpu_ctx, wl, paths)
    n_chunks = (n + 1023) // 1024
    k = len(wl["joins"])
    for routing in ["alternate"] + TRACED:
        ref = _oracle(wl, paths, routing)
        mpx = capi.DeviceMultiplexer(pipe, routing, chunk_size=1024)
        mpx.run_resident(0, n_chunks)
        st = mpx.finish()
        _, _, inter = mpx.fetch_log()
        assert list(inter) == list(ref["intermediates_per_round"]), routing
        assert st["num_intermediates"] == ref["num_intermediates"]
        if routing != "alternate":
            assert sum(st["stage_out"][p][k - 1] for p in range(len(paths))) == GOLD["with_cast"]["count_star"]
        mpx.close()
    pipe.close()


@pytest.mark.gpu
def test_key_flags_are_checked(gpu_ctx):
    from polr_amd import capi
    wl = workloads.key_semantics(cast=True)
    paths = _paths(wl)
    # a probe key of another width without POLR_KEY_BY_VALUE: refused, with the way out in the message
    wl["joins"][0]["key_flags"] = [0]
    with pytest.raises(capi.PolrError) as e:
        _device(gpu_ctx, wl, paths)
    assert e.value.code == capi.E_INVALID and "POLR_KEY_BY_VALUE" in str(e.value)
    # a perfect table cannot carry either flag; flags after finalize are refused
    keys = np.arange(100, dtype=np.int32)
    ht = capi.HashTable.from_columns(gpu_ctx, [keys], [keys])
    ht.set_key_flags(0, capi.KEY_NULL_EQUAL)
    with pytest.raises(capi.PolrError) as e:
        ht.finalize_perfect(0, 99)
    assert e.value.code == capi.E_UNSUPPORTED
    ht.finalize_hash()
    with pytest.raises(capi.PolrError):
        ht.set_key_flags(0, 0)
    ht.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cast,null_equal", [(True, False), (False, True), (True, True)])
def test_host_mirror_hash_join_key_semantics(gpu_ctx, cast, null_equal):
    """host/physical_hash_join.cpp with JoinCondition::left_is_cast / COMPARE_NOT_DISTINCT_FROM: the operator-level drop-in
    (chunk at a time through Execute) against a brute-force join by value"""
    rng = np.random.default_rng(5)
    nb, n = 700, 5000
    bk = rng.integers(-300, 300, nb).astype(np.int64)
    if cast:
        bk[::50] += 1 << 34  # values no INTEGER holds: they match nothing
    bv = (rng.random(nb) < 0.9).astype(np.uint8)
    bp = np.arange(nb, dtype=np.int32)
    pk = rng.integers(-350, 350, n).astype(np.int32)
    pv = (rng.random(n) < 0.9).astype(np.uint8)
    rows, pay = host.hash_join_probe_keysem(gpu_ctx, bk, bp, pk, build_key_valid=bv, probe_valid=pv, cast=cast,
                                            null_equal=null_equal)
    want = []
    by_key = {}
    for r in range(nb):
        by_key.setdefault(int(bk[r]) if bv[r] else None, []).append(r)
    for i in range(n):
        key = int(pk[i]) if pv[i] else None
        if key is None and not null_equal:
            continue
        want += [(i, r) for r in by_key.get(key, [])]
    assert sorted(zip(rows.tolist(), pay.tolist())) == sorted(want)
    assert len(want) > 1000
