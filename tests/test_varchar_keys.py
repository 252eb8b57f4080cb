"""VARCHAR join keys (JoinHashTable::Hash over string_t keys + RowOperations::Match comparing
them) -- oracle and device against what the reference itself did with polr_amd.workloads.varchar_keys
(tests/golden/varchar_keys.json, made by tests/golden/make_golden_varchar.py: fact.s = dim_s.k, dim_s.t = dim_t.k, fact.c =
dim_c.k; NULL strings on both sides, repeated keys, inline and heap strings).

On the device a VARCHAR key arrives the way the reference handles it: the KEY column is the 64-bit hash of the string (the
engine computes it for the bucket anyway), the strings are a verifying condition (POLR_CMP_STR_EQ over 16-byte string
cells and their heaps).  The tests truncate the hashes to 12 bits, so that most candidates the hash finds are collisions
the comparison has to reject.  The oracle, which has no strings, joins on dictionary codes (equal strings <-> equal codes).
"""
import numpy as np
import pytest

import common
from common import orc, workloads
from polr_amd import host

GOLD = common.load_golden("varchar_keys")
TRACED = ["adaptive_reinit", "init_once", "opportunistic"]


def _paths(wl):
    return host.generate_join_orders("each_last_once", 3, [3, 2, 2], wl["cond_left_index"],
                                     [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8)[0]


def _oracle(wl, paths, routing, collect_output=False):
    pcols, pvalid, ojoins = common.oracle_joins(wl["codes"])
    return orc.run_pipeline(pcols, ojoins, paths, routing=routing, caching=False, collect_output=collect_output,
                            probe_valid=pvalid)


def test_oracle_on_dictionary_codes_matches_reference():
    wl = workloads.varchar_keys()
    paths = _paths(wl)
    assert np.asarray(paths).tolist() == GOLD["paths"]
    res = _oracle(wl, paths, "alternate")
    assert np.array_equal(res["alt_matrix"], np.asarray(GOLD["alternate"], dtype=np.uint64))
    assert res["num_output_rows"] == GOLD["count_star"] and res["num_intermediates"] == GOLD["alternate_intms"]
    for routing in TRACED:
        g = GOLD["traces"][routing]
        r = _oracle(wl, paths, routing)
        assert list(r["intermediates_per_round"]) == g["rounds"] and r["num_intermediates"] == g["intms"], routing


def _device(gpu_ctx, wl, paths):
    from polr_amd import capi
    joins = capi.build_joins(gpu_ctx, wl)
    probe = wl["probe"]
    names = list(probe["cols"].keys())
    cols = list(probe["cols"].values())
    pv = [probe.get("valid", {}).get(n) for n in names]
    pstr = [(capi.string_cells(v), probe.get("string_valid", {}).get(n)) for n, v in probe["strings"].items()]
    n = len(cols[0])
    pipe = capi.Pipeline(gpu_ctx, cols + [c for (c, _h), _v in pstr], n, joins, paths, probe_valid=pv + [v for _c, v in pstr])
    for i, ((_c, heap), _v) in enumerate(pstr):
        pipe.set_probe_heap(len(cols) + i, heap)
    return pipe, n


@pytest.mark.gpu
@pytest.mark.parametrize("hash_bits", [12, 64])
@pytest.mark.parametrize("launch", ["rounds", "resident"])
def test_device_varchar_keys_match_reference(gpu_ctx, launch, hash_bits):
    from polr_amd import capi
    wl = workloads.varchar_keys(hash_bits=hash_bits)
    paths = _paths(wl)
    pipe, n = _device(gpu_ctx, wl, paths)
    n_chunks = (n + 1023) // 1024
    k = len(wl["joins"])
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
        assert sum(st["stage_out"][p][k - 1] for p in range(len(paths))) == GOLD["count_star"]
        mpx.close()
    pipe.close()


@pytest.mark.gpu
def test_device_varchar_keys_row_set(gpu_ctx):
    """the materialised join result (fixed-width columns of all four tables) against the oracle's on dictionary codes; and
    what the verifying comparison is worth: with 12-bit hashes and no STR_EQ condition the join returns half as many rows again"""
    from polr_amd import capi
    wl = workloads.varchar_keys(hash_bits=12)
    paths = _paths(wl)
    ref = _oracle(wl, paths, "default_path", collect_output=True)
    cw = wl["codes"]
    rows = ref["out_rows"]  # [n, 1 + k]: probe row, build ids in original join order
    want = sorted(zip(rows[:, 0].tolist(), cw["joins"][0]["payload"]["ps"][rows[:, 1]].tolist(),
                      cw["joins"][1]["payload"]["pt"][rows[:, 2]].tolist()))
    pipe, n = _device(gpu_ctx, wl, paths)
    mpx = capi.DeviceMultiplexer(pipe, "adaptive_reinit", chunk_size=1024)
    out = capi.Output(pipe, 1024, 4096)
    mpx.run_resident(0, (n + 1023) // 1024, out=out)
    mpx.finish()
    ids, _ = out.materialize(-1, 0, np.int32)
    ps, _ = out.materialize(0, 0, np.int32)
    pt, _ = out.materialize(1, 0, np.int32)
    assert sorted(zip(ids.tolist(), ps.tolist(), pt.tolist())) == want and len(want) == GOLD["count_star"]
    mpx.close()
    pipe.close()
    # without the comparison: hash collisions join
    for j in wl["joins"]:
        j["preds"] = []
    pipe, n = _device(gpu_ctx, wl, paths)
    mpx = capi.DeviceMultiplexer(pipe, "default_path", chunk_size=1024)
    mpx.run_resident(0, (n + 1023) // 1024)
    st = mpx.finish()
    assert st["stage_out"][0][len(wl["joins"]) - 1] > 1.3 * GOLD["count_star"]  # (measured: 16 181 against 10 633)
    mpx.close()
    pipe.close()


@pytest.mark.gpu
def test_str_eq_is_checked(gpu_ctx):
    from polr_amd import capi
    wl = workloads.varchar_keys()
    paths = _paths(wl)
    wl["joins"][0]["preds"] = [("str_eq", (-1, 1), "k")]  # left side: the INTEGER column c
    with pytest.raises(capi.PolrError) as e:
        _device(gpu_ctx, wl, paths)
    assert e.value.code == capi.E_INVALID and "STR_EQ" in str(e.value)
