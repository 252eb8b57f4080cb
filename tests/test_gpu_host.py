"""The C++ host mirror driving the device: POLARPipelineExecutor (host-routed = literal RunPath
transcription with one launch per path run; device-routed = router kernel) and the chunk-at-a-time
PhysicalHashJoin::Execute drop-in, against the reference's golden traces and the oracle."""
import numpy as np
import pytest

import common
from common import orc, workloads
from polr_amd import capi, host
from test_gpu_mpx import pipeline_for
from test_gpu_probe import SCENARIOS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(SCENARIOS))
@pytest.mark.parametrize("routing", ["adaptive_reinit", "init_once", "opportunistic", "dynamic",
                                     "exponential_backoff", "default_path"])
@pytest.mark.parametrize("device_routed", [False, True, 2])  # host-routed, self-routing launches, resident
def test_executor_matches_reference(gpu_ctx, name, routing, device_routed):
    gold = common.load_golden(name)
    g = gold["routing"]["each_last_once/%s/nocache" % routing]
    wl, paths, pipe, joins, n = pipeline_for(gpu_ctx, name, "each_last_once")
    out = capi.Output(pipe, 1024, 8192)
    res = host.run_pipeline(pipe, paths, routing, n, device_routed=device_routed, out=out)
    assert list(res["rounds"]) == g["rounds"]
    assert res["num_intermediates"] == g["intms"]
    assert res["input_tuple_count_per_path"] == g["tuple_counts"]
    n_rows, _, overflow = out.stats()
    assert not overflow and n_rows == g["n_rows"]


def test_host_and_device_routing_agree_bitwise(gpu_ctx):
    """same decisions, same resistances (doubles compared exactly) from the host classes and the router
    kernel: they are one source compiled twice"""
    wl, paths, pipe, joins, n = pipeline_for(gpu_ctx, "star_skew", "each_last_once")
    for routing in ("adaptive_reinit", "dynamic", "exponential_backoff"):
        a = host.run_pipeline(pipe, paths, routing, n, device_routed=False)
        for placement in (True, 2):  # self-routing launches, resident launch
            b = host.run_pipeline(pipe, paths, routing, n, device_routed=placement)
            assert np.array_equal(a["rounds"], b["rounds"])
            assert np.array_equal(a["round_path"], b["round_path"])
            assert np.array_equal(a["round_tuples"], b["round_tuples"])
            assert a["path_resistances"] == b["path_resistances"]


@pytest.mark.parametrize("perfect", [False, True])
def test_physical_hash_join_execute_protocol(gpu_ctx, perfect):
    """PhysicalHashJoin::Execute chunk by chunk: same rows as the oracle's chained / perfect probe,
    NULL probe keys never match, duplicates come back over HAVE_MORE_OUTPUT calls"""
    rng = np.random.default_rng(11)
    if perfect:
        bk = rng.permutation(np.arange(100, 5100, dtype=np.int32))[:3000]
        rng_args = (100, 5099)
    else:
        bk = np.repeat(np.arange(0, 400_000, 131, dtype=np.int32), rng.integers(1, 30, size=3054))
        rng_args = None
    bp = (bk.astype(np.int64) * 3 % 1000).astype(np.int32) + np.arange(len(bk), dtype=np.int32) % 7
    pk = rng.integers(0, 6000 if perfect else 400_000, 5000).astype(np.int32)
    pv = (rng.random(5000) > 0.05).astype(np.uint8)
    rows, pay, calls = host.hash_join_probe(gpu_ctx, bk, bp, pk, pv, perfect=rng_args)
    ht = orc.HashTable([bk], [bp])
    if perfect:
        assert ht.make_perfect(*rng_args)
    ref = orc.run_pipeline([pk], [orc.JoinSpec(ht, [(-1, 0)])], [[0]], routing="default_path", probe_valid=[pv])
    want = ref["out_rows"]
    want_pairs = np.stack([want[:, 0], bp[want[:, 1]].astype(np.uint32)], 1)
    got_pairs = np.stack([rows, pay.astype(np.uint32)], 1)
    assert np.array_equal(got_pairs[np.lexsort(got_pairs.T[::-1])], want_pairs[np.lexsort(want_pairs.T[::-1])])
    assert not np.any(pv[rows] == 0)
    n_chunks = (len(pk) + 1023) // 1024
    if perfect:
        assert calls == n_chunks  # one Execute per chunk, NEED_MORE_INPUT (perfect_hash_join_executor.cpp:207)
    else:
        assert calls >= 2 * n_chunks  # >= 2 Execute calls per probed chunk (SURVEY 3.4 ii)


@pytest.mark.parametrize("op", ["<", ">=", "<>"])
def test_physical_hash_join_with_a_non_equality_condition(gpu_ctx, op):
    """conditions [=, OP] on the operator-level drop-in: the equality keys the table, the other condition is evaluated
    on every candidate pair (JoinHashTable::predicates); its build column is not part of the output"""
    rng = np.random.default_rng(12)
    bk = np.repeat(np.arange(0, 3000, 7, dtype=np.int32), rng.integers(1, 6, size=429))
    bo = rng.integers(-40, 40, len(bk)).astype(np.int32)
    bp = (np.arange(len(bk), dtype=np.int32) * 3) % 1009
    pk = rng.integers(0, 3100, 6000).astype(np.int32)
    po = rng.integers(-40, 40, 6000).astype(np.int32)
    rows, pay = host.hash_join_probe_cond(gpu_ctx, bk, bo, bp, op, pk, po)
    ht = orc.HashTable([bk], [bp, bo])
    ref = orc.run_pipeline([pk, po], [orc.JoinSpec(ht, [(-1, 0)], preds=[(op, (-1, 1), 1)])], [[0]],
                           routing="default_path")
    want = ref["out_rows"]
    want_pairs = np.stack([want[:, 0], bp[want[:, 1]].astype(np.uint32)], 1)
    got_pairs = np.stack([rows, pay.astype(np.uint32)], 1)
    assert len(got_pairs) == len(want_pairs) > 0
    assert np.array_equal(got_pairs[np.lexsort(got_pairs.T[::-1])], want_pairs[np.lexsort(want_pairs.T[::-1])])
