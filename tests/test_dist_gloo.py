"""The N>1 path on CPU: two gloo ranks exercise the build-side broadcast protocol, the probe-partition
seeding and the whole-job throughput reduction that bench.py uses over RCCL on the GPUs."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common  # noqa: F401  (sys.path)
from polr_amd import dist as pdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # a fake finalized table: metadata blob + three buffers, host memory standing in for HBM
    rng = np.random.default_rng(3)
    src = [rng.integers(0, 255, size=n, dtype=np.uint8) for n in (4096, 17, 100_003)]
    meta = b"POLR" + bytes(range(40))

    def wrap(buf):
        arr, _n = buf
        return torch.from_numpy(arr)

    def alloc_like(m):
        assert m == meta
        bufs = [(np.zeros(len(s), dtype=np.uint8), len(s)) for s in src]
        return {"meta": m, "bufs": bufs}, bufs

    exported = (meta, [(s.copy(), len(s)) for s in src]) if rank == 0 else None
    table, nbytes = pdist.broadcast_table(dist, torch, dev, rank, exported, alloc_like, wrap)
    ok = nbytes == sum(len(s) for s in src)
    if rank != 0:
        ok = ok and all(np.array_equal(b[0], s) for b, s in zip(table["bufs"], src))
    # throughput reduction: sum of tuples over max of times
    value, dt, tup = pdist.whole_job_throughput(dist, torch, dev, world, 1000 * (rank + 1), 0.5 * (rank + 1), 4)
    ok = ok and abs(tup - 3000) < 1e-9 and abs(dt - 1.0) < 1e-9 and abs(value - 12000) < 1e-6
    seeds = [pdist.probe_partition_seed(1337, r) for r in range(world)]
    ok = ok and len(set(seeds)) == world and seeds[0] == 1337
    ok = ok and pdist.shard_queries(7, world, rank) == [q for q in range(7) if q % world == rank]
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def _partition_worker(rank, world, port, ret):
    """the multi-GPU data path on CPU: every rank generates ITS contiguous partition of the SSB-skew lineorder (the same
    function bench.py calls on the device), runs the whole multiplexed pipeline over it with the checker (oracle), and
    the ranks reduce COUNT(*) and an order-insensitive digest of the output rows -- no collective on the data path"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polr_amd import ssb_skew
    gold = common.load_golden("ssb_skew_sample")
    case = gold["cases"]["q4.1/3"]
    shape = gold["shape"]
    paths = np.asarray(case["paths"], dtype=np.int32)
    lo, hi = pdist.probe_partition(shape["n_lo"], world, rank)
    wl = ssb_skew.workload("q4.1", rows=(lo, hi), **shape)
    pcols, pvalid, joins = common.oracle_joins(wl)
    res = common.orc.run_pipeline(pcols, joins, paths, routing="adaptive_reinit", caching=False, collect_output=True)
    rows = res["out_rows"][:, 0].astype(np.int64) + lo  # global lineorder row of every output tuple
    local = torch.tensor([len(rows), int(rows.sum() % (1 << 61)), int(np.bitwise_xor.reduce(rows)) if len(rows) else 0,
                          hi - lo], dtype=torch.int64)
    allv = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allv, local)
    ret[rank] = [v.tolist() for v in allv] + [[lo, hi]]
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_pipelines_union_equals_single_rank():
    """output row set = union of the per-GPU sets, bit-identical to the single-GPU set; total
    intermediates are partition-dependent (each rank explores on its own), COUNT(*) and the rows are not"""
    from polr_amd import ssb_skew
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_partition_worker, args=(world, port, ret), nprocs=world, join=True)
        got = dict(ret)
    gold = common.load_golden("ssb_skew_sample")
    case = gold["cases"]["q4.1/3"]
    shape = gold["shape"]
    wl = ssb_skew.workload("q4.1", **shape)
    pcols, pvalid, joins = common.oracle_joins(wl)
    res = common.orc.run_pipeline(pcols, joins, np.asarray(case["paths"], dtype=np.int32), routing="adaptive_reinit",
                                  caching=False, collect_output=True)
    rows = res["out_rows"][:, 0].astype(np.int64)
    assert len(rows) == case["count_star"]  # the single-rank answer is the reference's
    per_rank = got[0][:world]
    assert got[1][:world] == per_rank
    assert sum(v[0] for v in per_rank) == len(rows)
    assert sum(v[1] for v in per_rank) % (1 << 61) == int(rows.sum() % (1 << 61))
    x = 0
    for v in per_rank:
        x ^= v[2]
    assert x == int(np.bitwise_xor.reduce(rows))
    # the partitions tile the table
    assert got[0][world][0] == 0 and got[1][world][1] == shape["n_lo"] and got[0][world][1] == got[1][world][0]
    assert sum(v[3] for v in per_rank) == shape["n_lo"]


def test_probe_partition_tiles_the_table():
    for n, world in ((600_000_000, 8), (1_000_000, 3), (5000, 4), (1024, 2)):
        edges = [pdist.probe_partition(n, world, r) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        for a, b in zip(edges, edges[1:]):
            assert a[1] == b[0] and a[1] % 1024 == 0


def test_two_rank_gloo_broadcast_and_reduction():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
