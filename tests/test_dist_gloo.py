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


def test_two_rank_gloo_broadcast_and_reduction():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
