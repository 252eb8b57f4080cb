"""Multi-GPU plumbing of the POLAR path (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm, "gloo" on CPU for the tests).

The path shards by probe partition: every rank owns a partition of the probe side and its own
multiplexer (one PipelineExecutor per thread in the reference, pipeline.cpp:145-174,
pipeline_executor.cpp:28-41), so there is NO collective on the data path.  The one exchange is the
build side: built once (rank 0) and broadcast, buffer by buffer, before probing starts.
Over xGMI a root-to-all broadcast is bound by one link (~153 GB/s per peer link): B / 153 GB/s.
"""
import numpy as np


def probe_partition_seed(base_seed, rank):
    """every rank generates / owns a different probe partition; rank 0 keeps the base seed so the
    1-GPU run is the same workload as the single-process one"""
    return base_seed if rank == 0 else base_seed + 7919 * rank


def probe_partition(n_total, world, rank, chunk_size=1024):
    """rows [lo, hi) of the probe table that rank `rank` of `world` owns: contiguous ranges in scan order (for SSB-skew:
    contiguous lo_orderkey ranges, so the skew phases stay intact per GPU -- SURVEY.md 8(e)), cut at source-chunk
    boundaries so that every rank sees whole chunks; the ranges tile [0, n_total) exactly"""
    lo = (rank * n_total) // world
    hi = ((rank + 1) * n_total) // world
    lo = (lo // chunk_size) * chunk_size
    hi = (hi // chunk_size) * chunk_size if rank + 1 < world else n_total
    return lo, hi


def broadcast_table(dist, torch, device, rank, exported, alloc_like, wrap):
    """Replicate one finalized build side from rank 0.

    exported:   on rank 0 (meta: bytes, [(ptr_or_array, nbytes)]) from HashTable.export(), else None
    alloc_like: meta bytes -> (table, [(ptr_or_array, nbytes)]) on the receiving ranks
    wrap:       (ptr_or_array, nbytes) -> uint8 tensor viewing that buffer on `device` (zero copy)
    returns (table or None on rank 0, bytes broadcast)
    """
    if rank == 0:
        meta, bufs = exported
        n_meta = torch.tensor([len(meta)], dtype=torch.int64, device=device)
    else:
        n_meta = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(n_meta, 0)
    if rank == 0:
        meta_t = torch.tensor(list(meta), dtype=torch.uint8, device=device)
    else:
        meta_t = torch.zeros(int(n_meta.item()), dtype=torch.uint8, device=device)
    dist.broadcast(meta_t, 0)
    table = None
    if rank != 0:
        table, bufs = alloc_like(bytes(meta_t.cpu().numpy().tobytes()))
    total = 0
    for buf in bufs:
        t = wrap(buf)
        dist.broadcast(t, 0)
        total += int(buf[1])
    return table, total


def whole_job_throughput(dist, torch, device, world, local_tuples, local_seconds, steps):
    """value of the bench contract: units all ranks processed / max-over-ranks time"""
    dt = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    tup = torch.tensor([float(local_tuples)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tup, op=dist.ReduceOp.SUM)
    return float(tup.item()) * steps / float(dt.item()), float(dt.item()), float(tup.item())


def shard_queries(n_queries, world, rank):
    """config 4 (full JOB: independent POLAR pipelines): query q runs on rank q % world"""
    return [q for q in range(n_queries) if q % world == rank]
