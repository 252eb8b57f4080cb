"""polr_bcast_build -- the path's one exchange step (RCCL inside the product library) -- on the one GPU of the test box:
a communicator of world size 1 exercises id creation, ncclCommInitRank, the metadata / buffer walk and ncclBroadcast
calls end to end; the receiving side (polr_ht_alloc_like + in-place buffer broadcast) is the same code the N > 1 run
uses and is covered structurally by tests/test_gpu_edge_cases.py (export -> alloc_like -> identical probes)."""
import numpy as np
import pytest

from polr_amd import capi, workloads


@pytest.mark.gpu
def test_bcast_build_world_of_one(gpu_ctx):
    wl = workloads.star_skew(n_fact=50_000)
    joins = capi.build_joins(gpu_ctx, wl)
    uid = capi.comm_unique_id()
    assert len(uid) == capi.COMM_ID_BYTES and any(uid)
    comm = capi.Comm(gpu_ctx, uid, 1, 0)
    total = 0
    for ht, _ in joins:
        meta, bufs = ht.export()
        same = comm.bcast_build(ht, root=0)
        assert same is ht
        total += sum(int(b[1]) for b in bufs)
    assert comm.bytes_broadcast() == total
    # the tables still probe as before
    cols = list(wl["probe"]["cols"].values())
    pipe = capi.Pipeline(gpu_ctx, cols, len(cols[0]), joins, workloads.default_paths(len(joins)))
    counts = pipe.probe_rounds([(0, len(cols[0]), 0, 0)])
    assert counts.sum() > 0
    pipe.close()
    comm.close()


@pytest.mark.gpu
def test_bcast_build_keeps_a_packed_composite_key(gpu_ctx):
    """the per-column [min, max] packing of a three-key table travels in the metadata blob"""
    import numpy as np
    rng = np.random.default_rng(5)
    bk = [rng.integers(-40, 40, 3000).astype(np.int32), rng.integers(0, 9, 3000).astype(np.uint8),
          rng.integers(100, 130, 3000).astype(np.int16)]
    ht = capi.HashTable.from_columns(gpu_ctx, bk, [])
    ht.finalize_hash()
    meta, bufs = ht.export()
    clone = capi.HashTable.alloc_like(gpu_ctx, meta)
    _m2, bufs2 = clone.export()
    for (src, n), (dst, n2) in zip(bufs, bufs2):
        assert n == n2
        if n:
            _d2d(dst, src, n)
    pick = rng.integers(0, 3000, 8000)
    pk = [b[pick].copy() for b in bk]
    pk[0][::5] += 1000  # out of range: no match
    counts = []
    for t in (ht, clone):
        pipe = capi.Pipeline(gpu_ctx, pk, 8000, [(t, [(-1, 0), (-1, 1), (-1, 2)])], [[0]])
        counts.append(int(pipe.probe_rounds([(0, 8000, 0, 0)]).sum()))
        pipe.close()
    assert counts[0] == counts[1] > 0


def _d2d(dst, src, n):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst, src, n, 3) == 0


def test_comm_api_rejects_null_arguments():
    L = capi.load()
    assert L.polr_comm_get_unique_id(None) == capi.E_INVALID
    assert L.polr_bcast_build(None, None, 0, None) == capi.E_INVALID
