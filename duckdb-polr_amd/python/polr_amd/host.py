"""ctypes binding of the C++ host mirror (libpolr_host.so): the reference's operator classes
(PhysicalMultiplexer, RoutingStrategy, POLARConfig + enumerators, PhysicalHashJoin,
POLARPipelineExecutor) re-stated over the device C ABI.  Test plumbing for those classes."""
import ctypes as C
import os

import numpy as np

from . import capi

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_PKG))
LIB_PATH = os.path.join(_ROOT, "libpolr_host.so")

ENUMERATOR = {"dfs_random": 0, "dfs_min_card": 1, "dfs_uncertain": 2, "bfs_random": 3, "bfs_min_card": 4,
              "bfs_uncertain": 5, "each_last_once": 6, "each_first_once": 7, "sample": 8}


class RunResult(C.Structure):
    _fields_ = [("num_intermediates", C.c_uint64), ("n_rounds", C.c_uint64),
                ("input_tuple_count_per_path", C.c_uint64 * capi.MAX_PATHS),
                ("path_resistances", C.c_double * capi.MAX_PATHS)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    capi.load()  # libpolr_host.so links against libpolr_hip.so
    L = C.CDLL(LIB_PATH)
    vp, u64, P = C.c_void_p, C.c_uint64, C.POINTER
    L.polr_host_last_error.restype = C.c_char_p
    L.polr_host_mpx_create.restype = vp
    L.polr_host_mpx_create.argtypes = [C.c_int, C.c_int, C.c_double, u64, u64, C.c_int]
    L.polr_host_mpx_destroy.argtypes = [vp]
    L.polr_host_mpx_execute.argtypes = [vp, u64, P(u64), P(u64), P(u64), P(u64)]
    L.polr_host_mpx_add_intermediates.argtypes = [vp, u64]
    L.polr_host_mpx_increase_input.argtypes = [vp, u64]
    L.polr_host_mpx_set_skips.argtypes = [vp, u64]
    L.polr_host_mpx_finalize_path_run.argtypes = [vp]
    L.polr_host_mpx_resistances.argtypes = [vp, P(C.c_double)]
    L.polr_host_mpx_tuple_counts.argtypes = [vp, P(u64)]
    L.polr_host_mpx_log.restype = u64
    L.polr_host_mpx_log.argtypes = [vp, C.c_char_p, u64]
    L.polr_host_join_path_weights.argtypes = [P(C.c_double), C.c_int, C.c_double, P(C.c_double)]
    L.polr_host_generate_join_orders.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    L.polr_host_generate_join_orders_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp,
                                                    vp, vp, vp]
    L.polr_host_generate_join_orders_nested.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp,
                                                        vp, C.c_int, vp, vp, vp, vp]
    L.polr_host_run_pipeline.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_double, u64, u64, u64, vp, u64,
                                         C.c_int, vp, P(RunResult), vp, vp, vp, u64]
    L.polr_host_hash_join_probe.restype = C.c_int64
    L.polr_host_hash_join_probe.argtypes = [vp, vp, vp, u64, C.c_int, C.c_int64, C.c_int64, vp, vp, u64, vp, vp, u64,
                                            P(u64)]
    _lib = L
    return L


class HostMultiplexer:
    """PhysicalMultiplexer + MultiplexerState (host classes)"""

    def __init__(self, n_paths, routing, regret_budget=0.01, init_tuple_count=1024, atc_multiplier=1, log=True):
        self.L = load()
        self.P = n_paths
        r = capi.ROUTING[routing] if isinstance(routing, str) else routing
        self.h = self.L.polr_host_mpx_create(n_paths, r, regret_budget, init_tuple_count, atc_multiplier, int(log))
        if not self.h:
            raise RuntimeError(self.L.polr_host_last_error().decode())

    def execute(self, input_size):
        off, cnt, path, skips = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        more = self.L.polr_host_mpx_execute(self.h, input_size, C.byref(off), C.byref(cnt), C.byref(path),
                                            C.byref(skips))
        return bool(more), off.value, cnt.value, path.value, skips.value

    def add_intermediates(self, n):
        self.L.polr_host_mpx_add_intermediates(self.h, int(n))

    def increase_input(self, n):
        self.L.polr_host_mpx_increase_input(self.h, int(n))

    def set_skips(self, n):
        self.L.polr_host_mpx_set_skips(self.h, int(n))

    def finalize_path_run(self):
        self.L.polr_host_mpx_finalize_path_run(self.h)

    def resistances(self):
        out = (C.c_double * self.P)()
        self.L.polr_host_mpx_resistances(self.h, out)
        return list(out)

    def tuple_counts(self):
        out = (C.c_uint64 * self.P)()
        self.L.polr_host_mpx_tuple_counts(self.h, out)
        return list(out)

    def log_csv(self):
        n = self.L.polr_host_mpx_log(self.h, None, 0)
        buf = C.create_string_buffer(n)
        self.L.polr_host_mpx_log(self.h, buf, n)
        return buf.value.decode()

    def __del__(self):
        try:
            self.L.polr_host_mpx_destroy(self.h)
        except Exception:
            pass


def join_path_weights(costs, regret_budget):
    n = len(costs)
    c = (C.c_double * n)(*costs)
    w = (C.c_double * n)(*([1.0] * n))
    load().polr_host_join_path_weights(c, n, regret_budget, w)
    return list(w)


def generate_join_orders(enumerator, n_probe_cols, n_build_cols, cond_left_index, est_card, max_join_orders=8,
                         routing="adaptive_reinit", node_info=None, return_routing=False):
    """POLARConfig::GenerateJoinOrders on join shapes.  cond_left_index: per join the list of BoundReference
    indices of its conditions.  node_info (needed by 'sample'): k + 1 tuples (base_table_card, predicate, unique) --
    the pipeline's source, then the build side of every join.  Returns (paths [P,k], bindings [P,k,2],
    dependencies [k,k]) or None."""
    L = load()
    k = len(n_build_cols)
    nb = np.ascontiguousarray(n_build_cols, dtype=np.int32)
    nc = np.ascontiguousarray([len(c) for c in cond_left_index], dtype=np.int32)
    li = np.zeros((k, 2), dtype=np.int32)
    for j, c in enumerate(cond_left_index):
        li[j, :len(c)] = c
    card = np.ascontiguousarray(est_card, dtype=np.uint64)
    rows = max(max_join_orders, 24) + 2  # (Pipeline::Ready's BFS fallback enumerates up to 24 + 1 orders)
    paths = np.zeros((rows, k), dtype=np.int32)
    bind = np.zeros((rows, k, 2), dtype=np.int32)
    deps = np.zeros((k, k), dtype=np.uint8)
    eff = C.c_int32(capi.ROUTING[routing])
    ncard = nflags = nparent = None
    n_nodes = 0
    if node_info is not None:
        assert len(node_info) == k + 1
        # (base_table_card, predicate, unique[, nested]) -- nested: the join order of a build side that is itself a join
        # tree, as a list of the same tuples (its source first); flattened behind the k + 1 top-level nodes
        flat = [(x, -1) for x in node_info]
        at = 0
        while at < len(flat):
            x = flat[at][0]
            for inner in (x[3] if len(x) > 3 and x[3] else []):
                flat.append((inner, at))
            at += 1
        n_nodes = len(flat)
        ncard = np.ascontiguousarray([int(x[0][0]) for x in flat], dtype=np.uint64)
        nflags = np.ascontiguousarray([(1 if x[0][1] else 0) | (2 if x[0][2] else 0) for x in flat], dtype=np.uint8)
        nparent = np.ascontiguousarray([x[1] for x in flat], dtype=np.int32)
    n = L.polr_host_generate_join_orders_nested(ENUMERATOR[enumerator], capi.ROUTING[routing], k, n_probe_cols,
                                                nb.ctypes.data, nc.ctypes.data, li.ctypes.data, card.ctypes.data,
                                                max_join_orders, paths.ctypes.data, bind.ctypes.data, deps.ctypes.data,
                                                n_nodes, None if ncard is None else ncard.ctypes.data,
                                                None if nflags is None else nflags.ctypes.data,
                                                None if nparent is None else nparent.ctypes.data, C.byref(eff))
    if n < 0:
        raise RuntimeError(L.polr_host_last_error().decode())
    if n == 0:
        return None
    if return_routing:
        # the routing the multiplexer ends up with: DEFAULT_PATH when only Pipeline::Ready's BFS_MIN_CARD fallback
        # found a bank (pipeline.cpp:216-225)
        names = {v: k_ for k_, v in capi.ROUTING.items()}
        return paths[:n].copy(), bind[:n].copy(), deps, names[eff.value]
    return paths[:n].copy(), bind[:n].copy(), deps


def run_pipeline(pipe, paths, routing, n_tuples, regret_budget=0.01, init_tuple_count=1024, atc_multiplier=1,
                 chunk_offsets=None, device_routed=False, out=None, max_rounds=1 << 20):
    """POLARPipelineExecutor::Execute over a device pipeline (capi.Pipeline); device_routed: False = host-routed,
    True = self-routing launches (DEVICE_ROUTED), 2 = one resident launch (DEVICE_RESIDENT)"""
    L = load()
    paths = np.ascontiguousarray(np.asarray(paths, dtype=np.int32).reshape(-1, pipe.k))
    res = RunResult()
    rp = np.zeros((max_rounds,), dtype=np.uint32)
    rt = np.zeros((max_rounds,), dtype=np.uint64)
    ri = np.zeros((max_rounds,), dtype=np.uint64)
    co_p, n_chunks = None, 0
    if chunk_offsets is not None:
        chunk_offsets = np.ascontiguousarray(chunk_offsets, dtype=np.uint64)
        co_p, n_chunks = chunk_offsets.ctypes.data, len(chunk_offsets) - 1
    r = capi.ROUTING[routing] if isinstance(routing, str) else routing
    rc = L.polr_host_run_pipeline(pipe.ctx.h, pipe.h, pipe.k, len(paths), paths.ctypes.data, r, regret_budget,
                                  init_tuple_count, atc_multiplier, n_tuples, co_p, n_chunks, int(device_routed),
                                  out.h if out else None, C.byref(res), rp.ctypes.data, rt.ctypes.data,
                                  ri.ctypes.data, max_rounds)
    if rc != 0:
        raise RuntimeError(L.polr_host_last_error().decode())
    n = res.n_rounds
    P = len(paths)
    return {"num_intermediates": res.num_intermediates,
            "input_tuple_count_per_path": [res.input_tuple_count_per_path[i] for i in range(P)],
            "path_resistances": [res.path_resistances[i] for i in range(P)],
            "rounds": ri[:n].copy(), "round_path": rp[:n].copy(), "round_tuples": rt[:n].copy()}


def hash_join_probe(ctx, build_keys, build_payload, probe_keys, probe_valid=None, perfect=None):
    """PhysicalHashJoin (device-backed) driven chunk by chunk through Execute()"""
    L = load()
    bk = np.ascontiguousarray(build_keys, dtype=np.int32)
    bp = np.ascontiguousarray(build_payload, dtype=np.int32)
    pk = np.ascontiguousarray(probe_keys, dtype=np.int32)
    pv = None if probe_valid is None else np.ascontiguousarray(probe_valid, dtype=np.uint8)
    cap = max(1024, 64 * len(pk))
    rows = np.zeros((cap,), dtype=np.uint32)
    pay = np.zeros((cap,), dtype=np.int32)
    calls = C.c_uint64()
    pmin, pmax = (perfect if perfect else (0, 0))
    n = L.polr_host_hash_join_probe(ctx.h, bk.ctypes.data, bp.ctypes.data, len(bk), int(perfect is not None), pmin,
                                    pmax, pk.ctypes.data, None if pv is None else pv.ctypes.data, len(pk),
                                    rows.ctypes.data, pay.ctypes.data, cap, C.byref(calls))
    if n < 0:
        raise RuntimeError(L.polr_host_last_error().decode())
    if n > cap:
        raise RuntimeError("output larger than the test buffer")
    return rows[:n].copy(), pay[:n].copy(), calls.value


def hash_join_probe_keysem(ctx, build_keys, build_payload, probe_keys, build_key_valid=None, probe_valid=None, cast=False,
                           null_equal=False):
    """PhysicalHashJoin whose one condition is `CAST(probe AS BIGINT) = build` (cast: build_keys are int64) and / or
    `probe IS NOT DISTINCT FROM build` (null_equal), chunk by chunk through Execute()"""
    L = load()
    bk = np.ascontiguousarray(build_keys, dtype=np.int64 if cast else np.int32)
    bp = np.ascontiguousarray(build_payload, dtype=np.int32)
    pk = np.ascontiguousarray(probe_keys, dtype=np.int32)
    bv = None if build_key_valid is None else np.ascontiguousarray(build_key_valid, dtype=np.uint8)
    pv = None if probe_valid is None else np.ascontiguousarray(probe_valid, dtype=np.uint8)
    cap = max(1024, 64 * len(pk))
    rows = np.zeros((cap,), dtype=np.uint32)
    pay = np.zeros((cap,), dtype=np.int32)
    L.polr_host_hash_join_probe_keysem.restype = C.c_int64
    L.polr_host_hash_join_probe_keysem.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_int, C.c_int] + [C.c_void_p] * 2 + \
        [C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    n = L.polr_host_hash_join_probe_keysem(ctx.h, bk.ctypes.data, None if bv is None else bv.ctypes.data, bp.ctypes.data,
                                           len(bk), int(cast), int(null_equal), pk.ctypes.data,
                                           None if pv is None else pv.ctypes.data, len(pk), rows.ctypes.data,
                                           pay.ctypes.data, cap)
    if n < 0:
        raise RuntimeError(L.polr_host_last_error().decode())
    if n > cap:
        raise RuntimeError("output larger than the test buffer")
    return rows[:n].copy(), pay[:n].copy()


COMPARISON = {"<>": 26, "<": 27, ">": 28, "<=": 29, ">=": 30}  # ExpressionType, expression_type.hpp:34-46


def hash_join_probe_cond(ctx, build_keys, build_other, build_payload, op, probe_keys, probe_other):
    """PhysicalHashJoin with conditions [probe_keys = build_keys, probe_other OP build_other], chunk by chunk"""
    L = load()
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (build_keys, build_other, build_payload, probe_keys,
                                                              probe_other)]
    bk, bo, bp, pk, po = arrs
    cap = max(1024, 64 * len(pk))
    rows = np.zeros((cap,), dtype=np.uint32)
    pay = np.zeros((cap,), dtype=np.int32)
    L.polr_host_hash_join_probe_cond.restype = C.c_int64
    L.polr_host_hash_join_probe_cond.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_int] + [C.c_void_p] * 2 + \
        [C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    n = L.polr_host_hash_join_probe_cond(ctx.h, bk.ctypes.data, bo.ctypes.data, bp.ctypes.data, len(bk), COMPARISON[op],
                                         pk.ctypes.data, po.ctypes.data, len(pk), rows.ctypes.data, pay.ctypes.data, cap)
    if n < 0:
        raise RuntimeError(L.polr_host_last_error().decode())
    if n > cap:
        raise RuntimeError("output larger than the test buffer")
    return rows[:n].copy(), pay[:n].copy()
