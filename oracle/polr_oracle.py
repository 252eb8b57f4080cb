"""oracle/polr_oracle.py -- ctypes front end of the CPU restatement (oracle/polr_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; nothing under duckdb-polr_amd/ imports it.  The product path never runs
through this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpolr_oracle.so")

MAX_JOINS, MAX_PATHS, MAX_KEYS = 16, 64, 4
MAX_PREDS = 4
CMP_PRED = {"<>": 1, "!=": 1, "<": 2, ">": 3, "<=": 4, ">=": 5}

ROUTING = {
    "alternate": 0, "adaptive_reinit": 1, "dynamic": 2, "init_once": 3, "opportunistic": 4,
    "default_path": 5, "backpressure": 6, "exponential_backoff": 7,
}
ENUMERATOR = {
    "dfs_random": 0, "dfs_min_card": 1, "dfs_uncertain": 2, "bfs_random": 3, "bfs_min_card": 4,
    "bfs_uncertain": 5, "each_last_once": 6, "each_first_once": 7, "sample": 8,
}


def build(force=False):
    """gcc-compile the C restatement (seconds)."""
    src = os.path.join(_HERE, "polr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", _LIB_PATH, src, "-lm"])
    return _LIB_PATH


class _Col(C.Structure):
    _fields_ = [("data", C.c_void_p), ("valid", C.c_void_p), ("width", C.c_int32), ("is_signed", C.c_int32)]


class _Join(C.Structure):
    _fields_ = [("ht", C.c_void_p), ("pht", C.c_void_p), ("n_keys", C.c_int32),
                ("key_src_join", C.c_int32 * MAX_KEYS), ("key_src_col", C.c_int32 * MAX_KEYS),
                ("estimated_cardinality", C.c_uint64), ("n_preds", C.c_int32), ("pred_op", C.c_int32 * MAX_PREDS),
                ("pred_src_join", C.c_int32 * MAX_PREDS), ("pred_src_col", C.c_int32 * MAX_PREDS),
                ("pred_build_col", C.c_int32 * MAX_PREDS)]


class _Config(C.Structure):
    _fields_ = [("routing", C.c_int32), ("regret_budget", C.c_double), ("init_tuple_count", C.c_uint64),
                ("atc_multiplier", C.c_uint64), ("caching", C.c_int32), ("log_tuples_routed", C.c_int32),
                ("vector_size", C.c_uint64), ("collect_output", C.c_int32), ("trailing_operators", C.c_int32)]


class _Result(C.Structure):
    _fields_ = [("num_intermediates", C.c_uint64), ("num_output_rows", C.c_uint64),
                ("input_tuple_count_per_path", C.c_uint64 * MAX_PATHS),
                ("n_rounds", C.c_uint64), ("intermediates_per_round", C.POINTER(C.c_uint64)),
                ("n_alt_rows", C.c_uint64), ("alt_matrix", C.POINTER(C.c_uint64)),
                ("n_trace", C.c_uint64), ("trace_path", C.POINTER(C.c_uint32)),
                ("trace_tuples", C.POINTER(C.c_uint32)),
                ("out_rows", C.POINTER(C.c_uint32)), ("n_sink_chunks", C.c_uint64)]


_lib = None


class _Filter(C.Structure):
    _fields_ = [("col", C.c_int32), ("op", C.c_int32), ("constant", C.c_int64)]


CMP = {"=": 0, "==": 0, "!=": 1, "<>": 1, "<": 2, ">": 3, "<=": 4, ">=": 5, "is null": 6, "is not null": 7}


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_murmurhash64.restype = C.c_uint64
        L.orc_murmurhash64.argtypes = [C.c_uint64]
        L.orc_hash_value.restype = C.c_uint64
        L.orc_hash_value.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_combine_hash.restype = C.c_uint64
        L.orc_combine_hash.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_ht_build.restype = C.c_void_p
        L.orc_ht_build.argtypes = [C.POINTER(_Col), C.c_int, C.POINTER(_Col), C.c_int, C.c_uint64]
        L.orc_ht_build2.restype = C.c_void_p
        L.orc_ht_build2.argtypes = [C.POINTER(_Col), C.c_int, C.POINTER(_Col), C.c_int, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_ht_free.argtypes = [C.c_void_p]
        for f in ("orc_ht_count", "orc_ht_capacity", "orc_ht_row_width", "orc_ht_pointer_offset"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_ht_col_offset.restype = C.c_uint64
        L.orc_ht_col_offset.argtypes = [C.c_void_p, C.c_int]
        L.orc_ht_has_null.restype = C.c_int
        L.orc_ht_has_null.argtypes = [C.c_void_p]
        L.orc_ht_rows.restype = C.c_void_p
        L.orc_ht_rows.argtypes = [C.c_void_p]
        L.orc_ht_orig_rows.restype = C.c_void_p
        L.orc_ht_orig_rows.argtypes = [C.c_void_p]
        L.orc_ht_bucket_heads.restype = C.c_void_p
        L.orc_ht_bucket_heads.argtypes = [C.c_void_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_pht_build.restype = C.c_void_p
        L.orc_pht_build.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.orc_pht_free.argtypes = [C.c_void_p]
        L.orc_pht_is_dense.restype = C.c_int
        L.orc_pht_is_dense.argtypes = [C.c_void_p]
        L.orc_pht_range.restype = C.c_uint64
        L.orc_pht_range.argtypes = [C.c_void_p]
        L.orc_pht_bitmap.restype = C.c_void_p
        L.orc_pht_bitmap.argtypes = [C.c_void_p]
        L.orc_pht_orig_rows.restype = C.c_void_p
        L.orc_pht_orig_rows.argtypes = [C.c_void_p]
        L.orc_run_pipeline.restype = C.c_int
        L.orc_run_pipeline.argtypes = [C.POINTER(_Col), C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.c_uint64, C.POINTER(_Join), C.c_int, C.c_void_p, C.c_int,
                                       C.POINTER(_Config), C.POINTER(_Result)]
        L.orc_result_free.argtypes = [C.POINTER(_Result)]
        L.orc_materialize_column.restype = C.c_int
        L.orc_materialize_column.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(_Col), C.c_void_p,
                                             C.c_void_p]
        L.orc_mpx_create.restype = C.c_void_p
        L.orc_mpx_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64]
        L.orc_mpx_free.argtypes = [C.c_void_p]
        L.orc_mpx_execute.restype = C.c_int
        L.orc_mpx_execute.argtypes = [C.c_void_p, C.c_uint64] + [C.POINTER(C.c_uint64)] * 4
        L.orc_mpx_add_intermediates.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_mpx_increase_input.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_mpx_finalize_path_run.argtypes = [C.c_void_p]
        L.orc_mpx_resistances.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.orc_join_path_weights.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.orc_enumerate.restype = C.c_int
        L.orc_enumerate.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_bindings.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(_Join), C.c_void_p, C.c_int, C.c_void_p]
        L.orc_scan_filter.restype = C.c_uint64
        L.orc_scan_filter.argtypes = [C.POINTER(_Col), C.c_uint64, C.POINTER(_Filter), C.c_int, C.c_uint64, C.c_void_p,
                                      C.POINTER(C.c_uint64), C.c_void_p]
        _lib = L
    return _lib


def _col(arr, valid=None):
    """numpy array (+ optional uint8 validity) -> orc_col_t; keeps the arrays alive on the struct."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype.kind == "V" or arr.dtype.itemsize == 16:
        width, signed = 16, 0
    else:
        width, signed = arr.dtype.itemsize, int(arr.dtype.kind == "i")
    c = _Col(arr.ctypes.data, None, width, signed)
    c._keep = [arr]
    if valid is not None:
        valid = np.ascontiguousarray(valid, dtype=np.uint8)
        c.valid = valid.ctypes.data
        c._keep.append(valid)
    return c


def _cols(pairs):
    cs = [_col(a, v) for a, v in pairs]
    arr = (_Col * max(len(cs), 1))(*cs)
    arr._keep = cs
    return arr


class HashTable:
    """Reference-layout chained hash table (JoinHashTable) built on the CPU."""

    def __init__(self, keys, payload=(), key_valid=None, payload_valid=None, null_equal=None):
        """null_equal: per key column, IS NOT DISTINCT FROM (NULL = NULL, rows with such a NULL stay in the table)"""
        L = lib()
        self.keys = [np.ascontiguousarray(k) for k in keys]
        self.payload = [np.ascontiguousarray(p) for p in payload]
        n = len(self.keys[0])
        kv = key_valid or [None] * len(self.keys)
        pv = payload_valid or [None] * len(self.payload)
        self.payload_valid = pv
        self._kc = _cols(list(zip(self.keys, kv)))
        self._pc = _cols(list(zip(self.payload, pv)))
        self.n_build_rows = n
        ne = (C.c_int * len(self.keys))(*[1 if x else 0 for x in (null_equal or [0] * len(self.keys))])
        self.h = L.orc_ht_build2(self._kc, len(self.keys), self._pc, len(self.payload), n, ne)
        if not self.h:
            raise ValueError("orc_ht_build failed")
        self.pht = None

    def make_perfect(self, min_value, max_value):
        """BuildPerfectHashTable; returns False (and stays chained) on a duplicate key."""
        p = lib().orc_pht_build(self.h, int(min_value), int(max_value))
        if not p:
            return False
        self.pht = p
        return True

    # accessors -----------------------------------------------------------------------------
    @property
    def count(self):
        return lib().orc_ht_count(self.h)

    @property
    def capacity(self):
        return lib().orc_ht_capacity(self.h)

    @property
    def row_width(self):
        return lib().orc_ht_row_width(self.h)

    @property
    def has_null(self):
        return bool(lib().orc_ht_has_null(self.h))

    def col_offset(self, i):
        return lib().orc_ht_col_offset(self.h, i)

    def rows_blob(self):
        n = self.count * self.row_width
        return np.ctypeslib.as_array(C.cast(lib().orc_ht_rows(self.h), C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n]

    def orig_rows(self):
        n = self.count
        return np.ctypeslib.as_array(C.cast(lib().orc_ht_orig_rows(self.h), C.POINTER(C.c_uint32)),
                                     shape=(max(n, 1),))[:n].copy()

    def bucket_heads(self):
        p = lib().orc_ht_bucket_heads(self.h)
        out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(self.capacity,)).copy()
        lib().orc_free(p)
        return out

    def pht_bitmap(self):
        n = lib().orc_pht_range(self.pht) + 1
        return np.ctypeslib.as_array(C.cast(lib().orc_pht_bitmap(self.pht), C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def pht_orig_rows(self):
        n = lib().orc_pht_range(self.pht) + 1
        return np.ctypeslib.as_array(C.cast(lib().orc_pht_orig_rows(self.pht), C.POINTER(C.c_uint32)),
                                     shape=(n,)).copy()

    def pht_is_dense(self):
        return bool(lib().orc_pht_is_dense(self.pht))

    def close(self):
        if self.pht:
            lib().orc_pht_free(self.pht)
            self.pht = None
        if self.h:
            lib().orc_ht_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class JoinSpec:
    """One multiplexed join: `ht` plus where each probe key comes from: (-1, probe col) or
    (join j, payload col of j)."""

    def __init__(self, ht, key_src, estimated_cardinality=0, preds=()):
        """preds: non-equality conditions [(op, (src_join, src_col), build payload column index)]: left OP right"""
        self.ht = ht
        self.key_src = list(key_src)
        self.estimated_cardinality = estimated_cardinality
        self.preds = list(preds)
        if self.preds and ht.pht:
            raise ValueError("a join with non-equality conditions is never a perfect-hash join "
                             "(physical_hash_join.cpp: conditions.size() == 1)")


def _joins_struct(joins):
    arr = (_Join * len(joins))()
    for i, j in enumerate(joins):
        arr[i].ht = j.ht.h
        arr[i].pht = j.ht.pht
        arr[i].n_keys = len(j.key_src)
        for c, (sj, sc) in enumerate(j.key_src):
            arr[i].key_src_join[c] = sj
            arr[i].key_src_col[c] = sc
        arr[i].estimated_cardinality = j.estimated_cardinality
        arr[i].n_preds = len(j.preds)
        for c, (op, (sj, sc), bc) in enumerate(j.preds):
            arr[i].pred_op[c] = CMP_PRED[op] if isinstance(op, str) else op
            arr[i].pred_src_join[c] = sj
            arr[i].pred_src_col[c] = sc
            arr[i].pred_build_col[c] = bc
    return arr


def run_pipeline(probe_cols, joins, paths, routing="adaptive_reinit", regret_budget=0.01, init_tuple_count=1024,
                 atc_multiplier=1, caching=True, log_tuples_routed=True, vector_size=1024, collect_output=True,
                 sel=None, chunk_offsets=None, probe_valid=None, trailing_operators=1):
    """Runs the restated POLAR pipeline; returns a dict of numpy results."""
    L = lib()
    probe_cols = [np.ascontiguousarray(c) for c in probe_cols]
    n_rows = len(probe_cols[0])
    pv = probe_valid or [None] * len(probe_cols)
    pc = _cols(list(zip(probe_cols, pv)))
    jarr = _joins_struct(joins)
    k = len(joins)
    paths = np.ascontiguousarray(np.asarray(paths, dtype=np.int32).reshape(-1, k))
    cfg = _Config(ROUTING[routing] if isinstance(routing, str) else routing, regret_budget, init_tuple_count,
                  atc_multiplier, int(caching), int(log_tuples_routed), vector_size, int(collect_output),
                  int(trailing_operators))
    res = _Result()
    sel_p, n_sel = None, 0
    if sel is not None:
        sel = np.ascontiguousarray(sel, dtype=np.uint32)
        sel_p, n_sel = sel.ctypes.data, len(sel)
    co_p, n_chunks = None, 0
    if chunk_offsets is not None:
        chunk_offsets = np.ascontiguousarray(chunk_offsets, dtype=np.uint64)
        co_p, n_chunks = chunk_offsets.ctypes.data, len(chunk_offsets) - 1
    rc = L.orc_run_pipeline(pc, len(probe_cols), n_rows, sel_p, n_sel, co_p, n_chunks, jarr, k, paths.ctypes.data,
                            len(paths), C.byref(cfg), C.byref(res))
    if rc != 0:
        raise ValueError("orc_run_pipeline failed (%d)" % rc)
    P = len(paths)

    def arr(ptr, n, dtype):
        if not ptr or n == 0:
            return np.zeros((0,), dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    out = {
        "num_intermediates": int(res.num_intermediates),
        "num_output_rows": int(res.num_output_rows),
        "input_tuple_count_per_path": [int(res.input_tuple_count_per_path[i]) for i in range(P)],
        "intermediates_per_round": arr(res.intermediates_per_round, res.n_rounds, np.uint64),
        "alt_matrix": arr(res.alt_matrix, res.n_alt_rows * P, np.uint64).reshape(-1, P),
        "trace_path": arr(res.trace_path, res.n_trace, np.uint32),
        "trace_tuples": arr(res.trace_tuples, res.n_trace, np.uint32),
        "out_rows": arr(res.out_rows, res.num_output_rows * (1 + k) if collect_output else 0,
                        np.uint32).reshape(-1, 1 + k),
        "n_sink_chunks": int(res.n_sink_chunks),
    }
    L.orc_result_free(C.byref(res))
    return out


def materialize_column(out_rows, k, src_join, col, valid=None):
    """Gather one output column by the oracle's row ids; returns (data, valid)."""
    out_rows = np.ascontiguousarray(out_rows, dtype=np.uint32)
    n = len(out_rows)
    c = _col(col, valid)
    col = np.ascontiguousarray(col)
    data = np.zeros((n,), dtype=col.dtype)
    vld = np.ones((n,), dtype=np.uint8)
    lib().orc_materialize_column(out_rows.ctypes.data, n, k, src_join, C.byref(c), data.ctypes.data,
                                 vld.ctypes.data)
    return data, vld


class Multiplexer:
    """PhysicalMultiplexer + RoutingStrategy in isolation."""

    def __init__(self, n_paths, routing, regret_budget=0.01, init_tuple_count=1024, atc_multiplier=1):
        self.P = n_paths
        self.h = lib().orc_mpx_create(n_paths, ROUTING[routing], regret_budget, init_tuple_count, atc_multiplier)

    def execute(self, input_size):
        off, cnt, path, skips = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        more = lib().orc_mpx_execute(self.h, input_size, C.byref(off), C.byref(cnt), C.byref(path), C.byref(skips))
        return bool(more), off.value, cnt.value, path.value, skips.value

    def add_intermediates(self, n):
        lib().orc_mpx_add_intermediates(self.h, n)

    def increase_input(self, n):
        lib().orc_mpx_increase_input(self.h, n)

    def finalize_path_run(self):
        lib().orc_mpx_finalize_path_run(self.h)

    def resistances(self):
        out = (C.c_double * self.P)()
        lib().orc_mpx_resistances(self.h, out)
        return list(out)

    def __del__(self):
        try:
            lib().orc_mpx_free(self.h)
        except Exception:
            pass


def join_path_weights(costs, regret_budget):
    n = len(costs)
    c = (C.c_double * n)(*costs)
    w = (C.c_double * n)(*([1.0] * n))
    lib().orc_join_path_weights(c, n, regret_budget, w)
    return list(w)


def enumerate_join_orders(enumerator, k, dependencies, estimated_cardinality, max_join_orders):
    deps = np.ascontiguousarray(np.asarray(dependencies, dtype=np.uint8).reshape(k, k))
    card = np.ascontiguousarray(np.asarray(estimated_cardinality, dtype=np.uint64))
    out = np.zeros(((max_join_orders + 2 + k) * 4, k), dtype=np.int32)
    n = lib().orc_enumerate(ENUMERATOR[enumerator], k, deps.ctypes.data, card.ctypes.data, max_join_orders,
                            out.ctypes.data)
    if n < 0:
        raise ValueError("enumerator %s not restated" % enumerator)
    return out[:n].copy()


def bindings(n_probe_cols, num_build_cols, joins, paths):
    k = len(joins)
    paths = np.ascontiguousarray(np.asarray(paths, dtype=np.int32).reshape(-1, k))
    nb = np.ascontiguousarray(np.asarray(num_build_cols, dtype=np.int32))
    out = np.zeros((len(paths), k, MAX_KEYS), dtype=np.int32)
    lib().orc_bindings(k, n_probe_cols, nb.ctypes.data, _joins_struct(joins), paths.ctypes.data, len(paths),
                       out.ctypes.data)
    return out


def scan_filter(cols, filters, vector_size=1024, valids=None):
    """table scan with pushed-down filters (RowGroup::TemplatedScan): cols = numpy arrays, filters =
    [(col, op, constant)] -> (sel uint32[n_sel], chunk_offsets uint64[n_chunks + 1])"""
    L = lib()
    valids = valids or [None] * len(cols)
    cs = _cols(list(zip(cols, valids)))
    n_rows = len(cols[0])
    fa = (_Filter * max(len(filters), 1))()
    for i, (col, op, const) in enumerate(filters):
        fa[i].col, fa[i].op, fa[i].constant = col, CMP[op] if isinstance(op, str) else op, int(const or 0)
    sel = np.zeros((max(n_rows, 1),), dtype=np.uint32)
    offs = np.zeros((n_rows + 2,), dtype=np.uint64)
    n_sel = C.c_uint64()
    n_chunks = L.orc_scan_filter(cs, n_rows, fa, len(filters), vector_size, sel.ctypes.data, C.byref(n_sel),
                                 offs.ctypes.data)
    return sel[:n_sel.value].copy(), offs[:n_chunks + 1].copy()
