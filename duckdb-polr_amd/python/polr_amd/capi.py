"""ctypes binding of include/polr_hip.h (libpolr_hip.so).

This is the only way Python code (tests, bench.py) reaches the device path: straight through the C
ABI.  There is no fallback: if the shared library is missing or no gfx950 device is visible every
entry point raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_PKG))  # duckdb-polr_amd/
LIB_PATH = os.path.join(_ROOT, "libpolr_hip.so")

MAX_JOINS, MAX_PATHS, MAX_KEYS = 8, 32, 4
MAX_PREDS = 4
CMP_PRED = {"=": 0, "==": 0, "<>": 1, "!=": 1, "<": 2, ">": 3, "<=": 4, ">=": 5, "str_eq": 8}  # (8 = POLR_CMP_STR_EQ: string cells)
COL_SIGNED, COL_DEVICE = 1, 2

OK, E_NO_DEVICE, E_INVALID, E_UNSUPPORTED, E_HIP, E_DUPLICATE, E_OVERFLOW = 0, -1, -2, -3, -4, -5, -6

ROUTING = {"alternate": 0, "adaptive_reinit": 1, "dynamic": 2, "init_once": 3, "opportunistic": 4,
           "default_path": 5, "backpressure": 6, "exponential_backoff": 7}

EXPORTS = [
    "polr_abi_version", "polr_ctx_create", "polr_ctx_destroy", "polr_last_error", "polr_ctx_sync",
    "polr_ht_upload_rows", "polr_ht_upload_columns", "polr_ht_finalize_hash", "polr_ht_finalize_perfect",
    "polr_pht_upload", "polr_ht_destroy", "polr_ht_get_info", "polr_ht_export", "polr_ht_alloc_like",
    "polr_pipeline_create", "polr_pipeline_set_selection", "polr_pipeline_update_probe", "polr_pipeline_destroy",
    "polr_out_create", "polr_out_reset", "polr_out_stats", "polr_out_fetch_ids", "polr_out_materialize",
    "polr_out_destroy", "polr_probe_rounds", "polr_probe_rounds_async",
    "polr_mpx_create", "polr_mpx_run", "polr_mpx_set_chunk_offsets", "polr_mpx_finish", "polr_mpx_fetch_log",
    "polr_mpx_destroy", "polr_mpx_reset", "polr_mpx_enable_timing", "polr_mpx_kernel_time", "polr_mpx_run_many", "polr_mpx_finish_many", "polr_mpx_run_resident", "polr_mpx_run_resident_ranges", "polr_mpx_run_resident_morsels",
    "polr_ht_finalize_auto", "polr_pipeline_launch_info", "polr_pipeline_scan_filter", "polr_pipeline_fetch_scan", "polr_mpx_use_scan_chunks", "polr_out_aggregate", "polr_out_aggregate_grouped",
    "polr_mpx_run_backpressure", "polr_pipeline_scan_filter_lip", "polr_comm_get_unique_id", "polr_comm_create", "polr_bcast_build", "polr_comm_bytes_broadcast", "polr_comm_destroy",
    "polr_ctx_set_pool_tuning", "polr_ctx_get_stream",
    "polr_ht_set_payload_heap", "polr_pipeline_set_probe_heap", "polr_out_aggregate_string", "polr_ht_set_key_flags",
    "polr_out_fuse_grouped", "polr_out_fused_result", "polr_out_aggregate_hashed",
]


class PolrError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("polr error %d: %s" % (code, text))
        self.code = code


class Col(C.Structure):
    _fields_ = [("data", C.c_void_p), ("valid", C.c_void_p), ("width", C.c_uint32), ("flags", C.c_uint32)]


KEY_BY_VALUE, KEY_NULL_EQUAL = 1, 2  # polr_ht_set_key_flags


class PoolTuning(C.Structure):
    """polr_pool_tuning: 0 = the library's default"""
    _fields_ = [("device_share", C.c_uint32), ("units_x", C.c_uint32), ("hi_unit", C.c_uint32),
                ("hi_lottery", C.c_uint32), ("hi_tuples_p1", C.c_uint32), ("idle_sleep", C.c_uint32),
                ("watchdog_us", C.c_uint32), ("share_after", C.c_uint32)]


class JoinDesc(C.Structure):
    _fields_ = [("ht", C.c_void_p), ("n_keys", C.c_uint32), ("key_src_join", C.c_int32 * MAX_KEYS),
                ("key_src_col", C.c_int32 * MAX_KEYS), ("n_preds", C.c_uint32), ("pred_op", C.c_uint32 * MAX_PREDS),
                ("pred_src_join", C.c_int32 * MAX_PREDS), ("pred_src_col", C.c_int32 * MAX_PREDS),
                ("pred_build_col", C.c_uint32 * MAX_PREDS)]


class Round(C.Structure):
    _fields_ = [("begin", C.c_uint64), ("count", C.c_uint64), ("path", C.c_uint32), ("emit", C.c_uint32)]


class ScanFilter(C.Structure):
    _fields_ = [("col", C.c_uint32), ("op", C.c_uint32), ("constant", C.c_int64)]


CMP = {"=": 0, "==": 0, "!=": 1, "<>": 1, "<": 2, ">": 3, "<=": 4, ">=": 5, "is null": 6, "is not null": 7}


class AggSpec(C.Structure):
    _fields_ = [("fn", C.c_uint32), ("src_join", C.c_int32), ("src_col", C.c_uint32)]


class AggValue(C.Structure):
    _fields_ = [("lo", C.c_int64), ("hi", C.c_int64), ("count", C.c_uint64), ("is_null", C.c_uint32),
                ("pad", C.c_uint32)]


AGG = {"count_star": 0, "count": 1, "sum": 2, "min": 3, "max": 4}


class LaunchInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("waves_per_workgroup", "workgroups_per_cu", "lds_bytes_per_workgroup",
                                          "compiled_stages", "tuple_slots", "n_cus", "flat", "lds_tables", "lds_table_bytes", "pad")]


class GroupKey(C.Structure):
    _fields_ = [("src_join", C.c_int32), ("src_col", C.c_uint32), ("min_value", C.c_int64), ("n_values", C.c_uint32),
                ("pad", C.c_uint32)]


class HtInfo(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("n_keys", C.c_uint32), ("n_rows", C.c_uint64), ("capacity", C.c_uint64),
                ("max_run", C.c_uint64), ("device_bytes", C.c_uint64), ("is_dense", C.c_uint32),
                ("has_null", C.c_uint32)]


class MpxConfig(C.Structure):
    _fields_ = [("routing", C.c_uint32), ("chunk_size", C.c_uint32), ("regret_budget", C.c_double),
                ("init_tuple_count", C.c_uint64), ("atc_multiplier", C.c_uint64), ("log_rounds", C.c_uint32),
                ("max_log_rounds", C.c_uint32)]


class MpxStats(C.Structure):
    _fields_ = [("num_tuples_processed", C.c_uint64), ("num_intermediates", C.c_uint64), ("num_rounds", C.c_uint64),
                ("input_tuple_count_per_path", C.c_uint64 * MAX_PATHS), ("path_resistances", C.c_double * MAX_PATHS),
                ("stage_out", (C.c_uint64 * MAX_JOINS) * MAX_PATHS)]


_lib = None


def load():
    """dlopen libpolr_hip.so and declare prototypes; raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PolrError(E_NO_DEVICE, "libpolr_hip.so not built (run __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int64
    P = C.POINTER
    L.polr_abi_version.restype = C.c_int
    L.polr_ctx_create.argtypes = [C.c_int, P(vp)]
    L.polr_ctx_destroy.argtypes = [vp]
    L.polr_ctx_destroy.restype = None
    L.polr_last_error.argtypes = [vp]
    L.polr_last_error.restype = C.c_char_p
    L.polr_ctx_sync.argtypes = [vp, vp]
    L.polr_ctx_set_pool_tuning.argtypes = [vp, C.POINTER(PoolTuning)]
    L.polr_ctx_get_stream.argtypes = [vp, C.POINTER(C.c_void_p)]
    L.polr_ht_set_payload_heap.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
    L.polr_ht_set_key_flags.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.polr_pipeline_set_probe_heap.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
    L.polr_out_aggregate_string.argtypes = [vp, vp, C.c_uint32, C.c_int32, C.c_uint32, C.c_char_p, C.c_uint32,
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.polr_ht_upload_rows.argtypes = [vp, vp, u64, u32, vp, vp, vp, u32, u32, P(vp)]
    L.polr_ht_upload_columns.argtypes = [vp, P(Col), u32, P(Col), u32, u64, P(vp)]
    L.polr_ht_finalize_hash.argtypes = [vp, vp]
    L.polr_ht_finalize_perfect.argtypes = [vp, i64, i64, vp]
    L.polr_pht_upload.argtypes = [vp, u32, u32, i64, i64, vp, P(Col), u32, P(vp)]
    L.polr_ht_destroy.argtypes = [vp]
    L.polr_ht_destroy.restype = None
    L.polr_ht_get_info.argtypes = [vp, P(HtInfo)]
    L.polr_ht_export.argtypes = [vp, vp, P(u64), P(vp), P(u64), P(u32)]
    L.polr_ht_alloc_like.argtypes = [vp, vp, u64, P(vp)]
    L.polr_pipeline_create.argtypes = [vp, P(Col), u32, u64, P(JoinDesc), u32, vp, u32, P(vp)]
    L.polr_pipeline_set_selection.argtypes = [vp, vp, u64, u32]
    L.polr_pipeline_update_probe.argtypes = [vp, u32, vp, vp, u64]
    L.polr_pipeline_destroy.argtypes = [vp]
    L.polr_pipeline_destroy.restype = None
    L.polr_out_create.argtypes = [vp, u32, u64, P(vp)]
    L.polr_out_reset.argtypes = [vp, vp]
    L.polr_out_stats.argtypes = [vp, vp, P(u64), P(u64), P(u32)]
    L.polr_out_fetch_ids.argtypes = [vp, vp, vp, u64]
    L.polr_out_materialize.argtypes = [vp, vp, i32, u32, vp, vp, u64, u32]
    L.polr_out_destroy.argtypes = [vp]
    L.polr_out_destroy.restype = None
    L.polr_probe_rounds.argtypes = [vp, vp, P(Round), u32, vp, vp]
    L.polr_probe_rounds_async.argtypes = [vp, vp, P(Round), u32, vp, vp]
    L.polr_mpx_create.argtypes = [vp, P(MpxConfig), P(vp)]
    L.polr_mpx_run.argtypes = [vp, vp, u64, u64, vp]
    L.polr_mpx_set_chunk_offsets.argtypes = [vp, vp, u64]
    L.polr_mpx_finish.argtypes = [vp, vp, P(MpxStats)]
    L.polr_mpx_fetch_log.argtypes = [vp, vp, vp, vp, vp, u64, P(u64)]
    L.polr_mpx_destroy.argtypes = [vp]
    L.polr_mpx_destroy.restype = None
    L.polr_mpx_reset.argtypes = [vp, vp]
    L.polr_mpx_run_many.argtypes = [vp, vp, vp, vp, u32, vp]
    L.polr_mpx_finish_many.argtypes = [vp, u32, vp]
    L.polr_pipeline_launch_info.argtypes = [vp, C.c_int, vp]
    L.polr_ht_finalize_auto.argtypes = [vp, C.c_int64, C.c_int64, vp, vp]
    L.polr_pipeline_scan_filter.argtypes = [vp, vp, vp, u32, u32, vp, vp]
    L.polr_pipeline_scan_filter_lip.argtypes = [vp, vp, vp, u32, u32, u32, vp, vp]
    L.polr_pipeline_fetch_scan.argtypes = [vp, vp, vp]
    L.polr_mpx_use_scan_chunks.argtypes = [vp]
    L.polr_out_aggregate.argtypes = [vp, vp, vp, u32, vp]
    L.polr_out_aggregate_grouped.argtypes = [vp, vp, vp, u32, vp, u32, vp, C.c_uint64, vp]
    L.polr_out_fuse_grouped.argtypes = [vp, vp, u32, vp, u32]
    L.polr_out_aggregate_hashed.argtypes = [vp, vp, vp, u32, vp, u32, C.c_uint64, vp, vp, vp, vp]
    L.polr_out_fused_result.argtypes = [vp, vp, vp, C.c_uint64, vp]
    L.polr_mpx_run_resident.argtypes = [vp, vp, vp, vp, u32, vp, u32]
    L.polr_mpx_run_resident_morsels.argtypes = [vp, vp, C.c_uint64, C.c_uint64, u32, u32, vp, u32]
    L.polr_mpx_run_backpressure.argtypes = [vp, vp, C.c_uint64, C.c_uint64, u32, vp, u32]
    L.polr_mpx_enable_timing.argtypes = [vp, C.c_int]
    L.polr_mpx_kernel_time.argtypes = [vp, P(C.c_double), P(u64)]
    L.polr_comm_get_unique_id.argtypes = [vp]
    L.polr_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, P(vp)]
    L.polr_bcast_build.argtypes = [vp, P(vp), C.c_int, vp]
    L.polr_comm_bytes_broadcast.argtypes = [vp, P(u64)]
    L.polr_comm_destroy.argtypes = [vp]
    L.polr_comm_destroy.restype = None
    _lib = L
    return L


def _np_col(arr, valid=None):
    arr = np.ascontiguousarray(arr)
    width = arr.dtype.itemsize
    flags = COL_SIGNED if arr.dtype.kind == "i" else 0
    c = Col(arr.ctypes.data, None, width, flags)
    c._keep = [arr]
    if valid is not None:
        valid = np.ascontiguousarray(valid, dtype=np.uint8)
        c.valid = valid.ctypes.data
        c._keep.append(valid)
    return c


def dev_col(ptr, width, signed=False, valid_ptr=None):
    """column descriptor for memory that is already on the device (e.g. a torch tensor's data_ptr())"""
    return Col(ptr, valid_ptr, width, COL_DEVICE | (COL_SIGNED if signed else 0))


def string_cells(values):
    """VARCHAR column -> (cells, heap): cells = one 16-byte string_t per value (string_type.hpp:23-28: length, then up to 12
    characters inline, or a 4-byte prefix and a pointer), heap = the bytes the pointers of the long ones point into (a numpy
    array: keep it alive until the heap has been handed over with set_payload_heap / set_probe_heap).  values: bytes / str."""
    vals = [v.encode() if isinstance(v, str) else bytes(v) for v in values]
    heap = np.frombuffer(b"".join(v for v in vals if len(v) > 12) or b"\0", dtype=np.uint8).copy()
    base = heap.ctypes.data
    cells = np.zeros((len(vals), 16), dtype=np.uint8)
    off = 0
    for i, v in enumerate(vals):
        cells[i, 0:4] = np.frombuffer(np.uint32(len(v)).tobytes(), dtype=np.uint8)
        if len(v) <= 12:
            cells[i, 4:4 + len(v)] = np.frombuffer(v, dtype=np.uint8)
        else:
            cells[i, 4:8] = np.frombuffer(v[:4], dtype=np.uint8)
            cells[i, 8:16] = np.frombuffer(np.uint64(base + off).tobytes(), dtype=np.uint8)
            off += len(v)
    return cells.reshape(-1).view("V16"), heap


def string_hashes(values, bits=64):
    """a 64-bit hash per string, as an engine hands over for a VARCHAR join key (JoinHashTable::Hash): the KEY column of
    such a join on both sides; the strings themselves ride along as a "str_eq" condition.  bits < 64 keeps only the low bits
    (tests: collisions on purpose, so that the verifying comparison has something to reject)"""
    import xxhash
    vals = [v.encode() if isinstance(v, str) else bytes(v) for v in values]
    h = np.array([xxhash.xxh64_intdigest(v) for v in vals], dtype=np.uint64)
    return h if bits >= 64 else (h & np.uint64((1 << bits) - 1))


def _col_array(cols):
    arr = (Col * max(len(cols), 1))(*cols)
    arr._keep = cols
    return arr


class Context:
    def __init__(self, device_id=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.polr_ctx_create(device_id, C.byref(h))
        if rc != OK:
            raise PolrError(rc, "polr_ctx_create(%d) failed: no gfx950 device visible" % device_id)
        self.h = h

    def check(self, rc):
        if rc != OK:
            raise PolrError(rc, self.L.polr_last_error(self.h).decode())

    def sync(self, stream=None):
        self.check(self.L.polr_ctx_sync(self.h, stream))

    def stream(self):
        """the context's own stream (polr_ctx_get_stream), to pass to run_resident so that a run is in line with
        Output.reset before it and the aggregates behind it"""
        st = C.c_void_p()
        self.check(self.L.polr_ctx_get_stream(self.h, C.byref(st)))
        return st

    def set_pool_tuning(self, **kw):
        """polr_ctx_set_pool_tuning; no arguments = every default.  hi_tuples=N is passed as hi_tuples_p1 = N + 1"""
        if not kw:
            self.check(self.L.polr_ctx_set_pool_tuning(self.h, None))
            return
        if "hi_tuples" in kw:
            kw["hi_tuples_p1"] = int(kw.pop("hi_tuples")) + 1
        t = PoolTuning(**{k: int(v) for k, v in kw.items()})
        self.check(self.L.polr_ctx_set_pool_tuning(self.h, C.byref(t)))

    def close(self):
        if self.h:
            self.L.polr_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HashTable:
    """A build side resident in HBM."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    @classmethod
    def from_columns(cls, ctx, keys, payload=(), key_valid=None, payload_valid=None):
        """keys/payload: numpy arrays or pre-built Col descriptors (device memory)"""
        kv = key_valid or [None] * len(keys)
        pv = payload_valid or [None] * len(payload)
        kc = [k if isinstance(k, Col) else _np_col(k, v) for k, v in zip(keys, kv)]
        pc = [p if isinstance(p, Col) else _np_col(p, v) for p, v in zip(payload, pv)]
        n = len(keys[0]) if not isinstance(keys[0], Col) else None
        return cls.from_cols(ctx, kc, pc, n)

    @classmethod
    def from_cols(cls, ctx, key_cols, payload_cols, n_rows):
        h = C.c_void_p()
        ctx.check(ctx.L.polr_ht_upload_columns(ctx.h, _col_array(key_cols), len(key_cols), _col_array(payload_cols),
                                               len(payload_cols), n_rows, C.byref(h)))
        return cls(ctx, h)

    def set_key_flags(self, key_col, flags):
        """polr_ht_set_key_flags (before finalize): KEY_BY_VALUE = the probe side may read another integer type (a CAST'ed
        key), KEY_NULL_EQUAL = IS NOT DISTINCT FROM"""
        self.ctx.check(self.ctx.L.polr_ht_set_key_flags(self.h, key_col, flags))

    def set_payload_heap(self, payload_col, heap):
        """polr_ht_set_payload_heap: the string heap the cells of payload column `payload_col` point into (before finalize)"""
        heap = np.ascontiguousarray(heap, dtype=np.uint8)
        self.ctx.check(self.ctx.L.polr_ht_set_payload_heap(self.h, payload_col, heap.ctypes.data, heap.nbytes))
        return self

    @classmethod
    def from_rows(cls, ctx, rows, n_rows, row_width, col_offset, col_width, col_signed, n_keys, n_payload):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        off = np.ascontiguousarray(col_offset, dtype=np.uint32)
        wid = np.ascontiguousarray(col_width, dtype=np.uint32)
        flg = np.ascontiguousarray([COL_SIGNED if s else 0 for s in col_signed], dtype=np.uint32)
        h = C.c_void_p()
        ctx.check(ctx.L.polr_ht_upload_rows(ctx.h, rows.ctypes.data, n_rows, row_width, off.ctypes.data,
                                            wid.ctypes.data, flg.ctypes.data, n_keys, n_payload, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_perfect(cls, ctx, key_width, key_signed, min_value, max_value, bitmap, payload=(), payload_valid=None):
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
        pv = payload_valid or [None] * len(payload)
        pc = [_np_col(p, v) for p, v in zip(payload, pv)]
        h = C.c_void_p()
        ctx.check(ctx.L.polr_pht_upload(ctx.h, key_width, COL_SIGNED if key_signed else 0, int(min_value),
                                        int(max_value), bitmap.ctypes.data, _col_array(pc), len(pc), C.byref(h)))
        return cls(ctx, h)

    def finalize_hash(self, stream=None):
        self.ctx.check(self.ctx.L.polr_ht_finalize_hash(self.h, stream))
        return self

    def finalize_perfect(self, min_value, max_value, stream=None):
        """returns False on a duplicate key (caller falls back to finalize_hash)"""
        rc = self.ctx.L.polr_ht_finalize_perfect(self.h, int(min_value), int(max_value), stream)
        if rc == E_DUPLICATE:
            return False
        self.ctx.check(rc)
        return True

    def finalize_auto(self, min_value, max_value, stream=None):
        """polr_ht_finalize_auto: perfect table for a dense unique integer key, hash table otherwise -> kind"""
        kind = C.c_uint32()
        self.ctx.check(self.ctx.L.polr_ht_finalize_auto(self.h, int(min_value), int(max_value), stream, C.byref(kind)))
        return kind.value

    def info(self):
        i = HtInfo()
        self.ctx.check(self.ctx.L.polr_ht_get_info(self.h, C.byref(i)))
        return {f[0]: getattr(i, f[0]) for f in HtInfo._fields_}

    def export(self):
        """(meta bytes, [(device ptr, nbytes)]) of the finalized table, for a broadcast"""
        L = self.ctx.L
        mb, nb = C.c_uint64(0), C.c_uint32(0)
        self.ctx.check(L.polr_ht_export(self.h, None, C.byref(mb), None, None, C.byref(nb)))
        meta = (C.c_uint8 * mb.value)()
        ptrs = (C.c_void_p * nb.value)()
        sizes = (C.c_uint64 * nb.value)()
        self.ctx.check(L.polr_ht_export(self.h, meta, C.byref(mb), ptrs, sizes, C.byref(nb)))
        return bytes(meta), [(ptrs[i], sizes[i]) for i in range(nb.value)]

    @classmethod
    def alloc_like(cls, ctx, meta):
        buf = (C.c_uint8 * len(meta)).from_buffer_copy(meta)
        h = C.c_void_p()
        ctx.check(ctx.L.polr_ht_alloc_like(ctx.h, buf, len(meta), C.byref(h)))
        return cls(ctx, h)

    def close(self):
        if self.h:
            self.ctx.L.polr_ht_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    def __init__(self, ctx, probe_cols, n_probe_rows, joins, paths, probe_valid=None):
        """probe_cols: numpy arrays or Col descriptors; joins: [(HashTable, [(src_join, src_col)])]"""
        self.ctx = ctx
        pv = probe_valid or [None] * len(probe_cols)
        pc = [c if isinstance(c, Col) else _np_col(c, v) for c, v in zip(probe_cols, pv)]
        self.k = len(joins)
        jd = (JoinDesc * self.k)()
        self._hts = [j[0] for j in joins]
        for i, (ht, key_src) in enumerate(joins):
            jd[i].ht = ht.h
            jd[i].n_keys = len(key_src)
            for c, (sj, sc) in enumerate(key_src):
                jd[i].key_src_join[c] = sj
                jd[i].key_src_col[c] = sc
            # non-equality conditions of the join ride on the table object: ht.preds = [(op, (src_join, src_col),
            # payload column index)] (build_joins sets it from the workload's "preds")
            preds = getattr(ht, "preds", None) or []
            jd[i].n_preds = len(preds)
            for c, (op, (sj, sc), bc) in enumerate(preds[:MAX_PREDS]):
                jd[i].pred_op[c] = CMP_PRED[op] if isinstance(op, str) else op
                jd[i].pred_src_join[c] = sj
                jd[i].pred_src_col[c] = sc
                jd[i].pred_build_col[c] = bc
        paths = np.ascontiguousarray(np.asarray(paths, dtype=np.int32).reshape(-1, self.k))
        self.n_paths = len(paths)
        h = C.c_void_p()
        ctx.check(ctx.L.polr_pipeline_create(ctx.h, _col_array(pc), len(pc), n_probe_rows, jd, self.k,
                                             paths.ctypes.data, self.n_paths, C.byref(h)))
        self.h = h

    def set_probe_heap(self, probe_col, heap):
        """polr_pipeline_set_probe_heap: the string heap the cells of probe column `probe_col` point into"""
        heap = np.ascontiguousarray(heap, dtype=np.uint8)
        self.ctx.check(self.ctx.L.polr_pipeline_set_probe_heap(self.h, probe_col, heap.ctypes.data, heap.nbytes))

    def set_selection(self, sel, device=False, n=None):
        if sel is None:
            self.ctx.check(self.ctx.L.polr_pipeline_set_selection(self.h, None, 0, 0))
        elif device:
            self.ctx.check(self.ctx.L.polr_pipeline_set_selection(self.h, sel, n, COL_DEVICE))
        else:
            sel = np.ascontiguousarray(sel, dtype=np.uint32)
            self.ctx.check(self.ctx.L.polr_pipeline_set_selection(self.h, sel.ctypes.data, len(sel), 0))

    def launch_info(self, materialize=False):
        i = LaunchInfo()
        self.ctx.check(self.ctx.L.polr_pipeline_launch_info(self.h, int(materialize), C.byref(i)))
        return {f[0]: getattr(i, f[0]) for f in LaunchInfo._fields_}

    def scan_filter(self, filters, vector_size=1024, stream=None, lip_joins=0):
        """polr_pipeline_scan_filter(_lip): filters = [(col, op, constant)] with op in CMP; lip_joins: bit j = also
        apply join j's filter at the source (LIP); the selection and the chunk boundaries stay on the device ->
        (n_selected, n_chunks)"""
        n = len(filters)
        arr = (ScanFilter * max(n, 1))()
        for i, (col, op, const) in enumerate(filters):
            arr[i].col, arr[i].op, arr[i].constant = col, CMP[op] if isinstance(op, str) else op, int(const or 0)
        ns, nc = C.c_uint64(), C.c_uint64()
        self.ctx.check(self.ctx.L.polr_pipeline_scan_filter_lip(self.h, stream, arr if n else None, n, int(lip_joins),
                                                                vector_size, C.byref(ns), C.byref(nc)))
        self.scan = (ns.value, nc.value)
        return self.scan

    def fetch_scan(self):
        ns, nc = self.scan
        sel = np.zeros((ns,), dtype=np.uint32)
        offs = np.zeros((nc + 1,), dtype=np.uint64)
        self.ctx.check(self.ctx.L.polr_pipeline_fetch_scan(self.h, sel.ctypes.data, offs.ctypes.data))
        return sel, offs

    def probe_rounds(self, rounds, out=None, stream=None):
        """rounds: [(begin, count, path, emit)] -> counts ndarray [n_rounds, k]"""
        n = len(rounds)
        arr = (Round * n)()
        for i, (b, c, p, e) in enumerate(rounds):
            arr[i].begin, arr[i].count, arr[i].path, arr[i].emit = int(b), int(c), int(p), int(e)
        counts = np.zeros((n, self.k), dtype=np.uint64)
        rc = self.ctx.L.polr_probe_rounds(self.h, stream, arr, n, out.h if out else None, counts.ctypes.data)
        if rc == E_OVERFLOW:
            raise PolrError(rc, self.ctx.L.polr_last_error(self.ctx.h).decode())
        self.ctx.check(rc)
        return counts

    def probe_rounds_async(self, rounds_struct, n, counts_dev_ptr, out=None, stream=None):
        self.ctx.check(self.ctx.L.polr_probe_rounds_async(self.h, stream, rounds_struct, n, out.h if out else None,
                                                          counts_dev_ptr))

    def close(self):
        if self.h:
            self.ctx.L.polr_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_rounds(rounds):
    arr = (Round * len(rounds))()
    for i, (b, c, p, e) in enumerate(rounds):
        arr[i].begin, arr[i].count, arr[i].path, arr[i].emit = int(b), int(c), int(p), int(e)
    return arr


class Output:
    def __init__(self, pipe, chunk_capacity=1024, max_chunks=1024):
        self.pipe = pipe
        self.ctx = pipe.ctx
        h = C.c_void_p()
        self.ctx.check(self.ctx.L.polr_out_create(pipe.h, chunk_capacity, max_chunks, C.byref(h)))
        self.h = h

    def reset(self, stream=None):
        self.ctx.check(self.ctx.L.polr_out_reset(self.h, stream))

    def stats(self, stream=None):
        n, c, o = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self.ctx.check(self.ctx.L.polr_out_stats(self.h, stream, C.byref(n), C.byref(c), C.byref(o)))
        return n.value, c.value, bool(o.value)

    def fetch_ids(self, stream=None):
        n, _, _ = self.stats(stream)
        out = np.zeros((n, 1 + self.pipe.k), dtype=np.uint32)
        self.ctx.check(self.ctx.L.polr_out_fetch_ids(self.h, stream, out.ctypes.data, n))
        return out

    def aggregate(self, specs, stream=None):
        """polr_out_aggregate: specs = [(fn, src_join, src_col)] with fn in AGG -> [python int or None]
        (ungrouped COUNT(*) / COUNT / SUM / MIN / MAX over the output, reduced on the device)"""
        n = len(specs)
        arr = (AggSpec * n)()
        for i, (fn, sj, sc) in enumerate(specs):
            arr[i].fn, arr[i].src_join, arr[i].src_col = AGG[fn] if isinstance(fn, str) else fn, sj, sc
        res = (AggValue * n)()
        self.ctx.check(self.ctx.L.polr_out_aggregate(self.h, stream, arr, n, res))
        out = []
        for r in res:
            out.append(None if r.is_null else (r.hi << 64) + (r.lo & 0xFFFFFFFFFFFFFFFF))
        return out

    def aggregate_grouped(self, keys, specs, stream=None):
        """polr_out_aggregate_grouped: keys = [(src_join, src_col, min, n_values)], specs as in aggregate() ->
        (values[n_groups][n_aggs] of python int / None, counts[n_groups][n_aggs], n_dropped)"""
        nk, na = len(keys), len(specs)
        ka = (GroupKey * nk)()
        n_groups = 1
        for i, (sj, sc, mn, nv) in enumerate(keys):
            ka[i].src_join, ka[i].src_col, ka[i].min_value, ka[i].n_values = sj, sc, int(mn), int(nv)
            n_groups *= int(nv)
        sa = (AggSpec * na)()
        for i, (fn, sj, sc) in enumerate(specs):
            sa[i].fn, sa[i].src_join, sa[i].src_col = AGG[fn] if isinstance(fn, str) else fn, sj, sc
        res = (AggValue * (n_groups * na))()
        dropped = C.c_uint64()
        self.ctx.check(self.ctx.L.polr_out_aggregate_grouped(self.h, stream, ka, nk, sa, na, res, n_groups,
                                                             C.byref(dropped)))
        vals = [[None if res[g * na + a].is_null else (res[g * na + a].hi << 64) + (res[g * na + a].lo & 0xFFFFFFFFFFFFFFFF)
                 for a in range(na)] for g in range(n_groups)]
        counts = [[res[g * na + a].count for a in range(na)] for g in range(n_groups)]
        return vals, counts, dropped.value

    def aggregate_hashed(self, cols, specs, max_groups, stream=None):
        """polr_out_aggregate_hashed: GROUP BY over group columns of any integer domain.  cols = [(src_join, src_col)],
        specs as in aggregate() -> {group key tuple (None = NULL): [value per aggregate (python int / None)]}"""
        nk, na = len(cols), len(specs)
        ka = (GroupKey * nk)()
        for i, (sj, sc) in enumerate(cols):
            ka[i].src_join, ka[i].src_col = sj, sc
        sa = (AggSpec * na)()
        for i, (fn, sj, sc) in enumerate(specs):
            sa[i].fn, sa[i].src_join, sa[i].src_col = AGG[fn] if isinstance(fn, str) else fn, sj, sc
        keys = np.zeros((max_groups, nk), dtype=np.int64)
        nulls = np.zeros((max_groups,), dtype=np.uint32)
        res = (AggValue * (max_groups * na))()
        n = C.c_uint64()
        self.ctx.check(self.ctx.L.polr_out_aggregate_hashed(self.h, stream, ka, nk, sa, na, max_groups, keys.ctypes.data,
                                                            nulls.ctypes.data, res, C.byref(n)))
        out = {}
        for g in range(n.value):
            key = tuple(None if (nulls[g] >> c) & 1 else int(keys[g, c]) for c in range(nk))
            out[key] = [None if res[g * na + a].is_null else (res[g * na + a].hi << 64) + (res[g * na + a].lo & 0xFFFFFFFFFFFFFFFF)
                        for a in range(na)]
        return out

    def fuse_grouped(self, keys, specs):
        """polr_out_fuse_grouped: fold the join result into group cells inside the run (flat pipelines of perfect tables;
        COUNT(*) / COUNT / SUM).  keys / specs as in aggregate_grouped; keys=None un-fuses.  reset() zeroes the cells."""
        if keys is None:
            self.ctx.check(self.ctx.L.polr_out_fuse_grouped(self.h, None, 0, None, 0))
            self._fused = None
            return
        nk, na = len(keys), len(specs)
        ka = (GroupKey * nk)()
        n_groups = 1
        for i, (sj, sc, mn, nv) in enumerate(keys):
            ka[i].src_join, ka[i].src_col, ka[i].min_value, ka[i].n_values = sj, sc, int(mn), int(nv)
            n_groups *= int(nv)
        sa = (AggSpec * na)()
        for i, (fn, sj, sc) in enumerate(specs):
            sa[i].fn, sa[i].src_join, sa[i].src_col = AGG[fn] if isinstance(fn, str) else fn, sj, sc
        self.ctx.check(self.ctx.L.polr_out_fuse_grouped(self.h, ka, nk, sa, na))
        self._fused = (n_groups, na)

    def fused_result(self, stream=None):
        """polr_out_fused_result -> (values[n_groups][n_aggs], counts[n_groups][n_aggs], n_dropped), as aggregate_grouped"""
        n_groups, na = self._fused
        res = (AggValue * (n_groups * na))()
        dropped = C.c_uint64()
        self.ctx.check(self.ctx.L.polr_out_fused_result(self.h, stream, res, n_groups, C.byref(dropped)))
        vals = [[None if res[g * na + a].is_null else (res[g * na + a].hi << 64) + (res[g * na + a].lo & 0xFFFFFFFFFFFFFFFF)
                 for a in range(na)] for g in range(n_groups)]
        counts = [[res[g * na + a].count for a in range(na)] for g in range(n_groups)]
        return vals, counts, dropped.value

    def aggregate_string(self, fn, src_join, src_col, stream=None, cap=4096):
        """polr_out_aggregate_string: MIN / MAX of a VARCHAR column over the output rows -> bytes, or None (no non-NULL row)"""
        buf = C.create_string_buffer(cap)
        n, null = C.c_uint32(), C.c_uint32()
        self.ctx.check(self.ctx.L.polr_out_aggregate_string(self.h, stream, AGG[fn] if isinstance(fn, str) else fn, src_join,
                                                            src_col, buf, cap, C.byref(n), C.byref(null)))
        if null.value:
            return None
        if n.value > cap:
            return self.aggregate_string(fn, src_join, src_col, stream, cap=n.value)
        return buf.raw[:n.value]

    def materialize(self, src_join, src_col, dtype, stream=None):
        n, _, _ = self.stats(stream)
        data = np.zeros((n,), dtype=dtype)
        valid = np.ones((n,), dtype=np.uint8)
        self.ctx.check(self.ctx.L.polr_out_materialize(self.h, stream, src_join, src_col, data.ctypes.data,
                                                       valid.ctypes.data, n, 0))
        return data, valid

    def close(self):
        if self.h:
            self.ctx.L.polr_out_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceMultiplexer:
    def __init__(self, pipe, routing, chunk_size=1024, regret_budget=0.01, init_tuple_count=1024, atc_multiplier=1,
                 log_rounds=True, max_log_rounds=1 << 20):
        self.pipe = pipe
        self.ctx = pipe.ctx
        cfg = MpxConfig(ROUTING[routing] if isinstance(routing, str) else routing, chunk_size, regret_budget,
                        init_tuple_count, atc_multiplier, int(log_rounds), max_log_rounds)
        h = C.c_void_p()
        self.ctx.check(self.ctx.L.polr_mpx_create(pipe.h, C.byref(cfg), C.byref(h)))
        self.h = h
        self.max_log_rounds = max_log_rounds

    def set_chunk_offsets(self, offsets):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.ctx.check(self.ctx.L.polr_mpx_set_chunk_offsets(self.h, offsets.ctypes.data, len(offsets) - 1))

    def use_scan_chunks(self):
        """source chunks = the pipeline's scan_filter result (device-resident boundaries)"""
        self.ctx.check(self.ctx.L.polr_mpx_use_scan_chunks(self.h))

    def run(self, chunk_begin, chunk_end, out=None, stream=None):
        self.ctx.check(self.ctx.L.polr_mpx_run(self.h, stream, chunk_begin, chunk_end, out.h if out else None))

    def run_resident(self, chunk_begin, chunk_end, out=None):
        """the same run as one launch (polr_mpx_run_resident with this single executor)"""
        run_resident([self], [(chunk_begin, chunk_end)], out)

    def finish(self, stream=None):
        st = MpxStats()
        self.ctx.check(self.ctx.L.polr_mpx_finish(self.h, stream, C.byref(st)))
        P = self.pipe.n_paths
        return {"num_tuples_processed": st.num_tuples_processed, "num_intermediates": st.num_intermediates,
                "num_rounds": st.num_rounds,
                "input_tuple_count_per_path": [st.input_tuple_count_per_path[i] for i in range(P)],
                "path_resistances": [st.path_resistances[i] for i in range(P)],
                "stage_out": [[st.stage_out[i][j] for j in range(self.pipe.k)] for i in range(P)]}

    def reset(self, stream=None):
        self.ctx.check(self.ctx.L.polr_mpx_reset(self.h, stream))

    def enable_timing(self, on=True):
        self.ctx.check(self.ctx.L.polr_mpx_enable_timing(self.h, int(on)))

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint64()
        self.ctx.check(self.ctx.L.polr_mpx_kernel_time(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def fetch_log(self, stream=None):
        n = C.c_uint64()
        cap = self.max_log_rounds
        path = np.zeros((cap,), dtype=np.uint32)
        tuples = np.zeros((cap,), dtype=np.uint64)
        inter = np.zeros((cap,), dtype=np.uint64)
        self.ctx.check(self.ctx.L.polr_mpx_fetch_log(self.h, stream, path.ctypes.data, tuples.ctypes.data,
                                                     inter.ctypes.data, cap, C.byref(n)))
        m = n.value
        return path[:m].copy(), tuples[:m].copy(), inter[:m].copy()

    def close(self):
        if self.h:
            self.ctx.L.polr_mpx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_many(mpxs, ranges, out=None):
    """polr_mpx_run_many: mpxs[i] routes chunks ranges[i] = (begin, end) concurrently (own streams)"""
    n = len(mpxs)
    hs = (C.c_void_p * n)(*[m.h for m in mpxs])
    b = np.ascontiguousarray([r[0] for r in ranges], dtype=np.uint64)
    e = np.ascontiguousarray([r[1] for r in ranges], dtype=np.uint64)
    ctx = mpxs[0].ctx
    ctx.check(ctx.L.polr_mpx_run_many(hs, None, b.ctypes.data, e.ctypes.data, n, out.h if out else None))


def _stats_dict(st, P, k):
    return {"num_tuples_processed": st.num_tuples_processed, "num_intermediates": st.num_intermediates,
            "num_rounds": st.num_rounds,
            "input_tuple_count_per_path": [st.input_tuple_count_per_path[i] for i in range(P)],
            "path_resistances": [st.path_resistances[i] for i in range(P)],
            "stage_out": [[st.stage_out[i][j] for j in range(k)] for i in range(P)]}


RUN_RESET, RUN_FINISH = 1, 2


def run_resident(mpxs, ranges, out=None, reset=False, finish=False, share=1, stream=None):
    """polr_mpx_run_resident: the same run as ONE launch (device-resident routing loop);
    reset / finish fold polr_mpx_reset / the closing FinalizePathRun into the same launch"""
    ctx = mpxs[0].ctx
    n = len(mpxs)
    hs = (C.c_void_p * n)(*[m.h for m in mpxs])
    b = np.ascontiguousarray([r[0] for r in ranges], dtype=np.uint64)
    e = np.ascontiguousarray([r[1] for r in ranges], dtype=np.uint64)
    ctx.check(ctx.L.polr_mpx_run_resident(hs, stream, b.ctypes.data, e.ctypes.data, n, out.h if out else None,
                                          (RUN_RESET if reset else 0) | (RUN_FINISH if finish else 0) | ((share & 0xFF) << 8 if share > 1 else 0)))


def run_resident_ranges(mpxs, range_lists, out=None, reset=False, finish=False, share=1):
    """polr_mpx_run_resident_ranges: range_lists[i] = [(begin, end), ...] of executor i (the same number for every one)"""
    ctx = mpxs[0].ctx
    n = len(mpxs)
    r = len(range_lists[0])
    hs = (C.c_void_p * n)(*[m.h for m in mpxs])
    flat = [x for lst in range_lists for x in lst]
    b = (C.c_uint64 * (n * r))(*[x[0] for x in flat])
    e = (C.c_uint64 * (n * r))(*[x[1] for x in flat])
    ctx.L.polr_mpx_run_resident_ranges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                  C.c_void_p, C.c_uint32]
    ctx.check(ctx.L.polr_mpx_run_resident_ranges(hs, None, b, e, r, n, out.h if out else None,
                                                 (RUN_RESET if reset else 0) | (RUN_FINISH if finish else 0) |
                                                 ((share & 0xFF) << 8 if share > 1 else 0)))


def run_resident_morsels(mpxs, chunk_begin, chunk_end, morsel_chunks=120, out=None, reset=False, finish=False, share=1):
    """polr_mpx_run_resident_morsels: the executors share [chunk_begin, chunk_end) and pull morsels from one cursor"""
    ctx = mpxs[0].ctx
    n = len(mpxs)
    hs = (C.c_void_p * n)(*[m.h for m in mpxs])
    ctx.check(ctx.L.polr_mpx_run_resident_morsels(hs, None, chunk_begin, chunk_end, morsel_chunks, n,
                                                  out.h if out else None,
                                                  (RUN_RESET if reset else 0) | (RUN_FINISH if finish else 0) |
                                                  ((share & 0xFF) << 8 if share > 1 else 0)))


def run_backpressure(mpxs, chunk_begin, chunk_end, morsel_chunks=120, out=None, reset=True, finish=True):
    """polr_mpx_run_backpressure: one executor per join order racing for morsels of the source"""
    ctx = mpxs[0].ctx
    n = len(mpxs)
    hs = (C.c_void_p * n)(*[m.h for m in mpxs])
    ctx.check(ctx.L.polr_mpx_run_backpressure(hs, None, chunk_begin, chunk_end, morsel_chunks, out.h if out else None,
                                              (RUN_RESET if reset else 0) | (RUN_FINISH if finish else 0)))


def finish_many_raw(mpxs):
    """polr_mpx_finish_many: settles the run(s) of these multiplexers and returns their statistics as the C structs
    (stats_dicts() turns them into dictionaries -- host-side bookkeeping a measurement keeps outside its clock)"""
    n = len(mpxs)
    hs = (C.c_void_p * n)(*[m.h for m in mpxs])
    stats = (MpxStats * n)()
    ctx = mpxs[0].ctx
    ctx.check(ctx.L.polr_mpx_finish_many(hs, n, stats))
    return stats


def stats_dicts(stats, mpxs):
    return [_stats_dict(stats[i], mpxs[i].pipe.n_paths, mpxs[i].pipe.k) for i in range(len(mpxs))]


def finish_many(mpxs):
    return stats_dicts(finish_many_raw(mpxs), mpxs)


COMM_ID_BYTES = 128


def comm_unique_id():
    """polr_comm_get_unique_id: 128 bytes made on ONE rank, to be shipped to the others out of band"""
    L = load()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = L.polr_comm_get_unique_id(buf)
    if rc != OK:
        raise PolrError(rc, "polr_comm_get_unique_id failed (is librccl.so present?)")
    return bytes(buf)


class Comm:
    """RCCL communicator of the library (one rank per GPU) for polr_bcast_build, the path's one exchange step"""

    def __init__(self, ctx, unique_id, world_size, rank):
        self.ctx = ctx
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        ctx.check(ctx.L.polr_comm_create(ctx.h, buf, world_size, rank, C.byref(h)))
        self.h = h
        self.rank = rank

    def bcast_build(self, ht, root=0, stream=None):
        """root: ht is the finalized HashTable to send (returned unchanged); other ranks: pass None, get a new one"""
        h = C.c_void_p(ht.h.value if ht is not None else None)
        self.ctx.check(self.ctx.L.polr_bcast_build(self.h, C.byref(h), root, stream))
        return ht if self.rank == root else HashTable(self.ctx, h)

    def bytes_broadcast(self):
        n = C.c_uint64()
        self.ctx.check(self.ctx.L.polr_comm_bytes_broadcast(self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            self.ctx.L.polr_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def string_payload_index(j, col):
    """payload column number of VARCHAR column `col` of workload join `j` as build_joins uploads it"""
    return len(j["payload"]) + list(j.get("strings", {}).keys()).index(col)


def build_joins(ctx, wl, auto=False):
    """upload + finalize the build sides of a workload dict.  auto=False: the way the reference's planner would
    (perfect table where the plan allows it and the build has no duplicate, hash table otherwise);
    auto=True: polr_ht_finalize_auto with the key column's min/max statistics (dense unique keys of any range
    become perfect tables)."""
    joins = []
    for j in wl["joins"]:
        pv = [j.get("payload_valid", {}).get(n) for n in j["payload"].keys()]
        # VARCHAR payload columns (j["strings"]: name -> bytes per build row) go behind the fixed-width ones, as 16-byte
        # string_t cells + one heap each (string_payload_index gives their column number)
        strs = [string_cells(v) for v in j.get("strings", {}).values()]
        ht = HashTable.from_columns(ctx, j["keys"], list(j["payload"].values()) + [c for c, _h in strs],
                                    key_valid=j.get("key_valid"), payload_valid=pv + [None] * len(strs))
        for i, (_c, heap) in enumerate(strs):
            ht.set_payload_heap(len(j["payload"]) + i, heap)
        # j["key_flags"]: per key column KEY_BY_VALUE (the probe side reads another integer type: a CAST'ed key) and / or
        # KEY_NULL_EQUAL (IS NOT DISTINCT FROM)
        for c, f in enumerate(j.get("key_flags", [])):
            if f:
                ht.set_key_flags(c, f)
        done = False
        if auto and not any(j.get("key_flags", [])) and len(j["keys"]) == 1 and j["keys"][0].dtype.kind in "iu" and len(j["keys"][0]):
            kv = j.get("key_valid")
            kk = j["keys"][0] if not kv or kv[0] is None else j["keys"][0][kv[0].astype(bool)]
            if len(kk):
                ht.finalize_auto(int(kk.min()), int(kk.max()))
                done = True
        if not done and j.get("perfect") is not None:
            done = ht.finalize_perfect(*j["perfect"])
        if not done:
            ht.finalize_hash()
        # conditions: (op, (src_join, src_col), build column NAME) -- a fixed-width payload column, or for "str_eq" one of the
        # join's VARCHAR columns (their column numbers: string_payload_index)
        names = list(j["payload"].keys())
        ht.preds = [(op, src, names.index(col) if col in names else string_payload_index(j, col))
                    for op, src, col in j.get("preds", [])]
        joins.append((ht, j["key_src"]))
    return joins
