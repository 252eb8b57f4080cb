/*
 * include/polr_hip.h -- the drop-in boundary of the MI355X POLAR path.
 *
 * A thin C ABI (plain pointers and sizes, no C++/torch types) under the reference's operator
 * classes.  The reference (d-justen/duckdb-polr) has no FFI seam: PhysicalMultiplexer,
 * PhysicalHashJoin and PhysicalAdaptiveUnion are C++ classes linked into libduckdb, and the
 * boundary sits inside POLARPipelineExecutor::RunPath (src/parallel/polar_pipeline_executor.cpp:
 * 427-538).  Each entry point below names the reference code whose job it takes over; the
 * C++ host mirror in duckdb-polr_amd/host/ (same class names as the reference) is the only
 * intended caller, INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: every call returns 0 on success or a negative POLR_E_* code; polr_last_error()
 * returns the text of the last failure of that context (thread-compatible: one context per host
 * thread, like one PipelineExecutor per thread in the reference).  The caller owns host buffers;
 * the library owns device buffers behind opaque handles.  `stream` is a hipStream_t passed as
 * void* (NULL = the context's own stream).  There is NO CPU fallback: without a visible gfx950
 * device polr_ctx_create fails with POLR_E_NO_DEVICE.
 */
#ifndef POLR_HIP_H
#define POLR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POLR_ABI_VERSION 1

#define POLR_MAX_JOINS 8   /* consecutive INNER hash joins multiplexed in one pipeline */
#define POLR_MAX_PATHS 32  /* >= max_join_orders used by the reference's experiments (24) */
#define POLR_MAX_KEYS 4    /* equality conditions per join.  Two columns of <= 32 bits, or one of <= 64, are compared as
                            * they are; any other combination (3 or 4 keys, a wide column in a composite) is packed
                            * EXACTLY into 64 bits from the build side's per-column [min, max] (sum of the columns' range
                            * bits <= 64, else POLR_E_UNSUPPORTED): never a hash, so no verification pass */

enum {
	POLR_OK = 0,
	POLR_E_NO_DEVICE = -1,
	POLR_E_INVALID = -2,     /* bad argument / shape mismatch (checked on the host before any launch) */
	POLR_E_UNSUPPORTED = -3, /* type or plan shape outside the path (fails loudly, never falls back) */
	POLR_E_HIP = -4,         /* a HIP runtime call failed; text in polr_last_error */
	POLR_E_DUPLICATE = -5,   /* perfect-hash build saw a duplicate key: caller keeps the chained table,
	                            like perfect_hash_join_executor.cpp:112-114 */
	POLR_E_OVERFLOW = -6     /* output buffer too small; counters are still exact */
};

/* MultiplexerRouting (src/include/duckdb/main/config.hpp:41-50), same order */
enum {
	POLR_ROUTE_ALTERNATE = 0,
	POLR_ROUTE_ADAPTIVE_REINIT = 1,
	POLR_ROUTE_DYNAMIC = 2,
	POLR_ROUTE_INIT_ONCE = 3,
	POLR_ROUTE_OPPORTUNISTIC = 4,
	POLR_ROUTE_DEFAULT_PATH = 5,
	POLR_ROUTE_BACKPRESSURE = 6,
	POLR_ROUTE_EXPONENTIAL_BACKOFF = 7
};

typedef struct polr_ctx polr_ctx;           /* one device + one stream + error text */
typedef struct polr_ht polr_ht;             /* a finalized build side resident in HBM */
typedef struct polr_pipeline polr_pipeline; /* probe columns + joins + join orders (POLARConfig) */
typedef struct polr_out polr_out;           /* chunked output of a run (DataChunk stream) */
typedef struct polr_mpx polr_mpx;           /* device-resident multiplexer state + router */

/* A flat column (FLAT vector after ToUnifiedFormat, src/include/duckdb/common/types/vector.hpp:22-27).
 * valid: NULL = all valid, else one byte per row (1 = valid).  width in bytes: 1, 2, 4, 8 or 16
 * (16 = string_t copied as an opaque cell, row_gather.cpp:47-86). */
#define POLR_COL_SIGNED 1u
#define POLR_COL_DEVICE 2u /* data/valid are device pointers (e.g. torch tensors) */
typedef struct polr_col {
	const void *data;
	const uint8_t *valid;
	uint32_t width;
	uint32_t flags;
} polr_col;

/* ---------------------------------------------------------------------------------------------
 * Context
 * ------------------------------------------------------------------------------------------- */
int polr_abi_version(void);
int polr_ctx_create(int device_id, polr_ctx **out);
void polr_ctx_destroy(polr_ctx *ctx);
const char *polr_last_error(const polr_ctx *ctx);
int polr_ctx_sync(polr_ctx *ctx, void *stream);
/* The context's own stream (what `stream == NULL` means for every call except the multiplexer runs, which default to a
 * stream of the first multiplexer): pass it to polr_mpx_run_resident* to put a run in line with polr_out_reset before
 * it and polr_out_aggregate* behind it -- a pass with its sink is then one ordered sequence, no host synchronisation. */
int polr_ctx_get_stream(polr_ctx *ctx, void **stream);
/* Tuning of the pool launch (polr_mpx_run_resident*): how a context's runs cut rounds into units and how its probe
 * waves poll.  A field left 0 keeps the library's default; NULL restores every default.  Results never depend on any
 * of these (only speed does), except watchdog_us, which bounds how long a router waits for its probe waves before the
 * run is given up with POLR_E_HIP.  Takes effect with the next run of the context. */
typedef struct polr_pool_tuning {
	uint32_t device_share; /* size every run's grid for 1/device_share of the device (1..16); 0: as the run's flags say */
	uint32_t units_x;      /* a big round is cut into about units_x x probe waves / routing executors units (1..4; default 4) */
	uint32_t hi_unit;      /* tuples per unit of a small round: 64..1024, a multiple of 64 (default: 1024 flat; generic 64,
	                          with more than 64 executors 256) */
	uint32_t hi_lottery;   /* power of two: wave w of a ring tries for hi ticket t only if w % lottery == t % lottery */
	uint32_t hi_tuples_p1; /* 1 + the size up to which a round counts as small (default 4096); 0: default */
	uint32_t idle_sleep;   /* 16: an idle probe wave's back-off stays at s_sleep 16 (default 64) */
	uint32_t watchdog_us;  /* microseconds; default 4 000 000 */
	uint32_t share_after;  /* work sharing (pipelines with repeated build keys): a probe wave that has spent this many
	                          steps on ONE unit gives half of what it still has to do -- the rest of its source range,
	                          of a run of build rows, or of the tuples waiting between two joins -- to the pool, and
	                          again after as many steps (16..65535; default 32: measured 16 / 32 / 64 / 128 on the 113 JOB-shaped
	                          pipelines); 0xFFFFFFFF: never */
} polr_pool_tuning;
int polr_ctx_set_pool_tuning(polr_ctx *ctx, const polr_pool_tuning *tuning);

/* ---------------------------------------------------------------------------------------------
 * Build sides.  Replaces what JoinHashTable::Finalize leaves in host memory
 * (src/execution/join_hashtable.cpp:324-377) with a device-native layout:
 *   - keys/payload de-serialised into SoA columns indexed by build row,
 *   - an open-addressing bucket array: 8-byte {key,row} slots for a unique 32-bit key, 16-byte
 *     {key64,start,count} slots + a row-id run array when the key repeats (bucket-contiguous
 *     instead of pointer-chained: the reference's chain order is not part of its contract,
 *     join_hashtable.cpp:284-303),
 *   - or the direct-address perfect table of perfect_hash_join_executor.cpp:20-122.
 * Build row ids reported anywhere in this API are the ordinals of the rows as uploaded.
 * ------------------------------------------------------------------------------------------- */

/* Upload the reference's row-format blob (RowLayout, src/common/types/row_layout.cpp:19-80;
 * JoinHashTable ctor join_hashtable.cpp:43-57): n_rows rows of row_width bytes, column i at
 * col_offset[i] (keys first, then payload), validity bits at the front of each row.  The
 * reference's own bucket array is not needed: the device re-buckets. */
int polr_ht_upload_rows(polr_ctx *ctx, const void *rows, uint64_t n_rows, uint32_t row_width,
                        const uint32_t *col_offset, const uint32_t *col_width, const uint32_t *col_flags,
                        uint32_t n_keys, uint32_t n_payload, polr_ht **out);
/* Same from columnar build data (DataChunks as sunk by PhysicalHashJoin::Sink,
 * physical_hash_join.cpp:217-286).  Rows with a NULL key are dropped (join_hashtable.cpp:170-192) unless the
 * column is POLR_KEY_NULL_EQUAL; row ids still refer to the rows as passed. */
int polr_ht_upload_columns(polr_ctx *ctx, const polr_col *keys, uint32_t n_keys, const polr_col *payload,
                           uint32_t n_payload, uint64_t n_rows, polr_ht **out);
/* Key semantics beyond "same type on both sides, NULL never matches": the reference multiplexes INNER hash joins whose
 * left side is CAST(column) (src/parallel/polar_config.cpp:75-82) and whose comparison is IS NOT DISTINCT FROM
 * (JoinHashTable::null_values_are_equal, src/execution/join_hashtable.cpp:35-36,170-192,642).  Per key column, BEFORE the
 * table is finalized:
 *   POLR_KEY_BY_VALUE    compare this column BY VALUE whatever integer type the probe side reads (an integer CAST on
 *                        either side of the condition): widths and signedness of the two sides may differ, a probe
 *                        value the build side does not hold never matches.  No cast copy of the probe column is made.
 *   POLR_KEY_NULL_EQUAL  IS NOT DISTINCT FROM: NULL = NULL on this column; build rows whose key is NULL there are kept.
 * A table with such a column is a hash table with its key in packed form (polr_ht_finalize_hash, _auto;
 * polr_ht_finalize_perfect refuses: the reference plans perfect hash joins for plain equalities only). */
#define POLR_KEY_BY_VALUE 1u
#define POLR_KEY_NULL_EQUAL 2u
int polr_ht_set_key_flags(polr_ht *ht, uint32_t key_col, uint32_t flags);
/* Finalize as a hash table (JoinHashTable::Finalize / InsertHashes, join_hashtable.cpp:305-377). */
int polr_ht_finalize_hash(polr_ht *ht, void *stream);
/* Finalize as a perfect hash table (BuildPerfectHashTable, perfect_hash_join_executor.cpp:20-122):
 * keys outside [min,max] are skipped; a duplicate inside the range returns POLR_E_DUPLICATE and
 * leaves the handle un-finalized so the caller can polr_ht_finalize_hash it instead. */
int polr_ht_finalize_perfect(polr_ht *ht, int64_t min_value, int64_t max_value, void *stream);
/* Let the device pick the index: a single integer key whose value range [min,max] (the build column's
 * statistics, as PerfectHashJoinStats carries them: build_min / build_max) is at most POLR_DENSE_FACTOR x
 * the build rows -- or at most 1 000 000 values, the reference planner's own bound (plan_comparison_join.cpp:118,125) --
 * becomes a perfect table (1 bit per key value: 3 M dense keys = 375 KB, L2-resident, where a
 * bucket table would cost one random 64-B HBM line per probe) -- the reference's own perfect-hash idea without
 * its 1 M-value cap (perfect_hash_join_executor.cpp:25-50), which was sized for CPU caches; anything else,
 * and a range with a duplicate key, becomes a hash table.  Match sets are identical either way; *kind_out
 * (may be NULL) tells which build-id convention applies (see polr_ht_get_info). */
#define POLR_DENSE_FACTOR 8
int polr_ht_finalize_auto(polr_ht *ht, int64_t min_value, int64_t max_value, void *stream, uint32_t *kind_out);
/* Upload a perfect table the reference already built (PerfectHashJoinExecutor members
 * perfect_hash_table / bitmap_build_idx, perfect_hash_join_executor.hpp): bitmap has range+1
 * bools, each payload column has range+1 cells. */
int polr_pht_upload(polr_ctx *ctx, uint32_t key_width, uint32_t key_flags, int64_t min_value, int64_t max_value,
                    const uint8_t *bitmap, const polr_col *payload, uint32_t n_payload, polr_ht **out);
void polr_ht_destroy(polr_ht *ht);

typedef struct polr_ht_info {
	uint32_t kind;      /* 0 = not finalized, 1 = perfect, 2 = hash {key,row} slots, 3 = hash {key,start,count} slots */
	uint32_t n_keys;
	uint64_t n_rows;    /* rows kept (NULL keys dropped) */
	uint64_t capacity;  /* hash slots, or range+1 */
	uint64_t max_run;   /* longest duplicate run */
	uint64_t device_bytes;
	uint32_t is_dense;  /* perfect: every slot of the range filled and no NULL build key */
	uint32_t has_null;
} polr_ht_info;
int polr_ht_get_info(const polr_ht *ht, polr_ht_info *info);
/* Broadcast support (RCCL, one exchange per build side): device pointers + byte sizes of the
 * buffers that make up the finalized table, so the caller (torch.distributed, backend "nccl")
 * can broadcast them in place.  The receiving rank first creates a same-shaped empty table with
 * polr_ht_alloc_like from the metadata blob. */
int polr_ht_export(const polr_ht *ht, void *meta, uint64_t *meta_bytes, void **dev_ptrs, uint64_t *dev_bytes,
                   uint32_t *n_buffers);
int polr_ht_alloc_like(polr_ctx *ctx, const void *meta, uint64_t meta_bytes, polr_ht **out);

/* ---- multi-GPU: the one exchange step of the path ----------------------------------
 * One POLAR pipeline per GPU (own multiplexer state, own probe partition) is the counterpart of one PipelineExecutor
 * per worker thread (src/parallel/pipeline.cpp:145-174); the only data every pipeline needs and only one has is the
 * finalized build side (JoinHashTable::Finalize, src/execution/join_hashtable.cpp:324-377).  polr_bcast_build ships
 * it from the rank that built it to all others with ncclBroadcast over xGMI (RCCL, loaded on first use): metadata,
 * then every device buffer in place.  Collective: every rank calls it, in the same order.  root: *ht is the table to
 * send; other ranks: *ht receives a new table (caller destroys it).  The 128-byte id comes from
 * polr_comm_get_unique_id on one rank and reaches the others out of band (the host engine's own channel). */
#define POLR_COMM_ID_BYTES 128
typedef struct polr_comm polr_comm;
int polr_comm_get_unique_id(void *id /* POLR_COMM_ID_BYTES */);
int polr_comm_create(polr_ctx *ctx, const void *id, int world_size, int rank, polr_comm **out);
int polr_bcast_build(polr_comm *comm, polr_ht **ht, int root, void *stream);
int polr_comm_bytes_broadcast(const polr_comm *comm, uint64_t *bytes);
void polr_comm_destroy(polr_comm *comm);

/* ---------------------------------------------------------------------------------------------
 * Pipeline = what POLARConfig::GenerateJoinOrders produces (src/parallel/polar_config.cpp:19-249):
 * the joins of the run, where each join reads its probe key (a probe-table column, or a build
 * column of an earlier join: left_expression_bindings, :152-229) and the candidate join orders.
 * ------------------------------------------------------------------------------------------- */
#define POLR_MAX_PREDS 4 /* non-equality conditions per join */
typedef struct polr_join_desc {
	polr_ht *ht;
	uint32_t n_keys;
	int32_t key_src_join[POLR_MAX_KEYS]; /* -1 = probe-table column, j >= 0 = payload column of join j */
	int32_t key_src_col[POLR_MAX_KEYS];
	/* the join's conditions other than equalities (JoinCondition::comparison; JoinHashTable::predicates,
	 * join_hashtable.cpp:50-52; RowOperations::Match, row_match.cpp:59-119): `left OP right` with the left side read
	 * like a key (pred_src_*) and the right side a payload column of this join's build side; a NULL on either side
	 * never matches; both sides must have the same width.  The join's output (and its share of the intermediates)
	 * are the pairs that pass. */
	uint32_t n_preds;
	uint32_t pred_op[POLR_MAX_PREDS]; /* POLR_CMP_EQ .. POLR_CMP_GE, POLR_CMP_STR_EQ.  EQ is the verifying comparison of a
	                                     composite key the engine hands over HASHED (an 8-byte key column = its hash of the
	                                     key columns, one EQ condition per column): the way in for composite keys whose
	                                     value ranges do not pack into 64 bits */
	int32_t pred_src_join[POLR_MAX_PREDS];
	int32_t pred_src_col[POLR_MAX_PREDS];
	uint32_t pred_build_col[POLR_MAX_PREDS];
} polr_join_desc;

int polr_pipeline_create(polr_ctx *ctx, const polr_col *probe_cols, uint32_t n_probe_cols, uint64_t n_probe_rows,
                         const polr_join_desc *joins, uint32_t k, const int32_t *paths /* n_paths x k */,
                         uint32_t n_paths, polr_pipeline **out);
/* The tuples entering the multiplexer, in scan order, when an upstream filter thinned the source
 * chunks (DICTIONARY/sliced vectors, vector.hpp:36-140): sel[i] = probe-table row.  NULL resets
 * to "all rows".  flags: POLR_COL_DEVICE if sel is a device pointer. */
int polr_pipeline_set_selection(polr_pipeline *p, const uint32_t *sel, uint64_t n_sel, uint32_t flags);
/* How the probe kernels of this pipeline are launched (diagnostic): waves per workgroup, workgroups resident per
 * CU (resident kernel), dynamic LDS bytes per workgroup, compiled stage count K and tuple slots W.
 * materialize != 0: the variant that writes row ids (an `out` object is passed to the runs). */
typedef struct polr_launch_info {
	uint32_t waves_per_workgroup, workgroups_per_cu, lds_bytes_per_workgroup, compiled_stages, tuple_slots, n_cus;
	uint32_t flat;            /* 1: counting runs use the flat pipeline (single-key unique-match joins on probe columns) */
	uint32_t lds_tables;      /* bit tables kept in LDS for the whole run */
	uint32_t lds_table_bytes;
	uint32_t pad;
} polr_launch_info;
int polr_pipeline_launch_info(polr_pipeline *p, int materialize, polr_launch_info *info);

/* ---- source side on the device ---------------------------------------
 * PhysicalTableScan with pushed-down table filters: the table is scanned in vectors of `vector_size`
 * rows (STANDARD_VECTOR_SIZE), every vector is thinned to the rows that pass ALL filters, in row order;
 * a vector without a survivor yields no chunk; NULL passes no comparison
 * (src/storage/table/row_group.cpp:316-452 RowGroup::TemplatedScan, src/storage/table/column_segment.cpp:194-475
 * FilterSelection; filter classes src/include/duckdb/planner/filter/{constant,null,conjunction}_filter.hpp).
 * The result -- selection (ascending probe-table rows) and the boundaries of the non-empty chunks --
 * stays in HBM and becomes the pipeline's source (as polr_pipeline_set_selection would install it);
 * multiplexers take the boundaries with polr_mpx_use_scan_chunks.  n_filters == 0: every row passes.
 * A negative constant against an unsigned column is POLR_E_INVALID; OR-conjunctions are not pushed
 * down on this path (POLR_E_UNSUPPORTED would be the caller's: keep them in a host filter). */
enum {
	POLR_CMP_EQ = 0, POLR_CMP_NE = 1, POLR_CMP_LT = 2, POLR_CMP_GT = 3, POLR_CMP_LE = 4, POLR_CMP_GE = 5,
	POLR_CMP_IS_NULL = 6, POLR_CMP_IS_NOT_NULL = 7,
	/* join conditions only: two columns of 16-byte string cells (string_t) hold the same string.  This is how a VARCHAR
	 * join key arrives: the KEY is the 64-bit hash the engine computes for it anyway (JoinHashTable::Hash,
	 * join_hashtable.cpp:141-155 -- an 8-byte key column on both sides), the strings themselves are this condition --
	 * the hash finds the candidates, the comparison decides, as RowOperations::Match does behind the bucket chain
	 * (row_match.cpp).  The cells' heaps: polr_ht_set_payload_heap / polr_pipeline_set_probe_heap. */
	POLR_CMP_STR_EQ = 8
};
typedef struct polr_scan_filter {
	uint32_t col;     /* probe-table column */
	uint32_t op;      /* POLR_CMP_* (ConstantFilter::comparison_type / IsNullFilter / IsNotNullFilter) */
	int64_t constant; /* ConstantFilter::constant, widened */
} polr_scan_filter;
int polr_pipeline_scan_filter(polr_pipeline *p, void *stream, const polr_scan_filter *filters, uint32_t n_filters,
                              uint32_t vector_size, uint64_t *n_selected, uint64_t *n_chunks);
/* The same scan with LIP (`PRAGMA enable_lip`, lookahead information passing): the source chunks are additionally
 * thinned by the filters of the joins named in `lip_joins` (bit j = join j) before they enter the pipeline --
 * PipelineExecutor::FetchFromSource src/parallel/pipeline_executor.cpp:396-465 probing
 * PhysicalHashJoin::ProbeBloomFilter physical_hash_join.cpp:579-635 for every join with build_bloom_filter
 * (eligibility physical_join.cpp:57-107: one condition, key traced back to a source column).  The reference's filter is
 * a bloom filter sized from a planner estimate; here it is the join's own index (no false positives): the survivors are
 * a subset of the reference's, the pipeline's output rows are the same.  POLR_E_INVALID for a join that is not keyed
 * by one source column. */
int polr_pipeline_scan_filter_lip(polr_pipeline *p, void *stream, const polr_scan_filter *filters, uint32_t n_filters,
                                  uint32_t lip_joins, uint32_t vector_size, uint64_t *n_selected, uint64_t *n_chunks);
/* read the scan result back (tests): sel[n_selected], chunk_offsets[n_chunks + 1]; either may be NULL */
int polr_pipeline_fetch_scan(polr_pipeline *p, uint32_t *sel, uint64_t *chunk_offsets);
/* Refresh the cells of probe column `col` in place (a new DataChunk arriving at the operator-level
 * drop-in, PhysicalHashJoin::Execute physical_hash_join.cpp:637-681): n_rows <= the row count the
 * pipeline was created with; becomes the new tuple count.  Only for columns the library owns (created
 * from host pointers). */
int polr_pipeline_update_probe(polr_pipeline *p, uint32_t col, const void *data, const uint8_t *valid,
                               uint64_t n_rows);
void polr_pipeline_destroy(polr_pipeline *p);

/* ---------------------------------------------------------------------------------------------
 * Output: a stream of fixed-capacity chunks of row ids (late materialisation).  Slot 0 = probe
 * row, slot 1+j = build row of join j in the ORIGINAL join order, i.e. PhysicalAdaptiveUnion
 * (src/execution/operator/polr/physical_adaptive_union.cpp:37-76) is already applied.
 * ------------------------------------------------------------------------------------------- */
/* max_chunks must cover outputs/chunk_capacity plus one partially filled chunk per emitting wave (at most the
 * resident waves of the device, <= 8192); too small -> POLR_E_OVERFLOW with exact counters, retry larger. */
int polr_out_create(polr_pipeline *p, uint32_t chunk_capacity, uint64_t max_chunks, polr_out **out);
int polr_out_reset(polr_out *o, void *stream);
int polr_out_stats(polr_out *o, void *stream, uint64_t *n_rows, uint64_t *n_chunks, uint32_t *overflowed);
/* compacted ids to the host: dst[n_rows][1+k] (slot-major per row) */
int polr_out_fetch_ids(polr_out *o, void *stream, uint32_t *dst, uint64_t dst_rows);
/* Gather one output column for every output row (RowOperations::Gather, row_gather.cpp:16-173 /
 * DataChunk::Slice for probe columns): src_join = -1 -> probe column src_col, else payload column
 * src_col of join src_join.  dst_data/dst_valid are host or device pointers (dst_flags). */
int polr_out_materialize(polr_out *o, void *stream, int32_t src_join, uint32_t src_col, void *dst_data,
                         uint8_t *dst_valid, uint64_t dst_rows, uint32_t dst_flags);

/* ---- sink side on the device -------------------------------------------
 * PhysicalUngroupedAggregate over the pipeline's output (src/execution/operator/aggregate/
 * physical_ungrouped_aggregate.cpp): COUNT(*), COUNT(x), SUM(x), MIN(x), MAX(x) over an integer column x of
 * the probe table (src_join = -1) or of a build side, gathered by the output row ids and reduced on the
 * device -- no column is materialised, only the results leave.  NULLs take no part; SUM is exact in 128 bits
 * (DuckDB: SUM(INTEGER|BIGINT) -> HUGEINT, sum.cpp:113-144) and, like MIN / MAX, NULL over no rows; COUNT is
 * never NULL.  The output object must have been filled by a materialising run (polr_probe_rounds / polr_mpx_run*
 * with an `out`).  VARCHAR / unsigned 64-bit columns: POLR_E_UNSUPPORTED. */
enum { POLR_AGG_COUNT_STAR = 0, POLR_AGG_COUNT = 1, POLR_AGG_SUM = 2, POLR_AGG_MIN = 3, POLR_AGG_MAX = 4 };
typedef struct polr_agg_spec {
	uint32_t fn;       /* POLR_AGG_* */
	int32_t src_join;  /* -1 = probe-table column, j >= 0 = payload column of join j (ignored for COUNT(*)) */
	uint32_t src_col;
} polr_agg_spec;
typedef struct polr_agg_value {
	int64_t lo, hi;    /* the value as a two's complement 128-bit integer (hi:lo) */
	uint64_t count;    /* rows that took part */
	uint32_t is_null;
	uint32_t pad;
} polr_agg_value;
int polr_out_aggregate(polr_out *o, void *stream, const polr_agg_spec *specs, uint32_t n_aggs,
                       polr_agg_value *results);
/* The same aggregates GROUPed BY up to 3 integer columns with small dense domains -- the case the reference
 * plans as PhysicalPerfectHashAggregate (src/execution/operator/aggregate/physical_perfecthash_aggregate.cpp:
 * every group column has a known [min, max] from its statistics; SSB Q4.x: GROUP BY d_year, c_nation).  Group
 * g = mixed-radix number of the key offsets, ((k0 - min0) * n1 + (k1 - min1)) * n2 + ...; results[g * n_aggs + a];
 * a group no row fell into has count 0 (COUNT) / is_null (SUM, MIN, MAX) and is simply absent from the
 * reference's result.  Rows whose group key is NULL or outside its domain are not aggregated; *n_dropped (may
 * be NULL) counts them (the caller sizes the domains from the build columns' statistics, so normally 0).
 * At most 2^20 groups. */
typedef struct polr_group_key {
	int32_t src_join;   /* -1 = probe-table column, j >= 0 = payload column of join j */
	uint32_t src_col;
	int64_t min_value;
	uint32_t n_values;  /* max - min + 1 */
	uint32_t pad;
} polr_group_key;
int polr_out_aggregate_grouped(polr_out *o, void *stream, const polr_group_key *keys, uint32_t n_keys,
                               const polr_agg_spec *specs, uint32_t n_aggs, polr_agg_value *results,
                               uint64_t n_groups, uint64_t *n_dropped);
/* The general GROUP BY sink -- group columns of ANY integer domain (what the reference plans as PhysicalHashAggregate,
 * src/execution/operator/aggregate/physical_hash_aggregate.cpp, when the columns' statistics do not allow the perfect-hash
 * form): a hash table of groups on the device, NULL a group value of its own (group_nulls: bit c = column c is NULL; its
 * group_keys entry is then 0).  cols: src_join / src_col of polr_group_key (min_value, n_values ignored).  The groups come
 * back in no particular order: group_keys[g * n_cols + c], results[g * n_aggs + a], *n_groups of them; POLR_E_OVERFLOW
 * (with *n_groups = what was found until then) when there are more than max_groups. */
int polr_out_aggregate_hashed(polr_out *o, void *stream, const polr_group_key *cols, uint32_t n_cols,
                              const polr_agg_spec *specs, uint32_t n_aggs, uint64_t max_groups, int64_t *group_keys,
                              uint32_t *group_nulls, polr_agg_value *results, uint64_t *n_groups);
/* The same GROUP BY FUSED into the run (SSB-skew Q4.1 as shipped: benchmark/ssb-skew/queries/q4-1.sql): an output object
 * with a fused sink makes the pipeline's LAST join fold every surviving tuple into the group cells instead of writing its
 * row ids -- nothing of the join result is written or read back.  For FLAT pipelines whose joins are all perfect tables
 * (polr_pipeline_launch_info(p, 1): flat) -- the star joins of SSB; COUNT(*), COUNT and SUM over columns of at most 4
 * bytes, at most 4096 groups; POLR_E_UNSUPPORTED otherwise (then: polr_out_aggregate_grouped over the emitted row ids).
 * polr_out_reset zeroes the cells; every run with this `out` adds to them; polr_out_fused_result reads them (results as
 * for polr_out_aggregate_grouped).  keys == NULL un-fuses. */
int polr_out_fuse_grouped(polr_out *o, const polr_group_key *keys, uint32_t n_keys, const polr_agg_spec *specs,
                          uint32_t n_aggs);
int polr_out_fused_result(polr_out *o, void *stream, polr_agg_value *results, uint64_t n_groups, uint64_t *n_dropped);
/* ---- VARCHAR columns and their sink (every JOB query ends in MIN of a VARCHAR: benchmark/imdb_plan_cost/queries/18a.sql:1-3) --------
 * A width-16 column holds string_t cells (src/include/duckdb/common/types/string_type.hpp:23-28: 4-byte length; up to 12
 * characters inline; longer: a 4-byte prefix and an 8-byte pointer into a string heap).  RowOperations::Gather copies such
 * cells as they are, the pointer staying into the table's heap (row_gather.cpp:47-86).  A column with non-inlined strings
 * needs that heap on the device: these calls copy the `heap_bytes` bytes at `heap_base` (the address range the uploaded
 * cells' pointers lie in) into HBM and rebase the pointers of the column's cells; owned by the table / pipeline. */
int polr_ht_set_payload_heap(polr_ht *ht, uint32_t payload_col, const void *heap_base, uint64_t heap_bytes);
int polr_pipeline_set_probe_heap(polr_pipeline *p, uint32_t probe_col, const void *heap_base, uint64_t heap_bytes);
/* MIN / MAX (fn = POLR_AGG_MIN / POLR_AGG_MAX) of a VARCHAR column over the pipeline's output rows, reduced on the
 * device (src/function/aggregate/distributive/minmax.cpp over string_t: bytes compared as unsigned, a proper prefix
 * sorts first; NULLs take no part).  The winning string's bytes go to dst (at most dst_cap of them), *len = its whole
 * length; *is_null = 1 when no row had a non-NULL value. */
int polr_out_aggregate_string(polr_out *o, void *stream, uint32_t fn, int32_t src_join, uint32_t src_col, char *dst,
                              uint32_t dst_cap, uint32_t *len, uint32_t *is_null);
void polr_out_destroy(polr_out *o);

/* ---------------------------------------------------------------------------------------------
 * Probe: RunPath for many routed slices at once (polar_pipeline_executor.cpp:427-538 +
 * PhysicalHashJoin::Execute physical_hash_join.cpp:637-681 + JoinHashTable::Probe /
 * ScanStructure::NextInnerJoin join_hashtable.cpp:396-565 + ProbePerfectHashTable
 * perfect_hash_join_executor.cpp:177-291).  A round = tuples [begin, begin+count) (positions in
 * scan order) sent down join order `path`.  counts[r*k + j] receives the number of tuples the join
 * at position j of that path produced: their sum over j is what RunPath feeds to
 * PhysicalMultiplexer::AddNumIntermediates (:486-487).  out may be NULL (count only).
 * ------------------------------------------------------------------------------------------- */
typedef struct polr_round {
	uint64_t begin;
	uint64_t count;
	uint32_t path;
	uint32_t emit; /* 0: count only (ALTERNATE drops the output of paths != 0, :445-447) */
} polr_round;

int polr_probe_rounds(polr_pipeline *p, void *stream, const polr_round *rounds, uint32_t n_rounds, polr_out *out,
                      uint64_t *counts /* host, n_rounds x k, filled after an internal sync */);
/* asynchronous form: counts stay on the device (n_rounds x k uint64), nothing is synchronised */
int polr_probe_rounds_async(polr_pipeline *p, void *stream, const polr_round *rounds, uint32_t n_rounds,
                            polr_out *out, uint64_t *counts_dev);

/* ---------------------------------------------------------------------------------------------
 * Device-resident multiplexer: PhysicalMultiplexer + RoutingStrategy + the reward update
 * (src/execution/operator/polr/physical_multiplexer.cpp:100-184, routing_strategy.cpp:7-463)
 * evaluated on the device between probe rounds -- by the last workgroup of a path-kernel launch
 * (polr_mpx_run, _run_many) or by the router wave of a resident launch (polr_mpx_run_resident) -- so a
 * whole morsel is routed without a host round trip per routing decision.  Decisions are bit-identical
 * to the host classes in duckdb-polr_amd/host (same source, same double arithmetic, no contraction).
 * ------------------------------------------------------------------------------------------- */
typedef struct polr_mpx_config {
	uint32_t routing;
	uint32_t chunk_size;       /* STANDARD_VECTOR_SIZE of the host engine (1024 in the reference snapshot) */
	double regret_budget;
	uint64_t init_tuple_count;
	uint64_t atc_multiplier;
	uint32_t log_rounds;       /* keep (path, tuples, intermediates) of every routing round */
	uint32_t max_log_rounds;
} polr_mpx_config;

typedef struct polr_mpx_stats {
	uint64_t num_tuples_processed;
	uint64_t num_intermediates;
	uint64_t num_rounds;
	uint64_t input_tuple_count_per_path[POLR_MAX_PATHS];
	double path_resistances[POLR_MAX_PATHS];
	/* tuples produced by the join at position j of join order p, summed over all rounds routed to p
	 * (the per-operator `elements` the reference's profiler reports, query_profiler.cpp:383-404) */
	uint64_t stage_out[POLR_MAX_PATHS][POLR_MAX_JOINS];
} polr_mpx_stats;

int polr_mpx_create(polr_pipeline *p, const polr_mpx_config *cfg, polr_mpx **out);
/* the source chunks are those of the pipeline's polr_pipeline_scan_filter result (boundaries stay in HBM);
 * to be called again after every new scan of the pipeline (a run with a stale attachment is POLR_E_INVALID) */
int polr_mpx_use_scan_chunks(polr_mpx *m);
/* Route and probe source chunks [chunk_begin, chunk_end) (chunk c = tuples [c*chunk_size, ...)
 * unless chunk offsets were set) entirely on the device; asynchronous. */
int polr_mpx_run(polr_mpx *m, void *stream, uint64_t chunk_begin, uint64_t chunk_end, polr_out *out);
/* Several executors at once: ms[i] routes chunks [chunk_begin[i], chunk_end[i]) with its own multiplexer
 * state on its own stream (streams may be NULL: every multiplexer owns one) -- the counterpart of the
 * reference's worker threads, one PipelineExecutor + MultiplexerState each (pipeline.cpp:145-174,
 * pipeline_executor.cpp:28-41).  One host thread pumps all of them, so their routing rounds overlap on the
 * device.  All multiplexers must belong to the same pipeline; `out` (may be NULL) is shared. */
int polr_mpx_run_many(polr_mpx **ms, void **streams, const uint64_t *chunk_begin, const uint64_t *chunk_end,
                      uint32_t n, polr_out *out);
/* The same run as ONE kernel launch ("resident"): the grid is split between the n executors
 * (workgroup % n; with n = 8 one executor per XCD), every executor has a router workgroup that keeps its
 * multiplexer state in LDS and probe workgroups that wait for its rounds on the device -- no launch and no
 * host step between two routing decisions.  Same routing, same results as polr_mpx_run / _run_many with the
 * same chunk ranges.  All executors run on `stream` (NULL: the first multiplexer's).  Asynchronous;
 * polr_mpx_finish(_many) synchronises and reports POLR_E_HIP if the device-side watchdog fired.
 * flags: POLR_RUN_RESET = start from a fresh MultiplexerState (what polr_mpx_reset does, without a launch
 * of its own); POLR_RUN_FINISH = close the run inside the same launch (PushFinalize's FinalizePathRun) and
 * leave the statistics where polr_mpx_finish(_many) picks them up without launching anything.
 * POLR_E_UNSUPPORTED: more than 64 executors / more executors than fit on the device at once. */
#define POLR_RUN_RESET 1u
#define POLR_RUN_FINISH 2u
/* POLR_RUN_SHARE(d): size the grid for 1/d of the device (d = 2..16), so that d resident runs on d streams are
 * co-resident and their rounds overlap (the small exploration rounds of one pass run beside the table-sized
 * round of another) */
#define POLR_RUN_SHARE(d) (((uint32_t)(d) & 0xFFu) << 8)
int polr_mpx_run_resident(polr_mpx **ms, void *stream, const uint64_t *chunk_begin, const uint64_t *chunk_end,
                          uint32_t n, polr_out *out, uint32_t flags);
/* The same with a LIST of chunk ranges per executor (range_begin / range_end: [n][ranges_per_executor], 1..8): executor i
 * routes its ranges one after the other with one multiplexer state -- a worker thread that was handed several morsels
 * up front.  Pairing a range from every part of a skewed table per executor evens out what the executors have to do
 * (they finish together) and keeps runs reproducible, which a shared cursor does not. */
int polr_mpx_run_resident_ranges(polr_mpx **ms, void *stream, const uint64_t *range_begin, const uint64_t *range_end,
                                 uint32_t ranges_per_executor, uint32_t n, polr_out *out, uint32_t flags);

/* Morsel-driven variant: the n executors SHARE the chunks [chunk_begin, chunk_end) and pull them `morsel_chunks`
 * chunks at a time from one device-side cursor -- the reference's worker threads pulling morsels from the
 * parallel scan state (a row group = 120 vectors; pipeline.cpp:145-174, table_scan.cpp) -- so a skewed source
 * cannot leave one executor with the expensive end of the table.  Which executor sees which morsel depends on
 * timing (as in the multi-threaded reference): result row set and COUNT(*) are deterministic, per-executor traces
 * and total intermediates are not. */
int polr_mpx_run_resident_morsels(polr_mpx **ms, void *stream, uint64_t chunk_begin, uint64_t chunk_end,
                                  uint32_t morsel_chunks, uint32_t n, polr_out *out, uint32_t flags);
/* MultiplexerRouting::BACKPRESSURE (src/parallel/pipeline.cpp:147-156, src/parallel/polar_config.cpp:128-147): the
 * reference schedules ONE task per join order over a single shared source state, so the join orders race for the
 * source and the cheaper ones end up with more of it.  ms: one multiplexer per join order of the pipeline (created with
 * POLR_ROUTE_BACKPRESSURE); executor i sends every morsel it pulls (morsel_chunks source chunks at a time; 120 = a row
 * group) down join order i.  Row set and COUNT(*) are deterministic, the split between the orders depends on timing
 * (as in the reference).  Statistics: executor i reports its tuples under path 0 of ITS statistics, its per-stage
 * counters under path i. */
int polr_mpx_run_backpressure(polr_mpx **ms, void *stream, uint64_t chunk_begin, uint64_t chunk_end,
                              uint32_t morsel_chunks, polr_out *out, uint32_t flags);
int polr_mpx_set_chunk_offsets(polr_mpx *m, const uint64_t *offsets, uint64_t n_chunks);
/* fresh MultiplexerState (a new PipelineExecutor / a new pass over the source) */
int polr_mpx_reset(polr_mpx *m, void *stream);
/* bracket every path-kernel launch of polr_mpx_run with HIP events on the launch stream and report
 * the summed device time and launch count since the last call (measurement only) */
int polr_mpx_enable_timing(polr_mpx *m, int enable);
int polr_mpx_kernel_time(polr_mpx *m, double *total_ms, uint64_t *n_launches);
/* PushFinalize's last FinalizePathRun (polar_pipeline_executor.cpp:150-151) + read back */
int polr_mpx_finish(polr_mpx *m, void *stream, polr_mpx_stats *stats);
/* the same for every executor of a polr_mpx_run_many (on the streams the runs used); stats[n] */
int polr_mpx_finish_many(polr_mpx **ms, uint32_t n, polr_mpx_stats *stats);
int polr_mpx_fetch_log(polr_mpx *m, void *stream, uint32_t *path, uint64_t *tuples, uint64_t *intermediates,
                       uint64_t max_rounds, uint64_t *n_rounds);
void polr_mpx_destroy(polr_mpx *m);

#ifdef __cplusplus
}
#endif
#endif
