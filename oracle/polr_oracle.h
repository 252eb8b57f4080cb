/*
 * oracle/polr_oracle.h -- CPU restatement of the reference's POLAR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under duckdb-polr_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there only
 * as the checker / the CPU baseline -- never as the thing shipped.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * root, d-justen/duckdb-polr).  Parity of this restatement is pinned by
 *   - the reference's own known-answer tests (test/polr/polr-minimal.test, test/polr/polr.test),
 *   - traces produced by the reference itself, compiled from its sources by oracle/ref_build.mk
 *     and driven by oracle/ref_driver.cpp (fixtures under tests/golden/).
 */
#ifndef POLR_ORACLE_H
#define POLR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t idx_t;

#define ORC_MAX_JOINS 16
#define ORC_MAX_PATHS 64
#define ORC_MAX_KEYS 4
#define ORC_MAX_PREDS 4
#define ORC_MAX_VECTOR 2048

/* MultiplexerRouting, src/include/duckdb/main/config.hpp:41-50 (same order) */
enum {
	ORC_ROUTE_ALTERNATE = 0,
	ORC_ROUTE_ADAPTIVE_REINIT = 1,
	ORC_ROUTE_DYNAMIC = 2,
	ORC_ROUTE_INIT_ONCE = 3,
	ORC_ROUTE_OPPORTUNISTIC = 4,
	ORC_ROUTE_DEFAULT_PATH = 5,
	ORC_ROUTE_BACKPRESSURE = 6,
	ORC_ROUTE_EXPONENTIAL_BACKOFF = 7
};

/* JoinEnumerator, src/include/duckdb/common/enums/join_enumerator.hpp:15-25 */
enum {
	ORC_ENUM_DFS_RANDOM = 0,
	ORC_ENUM_DFS_MIN_CARD = 1,
	ORC_ENUM_DFS_UNCERTAIN = 2,
	ORC_ENUM_BFS_RANDOM = 3,
	ORC_ENUM_BFS_MIN_CARD = 4,
	ORC_ENUM_BFS_UNCERTAIN = 5,
	ORC_ENUM_EACH_LAST_ONCE = 6,
	ORC_ENUM_EACH_FIRST_ONCE = 7,
	ORC_ENUM_SAMPLE = 8
};

/* A flat column: `data` holds n values of `width` bytes; `valid` is NULL (all valid) or one byte
 * per row (1 = valid).  width 16 = a DuckDB string_t treated as an opaque 16-byte cell. */
typedef struct orc_col {
	const void *data;
	const uint8_t *valid;
	int32_t width;
	int32_t is_signed;
} orc_col_t;

/* ---- hashing: src/include/duckdb/common/types/hash.hpp:22-43, src/common/types/hash.cpp:10-37,
 *      src/common/vector_operations/vector_hash.cpp:14-24 ---- */
uint64_t orc_murmurhash64(uint64_t x);
uint64_t orc_hash_value(const void *value, int width, int is_signed);
uint64_t orc_combine_hash(uint64_t a, uint64_t b);

/* ---- chained hash table: src/execution/join_hashtable.cpp:16-57 (layout), :194-282 (Build),
 *      :284-377 (InsertHashes/Finalize), src/common/types/row_layout.cpp:19-80 ---- */
typedef struct orc_ht orc_ht_t;
orc_ht_t *orc_ht_build(const orc_col_t *keys, int n_keys, const orc_col_t *payload, int n_payload, idx_t n_rows);
/* null_equal[c] != 0: IS NOT DISTINCT FROM on key column c (NULL = NULL; such rows stay in the table) */
orc_ht_t *orc_ht_build2(const orc_col_t *keys, int n_keys, const orc_col_t *payload, int n_payload, idx_t n_rows,
                        const int *null_equal);
void orc_ht_free(orc_ht_t *ht);
idx_t orc_ht_count(const orc_ht_t *ht);
idx_t orc_ht_capacity(const orc_ht_t *ht);
idx_t orc_ht_row_width(const orc_ht_t *ht);
idx_t orc_ht_pointer_offset(const orc_ht_t *ht);
idx_t orc_ht_col_offset(const orc_ht_t *ht, int col); /* keys first, then payload, then the hash/next slot */
int orc_ht_has_null(const orc_ht_t *ht);
const uint8_t *orc_ht_rows(const orc_ht_t *ht);       /* count x row_width bytes, reference row format */
const uint32_t *orc_ht_orig_rows(const orc_ht_t *ht); /* HT row ordinal -> build-table row */
/* bucket array as row ordinals (UINT64_MAX = empty), for upload tests; caller frees with orc_free */
uint64_t *orc_ht_bucket_heads(const orc_ht_t *ht);
void orc_free(void *p);

/* ---- perfect hash table: src/execution/operator/join/perfect_hash_join_executor.cpp:20-122 ---- */
typedef struct orc_pht orc_pht_t;
/* min/max are the build-key statistics as the planner hands them over (raw 64-bit patterns of the
 * key type).  Returns NULL when the build has a duplicate key inside the range (reference falls
 * back to the chained table, :112-114). */
orc_pht_t *orc_pht_build(const orc_ht_t *ht, int64_t min_value, int64_t max_value);
void orc_pht_free(orc_pht_t *pht);
int orc_pht_is_dense(const orc_pht_t *pht);
idx_t orc_pht_range(const orc_pht_t *pht);               /* build_range = max - min */
const uint8_t *orc_pht_bitmap(const orc_pht_t *pht);     /* range+1 bools */
const uint32_t *orc_pht_orig_rows(const orc_pht_t *pht); /* idx -> build-table row (valid where bitmap) */

/* ---- one join of the multiplexed run ---- */
typedef struct orc_join {
	orc_ht_t *ht;
	orc_pht_t *pht; /* non-NULL: perfect-hash probe path (physical_hash_join.cpp:647-650) */
	int32_t n_keys;
	int32_t key_src_join[ORC_MAX_KEYS]; /* -1: probe-table column; j>=0: payload column of join j */
	int32_t key_src_col[ORC_MAX_KEYS];
	idx_t estimated_cardinality; /* for the min-card enumerators */
	/* non-equality conditions of the join (JoinHashTable::predicates, join_hashtable.cpp:50-52): evaluated by
	 * RowOperations::Match together with the equalities (row_match.cpp:59-119, 141-263); the right-hand side is a
	 * column the build side carries (here: payload column pred_build_col of this join's table) */
	int32_t n_preds;
	int32_t pred_op[ORC_MAX_PREDS];       /* ORC_CMP_NE .. ORC_CMP_GE (below): left OP right */
	int32_t pred_src_join[ORC_MAX_PREDS]; /* left side, as key_src_join / key_src_col */
	int32_t pred_src_col[ORC_MAX_PREDS];
	int32_t pred_build_col[ORC_MAX_PREDS];
} orc_join_t;


typedef struct orc_config {
	int32_t routing;
	double regret_budget;
	idx_t init_tuple_count;
	idx_t atc_multiplier;
	int32_t caching;           /* ClientConfig::caching (PRAGMA disable_caching clears it) */
	int32_t log_tuples_routed; /* record per-round intermediates */
	idx_t vector_size;         /* STANDARD_VECTOR_SIZE of the build being restated (1024 in this snapshot) */
	int32_t collect_output;    /* 0: count only; 1: keep output row ids */
	int32_t trailing_operators; /* pass-through operators after the join run (1 = the PROJECTION every
	                               fixture plan has; 0 = the sink follows the joins directly) */
} orc_config_t;

typedef struct orc_result {
	idx_t num_intermediates; /* POLARPipelineExecutor::num_intermediates_produced */
	idx_t num_output_rows;
	idx_t input_tuple_count_per_path[ORC_MAX_PATHS];
	/* log_tuples_routed: one entry per FinalizePathRun (physical_multiplexer.cpp:138-140) */
	idx_t n_rounds;
	idx_t *intermediates_per_round;
	/* ALTERNATE: matrix [n_alt_rows][n_paths] (physical_multiplexer.cpp:142-146,194-208) */
	idx_t n_alt_rows;
	idx_t *alt_matrix;
	/* routing trace: one entry per multiplexer decision (path, tuples) incl. cache-skip chunks */
	idx_t n_trace;
	uint32_t *trace_path;
	uint32_t *trace_tuples;
	/* output row ids, num_output_rows x (1 + k): probe row, then build-table row per join in the
	 * ORIGINAL join order (i.e. after PhysicalAdaptiveUnion) */
	uint32_t *out_rows;
	idx_t n_sink_chunks;
} orc_result_t;

/* Runs source -> multiplexer -> {paths} -> adaptive union -> sink over the probe tuples.
 * `sel`/`n_sel`: the tuples entering the multiplexer, in scan order (NULL: all rows 0..n_probe-1).
 * `chunk_offsets` (n_chunks+1 entries into sel order) gives the source chunk boundaries (NULL:
 * fixed vector_size chunks).  Restates polar_pipeline_executor.cpp:80-538 +
 * pipeline_executor.cpp:88-167 for a pipeline whose only operators are the multiplexed joins. */
int orc_run_pipeline(const orc_col_t *probe_cols, int n_probe_cols, idx_t n_probe_rows, const uint32_t *sel,
                     idx_t n_sel, const idx_t *chunk_offsets, idx_t n_chunks, const orc_join_t *joins, int k,
                     const int32_t *paths, int n_paths, const orc_config_t *cfg, orc_result_t *res);
void orc_result_free(orc_result_t *res);

/* Output column materialisation for parity checks (join_hashtable.cpp:521-529,558-562 +
 * row_gather.cpp:16-86; perfect_hash_join_executor.cpp:200-206).  src_join = -1: probe column. */
/* ---- source side: table scan with pushed-down filters ------------------------------------------
 * RowGroup::TemplatedScan (src/storage/table/row_group.cpp:316-452): the table is read one vector of
 * `vector_size` rows at a time; every pushed-down filter thins the vector's selection in turn
 * (ColumnSegment::FilterSelection, src/storage/table/column_segment.cpp:304-475; a comparison keeps row idx iff
 * mask.RowIsValid(idx) && OP(vec[idx], constant), TemplatedFilterSelection :194-206; IS [NOT] NULL
 * TemplatedNullSelection); a vector whose selection runs empty is skipped (row_group.cpp:399-416), otherwise the
 * survivors form one chunk, in row order.
 * Output: sel[] = surviving rows (ascending), chunk_offsets[] = positions in sel where a chunk starts, plus
 * the end; returns the number of chunks, *n_sel = survivors.  Both arrays must hold n_rows (+1) entries. */
enum { ORC_CMP_EQ = 0, ORC_CMP_NE = 1, ORC_CMP_LT = 2, ORC_CMP_GT = 3, ORC_CMP_LE = 4, ORC_CMP_GE = 5,
       ORC_CMP_IS_NULL = 6, ORC_CMP_IS_NOT_NULL = 7 };
typedef struct orc_filter {
	int32_t col;
	int32_t op;
	int64_t constant;
} orc_filter_t;
idx_t orc_scan_filter(const orc_col_t *cols, idx_t n_rows, const orc_filter_t *filters, int n_filters,
                      idx_t vector_size, uint32_t *sel, idx_t *n_sel, idx_t *chunk_offsets);

int orc_materialize_column(const uint32_t *out_rows, idx_t n_out, int k, int src_join, const orc_col_t *col,
                           uint8_t *dst_data, uint8_t *dst_valid);

/* ---- routing in isolation (for strategy parity tests): feed intermediates, read decisions ---- */
typedef struct orc_mpx orc_mpx_t;
orc_mpx_t *orc_mpx_create(int n_paths, int routing, double regret_budget, idx_t init_tuple_count,
                          idx_t atc_multiplier);
void orc_mpx_free(orc_mpx_t *m);
/* PhysicalMultiplexer::Execute (physical_multiplexer.cpp:100-121): returns 1 for HAVE_MORE_OUTPUT */
int orc_mpx_execute(orc_mpx_t *m, idx_t input_size, idx_t *slice_offset, idx_t *slice_count, idx_t *path,
                    idx_t *cache_skips);
void orc_mpx_add_intermediates(orc_mpx_t *m, idx_t n);
void orc_mpx_increase_input(orc_mpx_t *m, idx_t n);
void orc_mpx_finalize_path_run(orc_mpx_t *m);
void orc_mpx_resistances(const orc_mpx_t *m, double *out);
/* CalculateJoinPathWeights, routing_strategy.cpp:267-316 */
void orc_join_path_weights(const double *costs, int n, double regret_budget, double *weights);

/* ---- POLARConfig::GenerateJoinOrders pieces (polar_config.cpp:19-249) ---- */
/* dependencies[i*k + j] = 1 when join i requires join j (polar_config.cpp:72-95).  Returns number
 * of join orders written to paths (n x k), path 0 is the original order. */
int orc_enumerate(int enumerator, int k, const uint8_t *dependencies, const idx_t *estimated_cardinality,
                  int max_join_orders, int32_t *paths);
/* left_expression_bindings (polar_config.cpp:152-229): for path p, position j, condition c: the
 * column index of the probe key inside that path's intermediate layout, or -1 when the key is a
 * probe-table column (unchanged).  num_build_cols[j] = columns join j adds. */
void orc_bindings(int k, int n_probe_cols, const int32_t *num_build_cols, const orc_join_t *joins,
                  const int32_t *paths, int n_paths, int32_t *bindings /* [n_paths][k][ORC_MAX_KEYS] */);

#ifdef __cplusplus
}
#endif
#endif
