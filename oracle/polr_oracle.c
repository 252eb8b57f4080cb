/*
 * oracle/polr_oracle.c -- CPU restatement of the reference's POLAR hot path (see polr_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, imported by, or called from the product path.
 * Plain C11; citations are file:line in the reference tree (d-justen/duckdb-polr).
 */
#include "polr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OP_NEED_MORE_INPUT 0
#define OP_HAVE_MORE_OUTPUT 1
#define OP_FINISHED 2

#define IDX_MAX UINT64_MAX

/* ===================================================================================== */
/* Hashing                                                                               */
/* ===================================================================================== */

/* src/include/duckdb/common/types/hash.hpp:22-29 */
uint64_t orc_murmurhash64(uint64_t x) {
	x ^= x >> 32;
	x *= 0xd6e8feb86659fd93ULL;
	x ^= x >> 32;
	x *= 0xd6e8feb86659fd93ULL;
	x ^= x >> 32;
	return x;
}

/* Loads a `width`-byte integer as the uint64 the reference's Hash<T> would feed murmurhash64:
 * hash.hpp:31-38 (generic T -> murmurhash32(uint32_t) i.e. T converted to uint32 first, so narrow
 * signed types sign-extend to 32 bits and are then zero-extended), hash.cpp:10-18 (64-bit). */
static uint64_t load_hash_input(const void *p, int width, int is_signed) {
	switch (width) {
	case 1:
		return is_signed ? (uint64_t)(uint32_t)(int32_t) * (const int8_t *)p : (uint64_t) * (const uint8_t *)p;
	case 2: {
		uint16_t v;
		memcpy(&v, p, 2);
		return is_signed ? (uint64_t)(uint32_t)(int32_t)(int16_t)v : (uint64_t)v;
	}
	case 4: {
		uint32_t v;
		memcpy(&v, p, 4);
		return (uint64_t)v;
	}
	default: {
		uint64_t v;
		memcpy(&v, p, 8);
		return v;
	}
	}
}

uint64_t orc_hash_value(const void *value, int width, int is_signed) {
	return orc_murmurhash64(load_hash_input(value, width, is_signed));
}

/* src/common/vector_operations/vector_hash.cpp:22-24 (CombineHashScalar) */
uint64_t orc_combine_hash(uint64_t a, uint64_t b) {
	return (a * 0xbf58476d1ce4e5b9ULL) ^ b;
}

/* raw bit pattern of a cell, zero-extended: key equality on the path is same-type equality
 * (join_hashtable.cpp:24 asserts left/right types equal), so pattern equality == value equality */
static uint64_t load_pattern(const void *p, int width) {
	uint64_t v = 0;
	memcpy(&v, p, width > 8 ? 8 : width);
	return v;
}

/* typed value for range checks (perfect hash): signed types sign-extended into int64 */
static int64_t load_signed(const void *p, int width) {
	switch (width) {
	case 1:
		return *(const int8_t *)p;
	case 2: {
		int16_t v;
		memcpy(&v, p, 2);
		return v;
	}
	case 4: {
		int32_t v;
		memcpy(&v, p, 4);
		return v;
	}
	default: {
		int64_t v;
		memcpy(&v, p, 8);
		return v;
	}
	}
}

void orc_free(void *p) {
	free(p);
}

/* ===================================================================================== */
/* Chained hash table                                                                    */
/* ===================================================================================== */

struct orc_ht {
	int n_keys, n_payload;
	int key_width[ORC_MAX_KEYS], key_signed[ORC_MAX_KEYS];
	int payload_width[64], payload_signed[64];
	idx_t offsets[ORC_MAX_KEYS + 64 + 1]; /* keys, payload, hash slot */
	idx_t flag_width, row_width, pointer_offset;
	idx_t count, capacity, bitmask;
	uint8_t *rows;      /* count * row_width */
	uint8_t **hash_map; /* capacity head pointers (NULL = empty) */
	uint32_t *orig_row; /* HT row ordinal -> build table row */
	int has_null;
	int key_null_equal[ORC_MAX_KEYS]; /* JoinHashTable::null_values_are_equal (join_hashtable.cpp:35-36) */
};

static idx_t next_pow2(idx_t v) {
	v--;
	v |= v >> 1;
	v |= v >> 2;
	v |= v >> 4;
	v |= v >> 8;
	v |= v >> 16;
	v |= v >> 32;
	v++;
	return v;
}

/* Build (join_hashtable.cpp:194-282): NULL keys are dropped (PrepareKeys :170-192, has_null set
 * :228-230); rows are serialised as [validity bytes][keys][payload][hash] with no alignment
 * (:43-57, row_layout.cpp:23-53; all our types are constant size so there is no heap pointer);
 * then Finalize (:340-377) walks the rows in block order and pushes each to the front of its
 * bucket chain (InsertHashesLoop<false> :296-301), overwriting the stored hash with the previous
 * head ("next") pointer. capacity = PointerTableCapacity (join_hashtable.hpp:265-267). */
orc_ht_t *orc_ht_build(const orc_col_t *keys, int n_keys, const orc_col_t *payload, int n_payload, idx_t n_rows) {
	return orc_ht_build2(keys, n_keys, payload, n_payload, n_rows, NULL);
}

/* null_equal[c] != 0: key column c is compared with IS NOT DISTINCT FROM (COMPARE_NOT_DISTINCT_FROM,
 * join_hashtable.cpp:35-36): PrepareKeys does not filter its NULLs (:182), on either side; a NULL hashes to
 * HashOp::NULL_HASH (vector_hash.cpp:15-19) and is serialised as an invalid cell of the row; RowOperations::Match lets
 * an invalid probe value match exactly the rows whose cell is invalid (row_match.cpp:73-83). */
#define ORC_NULL_HASH 0xbf58476d1ce4e5b9ull
orc_ht_t *orc_ht_build2(const orc_col_t *keys, int n_keys, const orc_col_t *payload, int n_payload, idx_t n_rows,
                        const int *null_equal) {
	if (n_keys < 1 || n_keys > ORC_MAX_KEYS || n_payload > 63) {
		return NULL;
	}
	orc_ht_t *ht = (orc_ht_t *)calloc(1, sizeof(orc_ht_t));
	for (int i = 0; i < n_keys; i++) {
		ht->key_null_equal[i] = null_equal ? null_equal[i] : 0;
	}
	ht->n_keys = n_keys;
	ht->n_payload = n_payload;
	int ncols = n_keys + n_payload + 1;
	ht->flag_width = (idx_t)(ncols + 7) / 8; /* ValidityBytes::ValidityMaskSize */
	idx_t w = ht->flag_width;
	for (int i = 0; i < n_keys; i++) {
		ht->key_width[i] = keys[i].width;
		ht->key_signed[i] = keys[i].is_signed;
		ht->offsets[i] = w;
		w += (idx_t)keys[i].width;
	}
	for (int i = 0; i < n_payload; i++) {
		ht->payload_width[i] = payload[i].width;
		ht->payload_signed[i] = payload[i].is_signed;
		ht->offsets[n_keys + i] = w;
		w += (idx_t)payload[i].width;
	}
	ht->offsets[n_keys + n_payload] = w;
	ht->pointer_offset = w;
	ht->row_width = w + 8;

	ht->rows = (uint8_t *)malloc((n_rows ? n_rows : 1) * ht->row_width);
	ht->orig_row = (uint32_t *)malloc((n_rows ? n_rows : 1) * sizeof(uint32_t));
	idx_t cnt = 0;
	for (idx_t r = 0; r < n_rows; r++) {
		int is_null = 0;
		for (int i = 0; i < n_keys; i++) {
			if (keys[i].valid && !keys[i].valid[r] && !ht->key_null_equal[i]) {
				is_null = 1;
			}
		}
		if (is_null) {
			ht->has_null = 1;
			continue;
		}
		uint8_t *row = ht->rows + cnt * ht->row_width;
		memset(row, 0xFF, ht->flag_width); /* all valid, row_scatter.cpp:110ff initialises validity to 1s */
		uint64_t h = 0;
		for (int i = 0; i < n_keys; i++) {
			const uint8_t *src = (const uint8_t *)keys[i].data + r * (idx_t)keys[i].width;
			uint64_t hv;
			if (keys[i].valid && !keys[i].valid[r]) { /* (a null-equal column: the row stays, its cell is invalid) */
				row[i / 8] &= (uint8_t) ~(1u << (i % 8));
				memset(row + ht->offsets[i], 0, (size_t)keys[i].width);
				hv = ORC_NULL_HASH;
			} else {
				memcpy(row + ht->offsets[i], src, (size_t)keys[i].width);
				hv = orc_hash_value(src, keys[i].width, keys[i].is_signed);
			}
			h = i == 0 ? hv : orc_combine_hash(h, hv); /* JoinHashTable::Hash :141-155 */
		}
		for (int i = 0; i < n_payload; i++) {
			int col = n_keys + i;
			const uint8_t *src = (const uint8_t *)payload[i].data + r * (idx_t)payload[i].width;
			if (payload[i].valid && !payload[i].valid[r]) {
				row[col / 8] &= (uint8_t) ~(1u << (col % 8)); /* ValidityBytes::SetInvalidUnsafe */
				memset(row + ht->offsets[col], 0, (size_t)payload[i].width);
			} else {
				memcpy(row + ht->offsets[col], src, (size_t)payload[i].width);
			}
		}
		memcpy(row + ht->pointer_offset, &h, 8);
		ht->orig_row[cnt] = (uint32_t)r;
		cnt++;
	}
	ht->count = cnt;
	idx_t want = cnt * 2;
	idx_t floor_cap = (262136 / 8) + 1; /* Storage::BLOCK_SIZE / sizeof(data_ptr_t) + 1 */
	ht->capacity = next_pow2(want > floor_cap ? want : floor_cap);
	ht->bitmask = ht->capacity - 1;
	ht->hash_map = (uint8_t **)calloc(ht->capacity, sizeof(uint8_t *));
	for (idx_t r = 0; r < cnt; r++) {
		uint8_t *row = ht->rows + r * ht->row_width;
		uint64_t h;
		memcpy(&h, row + ht->pointer_offset, 8);
		idx_t slot = h & ht->bitmask;
		uint8_t *prev = ht->hash_map[slot];
		memcpy(row + ht->pointer_offset, &prev, 8);
		ht->hash_map[slot] = row;
	}
	return ht;
}

void orc_ht_free(orc_ht_t *ht) {
	if (!ht) {
		return;
	}
	free(ht->rows);
	free(ht->hash_map);
	free(ht->orig_row);
	free(ht);
}

idx_t orc_ht_count(const orc_ht_t *ht) {
	return ht->count;
}
idx_t orc_ht_capacity(const orc_ht_t *ht) {
	return ht->capacity;
}
idx_t orc_ht_row_width(const orc_ht_t *ht) {
	return ht->row_width;
}
idx_t orc_ht_pointer_offset(const orc_ht_t *ht) {
	return ht->pointer_offset;
}
idx_t orc_ht_col_offset(const orc_ht_t *ht, int col) {
	return ht->offsets[col];
}
int orc_ht_has_null(const orc_ht_t *ht) {
	return ht->has_null;
}
const uint8_t *orc_ht_rows(const orc_ht_t *ht) {
	return ht->rows;
}
const uint32_t *orc_ht_orig_rows(const orc_ht_t *ht) {
	return ht->orig_row;
}
uint64_t *orc_ht_bucket_heads(const orc_ht_t *ht) {
	uint64_t *out = (uint64_t *)malloc(ht->capacity * sizeof(uint64_t));
	for (idx_t i = 0; i < ht->capacity; i++) {
		out[i] = ht->hash_map[i] ? (uint64_t)((ht->hash_map[i] - ht->rows) / ht->row_width) : UINT64_MAX;
	}
	return out;
}

static inline int row_col_valid(const orc_ht_t *ht, const uint8_t *row, int col) {
	return (row[col / 8] >> (col % 8)) & 1;
}

/* ===================================================================================== */
/* Perfect hash table                                                                    */
/* ===================================================================================== */

struct orc_pht {
	const orc_ht_t *ht;
	int64_t min_value, max_value;
	idx_t build_range;
	uint8_t *bitmap;       /* bitmap_build_idx */
	uint32_t *ht_row;      /* idx -> HT row ordinal (what the perfect columns were gathered from) */
	uint32_t *orig_row;    /* idx -> build-table row */
	idx_t unique_keys;
	int is_build_dense;
};

static inline int key_in_range(const orc_pht_t *p, const void *key, int width, int is_signed, idx_t *idx) {
	if (is_signed) {
		int64_t v = load_signed(key, width);
		if (p->min_value <= v && v <= p->max_value) {
			*idx = (idx_t)(v - p->min_value);
			return 1;
		}
	} else {
		uint64_t v = load_pattern(key, width);
		if ((uint64_t)p->min_value <= v && v <= (uint64_t)p->max_value) {
			*idx = (idx_t)(v - (uint64_t)p->min_value);
			return 1;
		}
	}
	return 0;
}

/* BuildPerfectHashTable / FullScanHashTable / TemplatedFillSelectionVectorBuild
 * (perfect_hash_join_executor.cpp:20-122): scan every HT row in block order, keep keys inside
 * [min,max]; first duplicate aborts (:112-114); dense iff unique_keys == range+1 && !has_null (:61-63). */
orc_pht_t *orc_pht_build(const orc_ht_t *ht, int64_t min_value, int64_t max_value) {
	orc_pht_t *p = (orc_pht_t *)calloc(1, sizeof(orc_pht_t));
	p->ht = ht;
	p->min_value = min_value;
	p->max_value = max_value;
	p->build_range = ht->key_signed[0] ? (idx_t)(max_value - min_value) : (idx_t)((uint64_t)max_value - (uint64_t)min_value);
	idx_t size = p->build_range + 1;
	p->bitmap = (uint8_t *)calloc(size, 1);
	p->ht_row = (uint32_t *)malloc(size * sizeof(uint32_t));
	p->orig_row = (uint32_t *)malloc(size * sizeof(uint32_t));
	memset(p->ht_row, 0xFF, size * sizeof(uint32_t));
	memset(p->orig_row, 0xFF, size * sizeof(uint32_t));
	for (idx_t r = 0; r < ht->count; r++) {
		const uint8_t *row = ht->rows + r * ht->row_width;
		idx_t idx;
		if (key_in_range(p, row + ht->offsets[0], ht->key_width[0], ht->key_signed[0], &idx)) {
			if (p->bitmap[idx]) {
				orc_pht_free(p);
				return NULL;
			}
			p->bitmap[idx] = 1;
			p->unique_keys++;
			p->ht_row[idx] = (uint32_t)r;
			p->orig_row[idx] = ht->orig_row[r];
		}
	}
	if (p->unique_keys == p->build_range + 1 && !ht->has_null) {
		p->is_build_dense = 1;
	}
	return p;
}

void orc_pht_free(orc_pht_t *p) {
	if (!p) {
		return;
	}
	free(p->bitmap);
	free(p->ht_row);
	free(p->orig_row);
	free(p);
}
int orc_pht_is_dense(const orc_pht_t *p) {
	return p->is_build_dense;
}
idx_t orc_pht_range(const orc_pht_t *p) {
	return p->build_range;
}
const uint8_t *orc_pht_bitmap(const orc_pht_t *p) {
	return p->bitmap;
}
const uint32_t *orc_pht_orig_rows(const orc_pht_t *p) {
	return p->orig_row;
}

/* ===================================================================================== */
/* Routing strategies + multiplexer                                                      */
/* ===================================================================================== */

struct orc_mpx {
	int P;
	int routing;
	double regret_budget;
	idx_t init_tuple_count;
	idx_t multiplier;
	/* MultiplexerState (physical_multiplexer.cpp:20-82) */
	double path_resistances[ORC_MAX_PATHS];
	double historic_resistances[ORC_MAX_PATHS];
	idx_t input_tuple_count_per_path[ORC_MAX_PATHS];
	int first_mpx_run;
	idx_t num_intermediates_current_path;
	idx_t num_tuples_processed;
	idx_t current_path_tuple_count;
	idx_t current_path_idx;
	idx_t num_cache_flushing_skips;
	int alternate_mode_active;
	/* RoutingStrategyState (routing_strategy.hpp:15-31) */
	idx_t chunk_size, next_path_idx, next_tuple_count, chunk_offset, rs_cache_skips;
	/* InitOnce (:77-85) */
	idx_t best_path_after_init, num_paths_initialized;
	int init_phase_done;
	/* AdaptiveReinit (:101-114) / ExponentialBackoff (:131-146) */
	idx_t window_offset, window_size;
	int visited_paths[ORC_MAX_PATHS];
	idx_t max_window_size;
	idx_t eb_min_resistance_path_idx;
	double eb_min_resistance;
	/* Dynamic (:163-176) */
	idx_t remaining_tuples[ORC_MAX_PATHS];
	int64_t remaining_tuples_diff[ORC_MAX_PATHS];
	double path_weights[ORC_MAX_PATHS];
	/* logs */
	int log_tuples_routed;
	idx_t n_rounds, cap_rounds;
	idx_t *intermediates_per_round;
	idx_t n_alt[ORC_MAX_PATHS], cap_alt[ORC_MAX_PATHS];
	idx_t *alt[ORC_MAX_PATHS];
};

orc_mpx_t *orc_mpx_create(int n_paths, int routing, double regret_budget, idx_t init_tuple_count,
                          idx_t atc_multiplier) {
	orc_mpx_t *m = (orc_mpx_t *)calloc(1, sizeof(orc_mpx_t));
	m->P = n_paths;
	m->routing = routing;
	m->regret_budget = regret_budget;
	m->init_tuple_count = init_tuple_count;
	m->multiplier = atc_multiplier;
	m->first_mpx_run = 1;
	/* ALTERNATE/DEFAULT_PATH/BACKPRESSURE strategies are constructed with init_tuple_count 0
	 * (routing_strategy.hpp:190,204); nothing reads it there. */
	if (routing == ORC_ROUTE_EXPONENTIAL_BACKOFF) {
		m->max_window_size = (idx_t)regret_budget; /* physical_multiplexer.cpp:50-52 */
		m->eb_min_resistance_path_idx = (idx_t)-1;
		m->eb_min_resistance = 1.7976931348623157e308;
	}
	return m;
}

void orc_mpx_free(orc_mpx_t *m) {
	if (!m) {
		return;
	}
	free(m->intermediates_per_round);
	for (int i = 0; i < ORC_MAX_PATHS; i++) {
		free(m->alt[i]);
	}
	free(m);
}

void orc_mpx_resistances(const orc_mpx_t *m, double *out) {
	memcpy(out, m->path_resistances, sizeof(double) * (size_t)m->P);
}

static idx_t argmin_resistance(const orc_mpx_t *m, double *min_out) {
	/* routing_strategy.cpp:38-46 (first strictly smaller wins) */
	double mn = m->path_resistances[0];
	idx_t idx = 0;
	for (int i = 1; i < m->P; i++) {
		if (m->path_resistances[i] < mn) {
			mn = m->path_resistances[i];
			idx = (idx_t)i;
		}
	}
	if (min_out) {
		*min_out = mn;
	}
	return idx;
}

typedef struct {
	double key;
	idx_t idx;
} cost_entry_t;

/* CalculateJoinPathWeights, routing_strategy.cpp:267-316.  The std::multimap<double, idx_t> is
 * restated as a stable insertion sort by key (multimap::emplace inserts at the upper bound of
 * equal keys), walked from the back (rbegin). */
void orc_join_path_weights(const double *costs, int n, double regret_budget, double *weights) {
	cost_entry_t sorted[ORC_MAX_PATHS];
	int cnt = 0;
	for (int i = 0; i < n; i++) {
		int pos = cnt;
		while (pos > 0 && sorted[pos - 1].key > costs[i]) {
			sorted[pos] = sorted[pos - 1];
			pos--;
		}
		sorted[pos].key = costs[i];
		sorted[pos].idx = (idx_t)i;
		cnt++;
	}
	/* path_weights.resize(n, 1): caller pre-fills with 1 (routing_strategy.cpp:333) */
	double cost_bottom = sorted[n - 1].key;
	for (int it = n - 2; it >= 0; it--) {
		const double cost_next = sorted[it].key;
		double cost_next_rounded = round(cost_next / 0.001) * 0.001;
		double cost_bottom_rounded = round(cost_bottom / 0.001) * 0.001;
		if (cost_next_rounded == cost_bottom_rounded) {
			cost_bottom += 0.001;
		}
		double cost_target = cost_next * (1 + regret_budget);
		double cost_avg = (cost_next + cost_bottom) / 2;
		if (cost_target >= cost_avg) {
			cost_target = 0.6 * cost_next + 0.4 * cost_bottom;
		}
		const double path_weight_bottom = (cost_next - cost_target) / (cost_next - cost_bottom);
		for (int it2 = n - 1; it2 > it; it2--) {
			weights[sorted[it2].idx] *= path_weight_bottom;
		}
		weights[sorted[it].idx] = 1 - path_weight_bottom;
		cost_bottom = cost_target;
	}
}

static idx_t determine_next_path(orc_mpx_t *m);

/* InitOnceRoutingStrategy::DetermineNextPath, routing_strategy.cpp:55-82 */
static idx_t init_once_next_path(orc_mpx_t *m) {
	if (m->init_phase_done) {
		m->rs_cache_skips = IDX_MAX;
		return m->best_path_after_init;
	}
	if (m->num_paths_initialized == (idx_t)m->P) {
		m->init_phase_done = 1;
		m->best_path_after_init = argmin_resistance(m, NULL);
		return m->best_path_after_init;
	}
	return m->num_paths_initialized++;
}

/* AdaptiveReinitRoutingStrategy::DetermineNextPath, routing_strategy.cpp:94-180 */
static idx_t adaptive_reinit_next_path(orc_mpx_t *m) {
	double *r = m->path_resistances;
	if (m->init_phase_done) {
		double min_resistance;
		idx_t min_idx = argmin_resistance(m, &min_resistance);
		if (min_resistance * 1.05 >= r[0]) { /* :111-114 */
			min_resistance = r[0];
			min_idx = 0;
		}
		if (m->window_offset == 0 || !m->visited_paths[min_idx]) { /* :116-142 */
			m->visited_paths[min_idx] = 1;
			double reinit_cost_estimate = 0;
			for (int i = 0; i < m->P; i++) {
				if (!m->visited_paths[i]) {
					reinit_cost_estimate += r[i] * (double)m->init_tuple_count;
				}
			}
			if (reinit_cost_estimate == 0) {
				for (int i = 0; i < m->P; i++) {
					m->visited_paths[i] = 0;
				}
				m->visited_paths[min_idx] = 1;
				for (int i = 0; i < m->P; i++) {
					reinit_cost_estimate += r[i] * (double)m->init_tuple_count;
				}
			}
			double tuple_count_before_reinit = reinit_cost_estimate / (m->regret_budget * min_resistance);
			m->window_size = (idx_t)tuple_count_before_reinit; /* idx_t = double, :140-141 */
		}
		if (min_resistance <= 0.525) { /* RESISTANCE_TOLERANCE, routing_strategy.hpp:110; :143-147 */
			m->window_offset = 0;
			return min_idx;
		}
		if (m->window_offset >= m->window_size) { /* :149-163 */
			m->window_offset = 0;
			for (int i = 0; i < m->P; i++) {
				if (!m->visited_paths[i]) {
					r[i] = 0;
					m->init_phase_done = 0;
				} else {
					m->visited_paths[i] = 0;
				}
			}
			m->init_phase_done = 0;
			return adaptive_reinit_next_path(m);
		}
		return min_idx;
	}
	for (int i = 0; i < m->P; i++) { /* :168-174 */
		if (r[i] == 0) {
			return (idx_t)i;
		}
	}
	m->init_phase_done = 1;
	return adaptive_reinit_next_path(m);
}

/* ExponentialBackoffRoutingStrategy::DetermineNextPath, routing_strategy.cpp:198-252 */
static idx_t exp_backoff_next_path(orc_mpx_t *m) {
	double *r = m->path_resistances;
	if (m->init_phase_done) {
		double cur_min;
		idx_t cur_idx = argmin_resistance(m, &cur_min);
		if (m->window_offset == 0) {
			if (m->window_size == 0) {
				m->window_size = 1;
			} else if (cur_idx == m->eb_min_resistance_path_idx ||
			           cur_min * 1.1 >= r[m->eb_min_resistance_path_idx]) {
				idx_t doubled = m->window_size * 2;
				m->window_size = m->max_window_size < doubled ? m->max_window_size : doubled;
			} else {
				m->window_size = 1;
			}
		} else if (m->window_offset >= m->window_size) {
			m->window_offset = 0;
			m->init_phase_done = 0;
			for (int i = 0; i < m->P; i++) {
				if ((idx_t)i != m->eb_min_resistance_path_idx) {
					r[i] = 0;
				}
			}
			return exp_backoff_next_path(m);
		}
		m->eb_min_resistance = cur_min;
		m->eb_min_resistance_path_idx = cur_idx;
		return cur_idx;
	}
	for (int i = 0; i < m->P; i++) {
		if (r[i] == 0) {
			return (idx_t)i;
		}
	}
	m->init_phase_done = 1;
	return exp_backoff_next_path(m);
}

/* DynamicRoutingStrategy::DetermineNextPath, routing_strategy.cpp:318-406 */
static idx_t dynamic_next_path(orc_mpx_t *m) {
	if (m->init_phase_done) {
		idx_t max_remaining = m->remaining_tuples[0];
		idx_t max_idx = 0;
		for (int i = 1; i < m->P; i++) {
			if (m->remaining_tuples[i] > max_remaining) {
				max_remaining = m->remaining_tuples[i];
				max_idx = (idx_t)i;
			}
		}
		if (max_remaining > 0) {
			return max_idx;
		}
		for (int i = 0; i < m->P; i++) {
			m->path_weights[i] = 1;
		}
		orc_join_path_weights(m->path_resistances, m->P, m->regret_budget, m->path_weights);

		idx_t input_tuples = m->chunk_size * m->multiplier - m->chunk_offset;
		idx_t remaining_tuples_sum = 0;
		for (int i = 0; i < m->P; i++) {
			/* int remaining_tuples = diff + std::round(weight * input_tuples)  (:338) */
			int remaining = (int)((double)m->remaining_tuples_diff[i] + round(m->path_weights[i] * (double)input_tuples));
			if (remaining < 0) {
				m->remaining_tuples_diff[i] += (int64_t)m->remaining_tuples[i];
				m->remaining_tuples[i] = 0;
			} else {
				m->remaining_tuples[i] = (idx_t)remaining;
				m->remaining_tuples_diff[i] = 0;
			}
			remaining_tuples_sum += m->remaining_tuples[i];
		}
		idx_t sum_after = 0;
		for (int i = 0; i < m->P; i++) {
			m->remaining_tuples[i] =
			    (idx_t)round((double)m->remaining_tuples[i] / (double)remaining_tuples_sum * (double)input_tuples);
			if (m->remaining_tuples[i] < 64) {
				m->remaining_tuples_diff[i] = (int64_t)m->remaining_tuples[i];
				m->remaining_tuples[i] = 0;
			}
			sum_after += m->remaining_tuples[i];
		}
		if (sum_after != input_tuples) {
			idx_t control_sum = 0, max_normalized = 0, max_normalized_idx = 0;
			for (int i = 0; i < m->P; i++) {
				if (m->remaining_tuples[i] > 0) {
					idx_t normalized =
					    (idx_t)round((double)m->remaining_tuples[i] / (double)sum_after * (double)input_tuples);
					/* diff -= normalized - remaining  (idx_t arithmetic, wraps like the reference) */
					m->remaining_tuples_diff[i] -= (int64_t)(normalized - m->remaining_tuples[i]);
					m->remaining_tuples[i] = normalized;
					control_sum += normalized;
					if (normalized > max_normalized) {
						max_normalized = normalized;
						max_normalized_idx = (idx_t)i;
					}
				}
			}
			if (control_sum != input_tuples) {
				m->remaining_tuples[max_normalized_idx] -= control_sum - (idx_t)(int)input_tuples;
				control_sum -= control_sum - (idx_t)(int)input_tuples;
			}
		}
		return dynamic_next_path(m);
	}
	for (int i = 0; i < m->P; i++) {
		if (m->path_resistances[i] == 0) {
			return (idx_t)i;
		}
	}
	m->init_phase_done = 1;
	return dynamic_next_path(m);
}

static idx_t determine_next_path(orc_mpx_t *m) {
	switch (m->routing) {
	case ORC_ROUTE_OPPORTUNISTIC: /* routing_strategy.cpp:35-49 */
		return argmin_resistance(m, NULL);
	case ORC_ROUTE_INIT_ONCE:
		return init_once_next_path(m);
	case ORC_ROUTE_ADAPTIVE_REINIT:
		return adaptive_reinit_next_path(m);
	case ORC_ROUTE_EXPONENTIAL_BACKOFF:
		return exp_backoff_next_path(m);
	case ORC_ROUTE_DYNAMIC:
		return dynamic_next_path(m);
	case ORC_ROUTE_DEFAULT_PATH:
	case ORC_ROUTE_BACKPRESSURE: /* routing_strategy.cpp:454-457; physical_multiplexer.cpp:47-49 */
		m->rs_cache_skips = IDX_MAX;
		return 0;
	default:
		return 0;
	}
}

static idx_t min_idx(idx_t a, idx_t b) {
	return a < b ? a : b;
}

static idx_t determine_next_tuple_count(orc_mpx_t *m) {
	switch (m->routing) {
	case ORC_ROUTE_OPPORTUNISTIC: /* :51-53 */
	case ORC_ROUTE_DEFAULT_PATH:  /* :459-461 */
	case ORC_ROUTE_BACKPRESSURE:
		return m->chunk_size;
	case ORC_ROUTE_INIT_ONCE: /* :84-92 */
		if (m->init_phase_done) {
			return m->chunk_size - m->chunk_offset;
		}
		return min_idx(m->init_tuple_count, m->chunk_size - m->chunk_offset);
	case ORC_ROUTE_ADAPTIVE_REINIT: /* :182-196 */
		if (m->init_phase_done) {
			if (m->window_offset < m->window_size) {
				m->rs_cache_skips = (idx_t)round((double)m->window_size / (double)m->chunk_size);
				m->window_offset += m->window_size;
			} else {
				m->rs_cache_skips = 0;
			}
			return m->chunk_size - m->chunk_offset;
		}
		m->rs_cache_skips = 0;
		return min_idx(m->init_tuple_count, m->chunk_size - m->chunk_offset);
	case ORC_ROUTE_EXPONENTIAL_BACKOFF: /* :254-265 */
		if (m->init_phase_done) {
			m->rs_cache_skips = m->window_size;
			m->window_offset += m->window_size;
			return m->chunk_size - m->chunk_offset;
		}
		m->rs_cache_skips = 0;
		return min_idx(m->init_tuple_count, m->chunk_size - m->chunk_offset);
	case ORC_ROUTE_DYNAMIC: { /* :408-438 */
		m->rs_cache_skips = 0;
		if (m->init_phase_done) {
			idx_t max_remaining = m->remaining_tuples[0];
			idx_t max_idx = 0;
			for (int i = 1; i < m->P; i++) {
				if (m->remaining_tuples[i] > max_remaining) {
					max_remaining = m->remaining_tuples[i];
					max_idx = (idx_t)i;
				}
			}
			if (max_remaining > 0) {
				idx_t remaining_input = m->chunk_size - m->chunk_offset;
				if (max_remaining > remaining_input) {
					m->rs_cache_skips = (max_remaining - remaining_input) / m->chunk_size;
					m->remaining_tuples[max_idx] -= m->rs_cache_skips * m->chunk_size + remaining_input;
					return remaining_input;
				}
				m->remaining_tuples[max_idx] = 0;
				return max_remaining;
			}
		}
		return min_idx(m->init_tuple_count, m->chunk_size - m->chunk_offset);
	}
	default:
		return m->chunk_size;
	}
}

static void push_idx(idx_t **arr, idx_t *n, idx_t *cap, idx_t v) {
	if (*n == *cap) {
		*cap = *cap ? *cap * 2 : 1024;
		*arr = (idx_t *)realloc(*arr, *cap * sizeof(idx_t));
	}
	(*arr)[(*n)++] = v;
}

/* PhysicalMultiplexer::FinalizePathRun, physical_multiplexer.cpp:132-174 (time_resistance off) */
void orc_mpx_finalize_path_run(orc_mpx_t *m) {
	m->input_tuple_count_per_path[m->current_path_idx] += m->current_path_tuple_count;
	m->num_tuples_processed += m->current_path_tuple_count;
	if (m->log_tuples_routed) {
		push_idx(&m->intermediates_per_round, &m->n_rounds, &m->cap_rounds, m->num_intermediates_current_path);
	}
	if (m->alternate_mode_active) {
		idx_t p = m->current_path_idx;
		push_idx(&m->alt[p], &m->n_alt[p], &m->cap_alt[p], m->num_intermediates_current_path);
		m->num_intermediates_current_path = 0;
		return;
	}
	double constant_overhead = 0.5;
	double path_resistance =
	    (double)m->num_intermediates_current_path / (double)m->current_path_tuple_count + constant_overhead;
	if (m->historic_resistances[m->current_path_idx] != 0) {
		/* SMOOTHING_FACTOR = 0.5, physical_multiplexer.hpp:25 */
		path_resistance = m->historic_resistances[m->current_path_idx] * 0.5 + (1 - 0.5) * path_resistance;
	}
	m->path_resistances[m->current_path_idx] = path_resistance;
	m->historic_resistances[m->current_path_idx] = path_resistance;
	m->num_intermediates_current_path = 0;
}

void orc_mpx_add_intermediates(orc_mpx_t *m, idx_t n) {
	m->num_intermediates_current_path += n;
}
void orc_mpx_increase_input(orc_mpx_t *m, idx_t n) {
	m->current_path_tuple_count += n;
}

/* PhysicalMultiplexer::Execute (physical_multiplexer.cpp:100-121) + RoutingStrategy::Route
 * (routing_strategy.hpp:47-53) + SelectTuples (routing_strategy.cpp:7-33) +
 * AlternateRoutingStrategy::Route (:440-452). */
int orc_mpx_execute(orc_mpx_t *m, idx_t input_size, idx_t *slice_offset, idx_t *slice_count, idx_t *path,
                    idx_t *cache_skips) {
	if (!m->first_mpx_run) {
		orc_mpx_finalize_path_run(m);
	} else {
		m->first_mpx_run = 0;
		if (m->routing == ORC_ROUTE_ALTERNATE) {
			m->alternate_mode_active = 1;
		}
	}
	int result;
	idx_t off = 0;
	if (m->routing == ORC_ROUTE_ALTERNATE) {
		m->next_path_idx = m->next_tuple_count == 0 ? 0 : (m->next_path_idx + 1) % (idx_t)m->P;
		m->next_tuple_count = input_size;
		result = m->next_path_idx == (idx_t)m->P - 1 ? OP_NEED_MORE_INPUT : OP_HAVE_MORE_OUTPUT;
	} else {
		m->chunk_size = input_size;
		m->next_path_idx = determine_next_path(m);
		m->next_tuple_count = determine_next_tuple_count(m);
		if (m->next_tuple_count == input_size) {
			result = OP_NEED_MORE_INPUT; /* chunk.Reference(input) */
		} else {
			off = m->chunk_offset;
			if (m->chunk_offset + m->next_tuple_count == input_size) {
				m->chunk_offset = 0;
				result = OP_NEED_MORE_INPUT;
			} else {
				m->chunk_offset += m->next_tuple_count;
				result = OP_HAVE_MORE_OUTPUT;
			}
		}
	}
	m->current_path_tuple_count = m->next_tuple_count;
	m->current_path_idx = m->next_path_idx;
	m->num_cache_flushing_skips = m->rs_cache_skips;
	*slice_offset = off;
	*slice_count = m->next_tuple_count;
	*path = m->current_path_idx;
	*cache_skips = m->num_cache_flushing_skips;
	return result == OP_HAVE_MORE_OUTPUT;
}

/* ===================================================================================== */
/* Executor                                                                              */
/* ===================================================================================== */

/* A DataChunk on the path, late-materialised: column 0 = probe-table row of each tuple, column
 * 1+pos = the row the join at path position pos matched (HT row ordinal for a chained join, index
 * into the perfect table for a perfect-hash join).  All values the reference would carry in the
 * chunk are reachable through these ids, so slicing/appending ids == slicing/appending the chunk. */
typedef struct {
	idx_t count;
	int ncols;
	uint32_t *cols[1 + ORC_MAX_JOINS];
} chunk_t;

static void chunk_init(chunk_t *c, int ncols, idx_t cap) {
	c->count = 0;
	c->ncols = ncols;
	for (int i = 0; i < 1 + ORC_MAX_JOINS; i++) {
		c->cols[i] = i < ncols ? (uint32_t *)malloc(cap * sizeof(uint32_t)) : NULL;
	}
}
static void chunk_destroy(chunk_t *c) {
	for (int i = 0; i < 1 + ORC_MAX_JOINS; i++) {
		free(c->cols[i]);
		c->cols[i] = NULL;
	}
}
static void chunk_swap(chunk_t *a, chunk_t *b) {
	chunk_t t = *a;
	*a = *b;
	*b = t;
}

/* HashJoinOperatorState + ScanStructure (physical_hash_join.cpp:484-506, join_hashtable.hpp) */
typedef struct {
	int has_scan_structure;
	idx_t count;
	uint32_t *sel_vector;
	uint8_t **pointers;
} join_state_t;

typedef struct {
	const orc_col_t *probe_cols;
	const orc_join_t *joins;
	int k, P;
	const int32_t *paths;
	const orc_config_t *cfg;
	idx_t V;
	orc_mpx_t *mpx;
	/* POLARPipelineExecutor members (polar_pipeline_executor.hpp) */
	chunk_t join_intermediate_chunks[ORC_MAX_PATHS][ORC_MAX_JOINS];
	chunk_t cached_join_chunks[ORC_MAX_PATHS][ORC_MAX_JOINS];
	join_state_t join_states[ORC_MAX_PATHS][ORC_MAX_JOINS];
	chunk_t mpx_output_chunk;
	chunk_t join_out_chunk;
	int in_process_joins[ORC_MAX_JOINS * 4];
	int n_in_process_joins;
	int in_process_operators; /* only the multiplexer can be in process in this pipeline */
	int finalized;
	idx_t num_intermediates_produced;
	const idx_t *current_join_path_dummy;
	/* sink */
	orc_result_t *res;
	idx_t out_cap;
	idx_t trace_cap;
} exec_t;

/* value of key c of join `join_idx` for tuple t of `chunk` when running path `path`
 * (the reference resolves this through the rebound BoundReferenceExpression,
 * physical_hash_join.cpp:541-577 + polar_config.cpp:152-229; execute_reference.cpp:13-24) */
static const uint8_t *key_cell(const exec_t *e, const int32_t *path, const chunk_t *chunk, idx_t t, int join_idx,
                               int c, int *valid) {
	const orc_join_t *j = &e->joins[join_idx];
	int src = j->key_src_join[c];
	int col = j->key_src_col[c];
	if (src < 0) {
		const orc_col_t *pc = &e->probe_cols[col];
		idx_t row = chunk->cols[0][t];
		*valid = pc->valid ? pc->valid[row] : 1;
		return (const uint8_t *)pc->data + row * (idx_t)pc->width;
	}
	int pos = -1;
	for (int q = 0; q < e->k; q++) {
		if (path[q] == src) {
			pos = q;
			break;
		}
	}
	const orc_join_t *sj = &e->joins[src];
	idx_t id = chunk->cols[1 + pos][t];
	idx_t ht_row = sj->pht ? sj->pht->ht_row[id] : id;
	const uint8_t *row = sj->ht->rows + ht_row * sj->ht->row_width;
	int lcol = sj->ht->n_keys + col;
	*valid = row_col_valid(sj->ht, row, lcol);
	return row + sj->ht->offsets[lcol];
}

/* left side of predicate c of join join_idx for tuple t (same sources as the equality keys) */
static const uint8_t *pred_cell(const exec_t *e, const int32_t *path, const chunk_t *chunk, idx_t t, int join_idx,
                                int c, int *valid) {
	const orc_join_t *j = &e->joins[join_idx];
	int src = j->pred_src_join[c];
	int col = j->pred_src_col[c];
	if (src < 0) {
		const orc_col_t *pc = &e->probe_cols[col];
		idx_t row = chunk->cols[0][t];
		*valid = pc->valid ? pc->valid[row] : 1;
		return (const uint8_t *)pc->data + row * (idx_t)pc->width;
	}
	int pos = -1;
	for (int q = 0; q < e->k; q++) {
		if (path[q] == src) {
			pos = q;
			break;
		}
	}
	const orc_join_t *sj = &e->joins[src];
	idx_t id = chunk->cols[1 + pos][t];
	idx_t ht_row = sj->pht ? sj->pht->ht_row[id] : id;
	const uint8_t *row = sj->ht->rows + ht_row * sj->ht->row_width;
	int lcol = sj->ht->n_keys + col;
	*valid = row_col_valid(sj->ht, row, lcol);
	return row + sj->ht->offsets[lcol];
}

static int64_t cell_i64(const uint8_t *p, int width, int is_signed) {
	switch (width) {
	case 1:
		return is_signed ? (int64_t) * (const int8_t *)p : (int64_t)*p;
	case 2: {
		uint16_t v;
		memcpy(&v, p, 2);
		return is_signed ? (int64_t)(int16_t)v : (int64_t)v;
	}
	case 4: {
		uint32_t v;
		memcpy(&v, p, 4);
		return is_signed ? (int64_t)(int32_t)v : (int64_t)v;
	}
	default: {
		int64_t v;
		memcpy(&v, p, 8);
		return v;
	}
	}
}

/* TemplatedMatchType<T, OP> (row_match.cpp:59-119): both sides valid and `left OP right` (a NULL on either side never
 * matches); 8-byte unsigned columns compare as unsigned */
static int pred_holds(int op, const uint8_t *l, const uint8_t *r, int width, int is_signed) {
	if (width == 8 && !is_signed) {
		uint64_t a, b;
		memcpy(&a, l, 8);
		memcpy(&b, r, 8);
		switch (op) {
		case ORC_CMP_NE:
			return a != b;
		case ORC_CMP_LT:
			return a < b;
		case ORC_CMP_GT:
			return a > b;
		case ORC_CMP_LE:
			return a <= b;
		default:
			return a >= b;
		}
	}
	const int64_t a = cell_i64(l, width, is_signed), b = cell_i64(r, width, is_signed);
	switch (op) {
	case ORC_CMP_NE:
		return a != b;
	case ORC_CMP_LT:
		return a < b;
	case ORC_CMP_GT:
		return a > b;
	case ORC_CMP_LE:
		return a <= b;
	default:
		return a >= b;
	}
}

/* ScanStructure::NextInnerJoin (join_hashtable.cpp:531-565) with ScanInnerJoin (:466-487),
 * ResolvePredicates/RowOperations::Match for equality keys (:455-464, row_match.cpp:59-119),
 * AdvancePointers (:489-501).  At most one match per probe tuple per call. */
static void next_inner_join(exec_t *e, const int32_t *path, int pos, join_state_t *st, const chunk_t *left,
                            chunk_t *result) {
	const int join_idx = path[pos];
	const orc_join_t *j = &e->joins[join_idx];
	const orc_ht_t *ht = j->ht;
	if (st->count == 0) {
		return;
	}
	idx_t result_count = 0;
	uint32_t result_vector[ORC_MAX_VECTOR];
	while (1) {
		result_count = 0;
		for (idx_t i = 0; i < st->count; i++) {
			idx_t idx = st->sel_vector[i];
			const uint8_t *row = st->pointers[idx];
			int match = 1;
			for (int c = 0; c < j->n_keys && match; c++) {
				int valid;
				const uint8_t *cell = key_cell(e, path, left, idx, join_idx, c, &valid);
				/* TemplatedMatchType<T, Equals, NO_MATCH_SEL=false>: row validity bit && equal; an invalid probe value (only
				 * a null-equal column lets one get this far) matches exactly the invalid cells (row_match.cpp:73-83) */
				if (!valid) {
					if (row_col_valid(ht, row, c)) {
						match = 0;
					}
				} else if (!row_col_valid(ht, row, c) ||
				           memcmp(cell, row + ht->offsets[c], (size_t)ht->key_width[c]) != 0) {
					match = 0;
				}
			}
			for (int c = 0; c < j->n_preds && match; c++) {
				int valid;
				const uint8_t *cell = pred_cell(e, path, left, idx, join_idx, c, &valid);
				const int lcol = ht->n_keys + j->pred_build_col[c];
				if (!valid || !row_col_valid(ht, row, lcol) ||
				    !pred_holds(j->pred_op[c], cell, row + ht->offsets[lcol], ht->payload_width[j->pred_build_col[c]],
				                ht->payload_signed[j->pred_build_col[c]])) {
					match = 0;
				}
			}
			if (match) {
				result_vector[result_count++] = (uint32_t)idx;
			}
		}
		if (result_count > 0) {
			break;
		}
		/* AdvancePointers() */
		idx_t new_count = 0;
		for (idx_t i = 0; i < st->count; i++) {
			idx_t idx = st->sel_vector[i];
			uint8_t *next;
			memcpy(&next, st->pointers[idx] + ht->pointer_offset, 8);
			st->pointers[idx] = next;
			if (next) {
				st->sel_vector[new_count++] = (uint32_t)idx;
			}
		}
		st->count = new_count;
		if (st->count == 0) {
			return;
		}
	}
	/* result.Slice(left, result_vector) + GatherResult: ids instead of values */
	for (idx_t i = 0; i < result_count; i++) {
		idx_t idx = result_vector[i];
		for (int c = 0; c < 1 + pos; c++) {
			result->cols[c][i] = left->cols[c][idx];
		}
		result->cols[1 + pos][i] = (uint32_t)((st->pointers[idx] - ht->rows) / ht->row_width);
	}
	result->count = result_count;
	/* AdvancePointers() so the next Next() yields the next duplicate (:563) */
	idx_t new_count = 0;
	for (idx_t i = 0; i < st->count; i++) {
		idx_t idx = st->sel_vector[i];
		uint8_t *next;
		memcpy(&next, st->pointers[idx] + ht->pointer_offset, 8);
		st->pointers[idx] = next;
		if (next) {
			st->sel_vector[new_count++] = (uint32_t)idx;
		}
	}
	st->count = new_count;
}

/* PhysicalHashJoin::Execute (physical_hash_join.cpp:637-681), JoinHashTable::Probe
 * (join_hashtable.cpp:396-418), ProbePerfectHashTable (perfect_hash_join_executor.cpp:177-291) */
static int hash_join_execute(exec_t *e, const int32_t *path, int pos, join_state_t *st, const chunk_t *input,
                             chunk_t *out) {
	const int join_idx = path[pos];
	const orc_join_t *j = &e->joins[join_idx];
	const orc_ht_t *ht = j->ht;
	if (ht->count == 0) {
		return OP_FINISHED; /* EmptyResultIfRHSIsEmpty() for INNER (:643-645) */
	}
	if (j->pht) {
		const orc_pht_t *p = j->pht;
		idx_t n = 0;
		for (idx_t i = 0; i < input->count; i++) {
			int valid;
			const uint8_t *cell = key_cell(e, path, input, i, join_idx, 0, &valid);
			if (!valid) {
				continue;
			}
			idx_t idx;
			if (key_in_range(p, cell, ht->key_width[0], ht->key_signed[0], &idx) && p->bitmap[idx]) {
				for (int c = 0; c < 1 + pos; c++) {
					out->cols[c][n] = input->cols[c][i];
				}
				out->cols[1 + pos][n] = (uint32_t)idx;
				n++;
			}
		}
		out->count = n;
		return OP_NEED_MORE_INPUT;
	}
	if (st->has_scan_structure) {
		next_inner_join(e, path, pos, st, input, out);
		if (out->count > 0) {
			return OP_HAVE_MORE_OUTPUT;
		}
		st->has_scan_structure = 0;
		return OP_NEED_MORE_INPUT;
	}
	/* Probe: PrepareKeys drops NULL keys (:170-192); Hash (:141-155); ApplyBitmask (:126-139);
	 * InitializeSelectionVector keeps tuples whose bucket is non-empty (:503-515) */
	st->has_scan_structure = 1;
	st->count = 0;
	for (idx_t i = 0; i < input->count; i++) {
		uint64_t h = 0;
		int all_valid = 1;
		for (int c = 0; c < j->n_keys; c++) {
			int valid;
			const uint8_t *cell = key_cell(e, path, input, i, join_idx, c, &valid);
			if (!valid && !ht->key_null_equal[c]) {
				all_valid = 0;
				break;
			}
			uint64_t hv = valid ? orc_hash_value(cell, ht->key_width[c], ht->key_signed[c]) : ORC_NULL_HASH;
			h = c == 0 ? hv : orc_combine_hash(h, hv);
		}
		if (!all_valid) {
			continue;
		}
		uint8_t *head = ht->hash_map[h & ht->bitmask];
		st->pointers[i] = head;
		if (head) {
			st->sel_vector[st->count++] = (uint32_t)i;
		}
	}
	next_inner_join(e, path, pos, st, input, out);
	return OP_HAVE_MORE_OUTPUT;
}

static void sink_chunk(exec_t *e, const chunk_t *c) {
	/* c is in ORIGINAL join order (after adaptive union): cols[1+x] = id for join x */
	orc_result_t *r = e->res;
	r->n_sink_chunks++;
	if (e->cfg->collect_output) {
		idx_t w = (idx_t)(1 + e->k);
		if (r->num_output_rows + c->count > e->out_cap) {
			while (r->num_output_rows + c->count > e->out_cap) {
				e->out_cap = e->out_cap ? e->out_cap * 2 : 4096;
			}
			r->out_rows = (uint32_t *)realloc(r->out_rows, e->out_cap * w * sizeof(uint32_t));
		}
		for (idx_t i = 0; i < c->count; i++) {
			uint32_t *dst = r->out_rows + (r->num_output_rows + i) * w;
			dst[0] = c->cols[0][i];
			for (int x = 0; x < e->k; x++) {
				const orc_join_t *j = &e->joins[x];
				uint32_t id = c->cols[1 + x][i];
				dst[1 + x] = j->pht ? j->pht->orig_row[id] : j->ht->orig_row[id];
			}
		}
	}
	r->num_output_rows += c->count;
}

/* PhysicalAdaptiveUnion::Execute (physical_adaptive_union.cpp:37-76): the build columns of the join
 * at path position i move to the slot of join path[i] in the original order. */
static void adaptive_union(exec_t *e, const int32_t *path, const chunk_t *input, chunk_t *result) {
	result->count = input->count;
	memcpy(result->cols[0], input->cols[0], input->count * sizeof(uint32_t));
	for (int i = 0; i < e->k; i++) {
		memcpy(result->cols[1 + path[i]], input->cols[1 + i], input->count * sizeof(uint32_t));
	}
}

static void chunk_append(chunk_t *dst, const chunk_t *src) {
	for (int c = 0; c < src->ncols; c++) {
		memcpy(dst->cols[c] + dst->count, src->cols[c], src->count * sizeof(uint32_t));
	}
	dst->count += src->count;
}

/* POLARPipelineExecutor::CacheJoinChunk, polar_pipeline_executor.cpp:166-195
 * (CACHE_THRESHOLD = 64, pipeline_executor.hpp:95) */
static void cache_join_chunk(exec_t *e, chunk_t *current_chunk, int operator_idx) {
	if (e->V < 128 || !e->cfg->caching) {
		return;
	}
	idx_t current_path = e->mpx->current_path_idx;
	chunk_t *cache = &e->cached_join_chunks[current_path][operator_idx];
	if (current_chunk->count < 64) {
		chunk_append(cache, current_chunk);
		if (cache->count >= e->V - 64) {
			chunk_swap(current_chunk, cache); /* current_chunk.Move(*chunk_cache); cache re-initialised */
			cache->count = 0;
		} else {
			current_chunk->count = 0;
		}
	}
}

/* POLARPipelineExecutor::RunPath, polar_pipeline_executor.cpp:427-538 */
static void run_path(exec_t *e, chunk_t *chunk, chunk_t *result, idx_t start_idx_in) {
	orc_mpx_t *mpx = e->mpx;
	idx_t cache_skips_left = mpx->num_cache_flushing_skips; /* by value (:432) */
	idx_t current_path = mpx->current_path_idx;
	const int32_t *path = e->paths + current_path * (idx_t)e->k;
	long start_idx = start_idx_in == IDX_MAX ? -1 : (long)start_idx_in;
	int running_cache = start_idx != 0 && e->n_in_process_joins == 0;
	int must_rerun_cache = 0;
	const int k = e->k;

	if (start_idx == k) {
		if (mpx->routing == ORC_ROUTE_ALTERNATE && current_path != 0) {
			return;
		}
		adaptive_union(e, path, chunk, result);
		return;
	}
	long local_join_idx = start_idx;
	if (e->n_in_process_joins > 0) {
		local_join_idx = e->in_process_joins[--e->n_in_process_joins];
		start_idx = 0;
	}
	while (1) {
		chunk_t *prev_chunk =
		    local_join_idx == start_idx ? chunk : &e->join_intermediate_chunks[current_path][local_join_idx - 1];
		chunk_t *current_chunk = &e->join_intermediate_chunks[current_path][local_join_idx];
		current_chunk->count = 0;
		join_state_t *st = &e->join_states[current_path][local_join_idx];
		int join_result = hash_join_execute(e, path, (int)local_join_idx, st, prev_chunk, current_chunk);

		if (running_cache && local_join_idx == start_idx) {
			must_rerun_cache = join_result == OP_HAVE_MORE_OUTPUT;
		}
		orc_mpx_add_intermediates(mpx, current_chunk->count);
		e->num_intermediates_produced += current_chunk->count;
		if (cache_skips_left != 0) {
			cache_join_chunk(e, current_chunk, (int)local_join_idx);
		}
		if (join_result == OP_HAVE_MORE_OUTPUT) {
			e->in_process_joins[e->n_in_process_joins++] = (int)local_join_idx;
		}
		if (current_chunk->count == 0) {
			if (e->n_in_process_joins > 0) {
				local_join_idx = e->in_process_joins[--e->n_in_process_joins];
				continue;
			}
			break;
		}
		local_join_idx++;
		if (local_join_idx >= k) {
			if (mpx->routing == ORC_ROUTE_ALTERNATE && current_path != 0) {
				if (e->n_in_process_joins > 0) {
					local_join_idx = e->in_process_joins[--e->n_in_process_joins];
					continue;
				}
				break;
			}
			adaptive_union(e, path, current_chunk, result);
			if (must_rerun_cache) {
				chunk_swap(&e->cached_join_chunks[current_path][start_idx - 1],
				           &e->join_intermediate_chunks[current_path][start_idx - 1]);
			}
			return;
		}
	}
}

/* FlushInProcessJoins, polar_pipeline_executor.cpp:197-221 */
static int flush_in_process_joins(exec_t *e, chunk_t *result, int *did_flush) {
	if (e->n_in_process_joins == 0) {
		return OP_FINISHED;
	}
	/* in_process_operators.top() can only be multiplexer_idx+1 here, never greater (:202-207) */
	result->count = 0;
	run_path(e, &e->mpx_output_chunk, result, IDX_MAX);
	*did_flush = 1;
	if (result->count == 0) {
		return OP_FINISHED;
	} else if (e->n_in_process_joins == 0) {
		return OP_NEED_MORE_INPUT;
	}
	return OP_HAVE_MORE_OUTPUT;
}

/* FlushJoinCaches, polar_pipeline_executor.cpp:223-253 */
static int flush_join_caches(exec_t *e, chunk_t *result, int *did_flush) {
	idx_t cache_skips_left = e->mpx->num_cache_flushing_skips;
	if (cache_skips_left > 0 && !e->finalized) {
		return OP_FINISHED;
	}
	idx_t current_path = e->mpx->current_path_idx;
	for (int i = 0; i < e->k; i++) {
		chunk_t *cached = &e->cached_join_chunks[current_path][i];
		if (cached->count > 0) {
			*did_flush = 1;
			result->count = 0;
			run_path(e, cached, result, (idx_t)(i + 1));
			/* after a must_rerun_cache swap `cached` is the old intermediate chunk: Reset() it */
			e->cached_join_chunks[current_path][i].count = 0;
			if (result->count > 0) {
				for (int j = i; j < e->k; j++) {
					if (e->cached_join_chunks[current_path][j].count > 0) {
						return OP_HAVE_MORE_OUTPUT;
					}
				}
				return OP_NEED_MORE_INPUT;
			}
		}
	}
	return OP_FINISHED;
}

static void trace_push(exec_t *e, idx_t path, idx_t tuples) {
	orc_result_t *r = e->res;
	if (r->n_trace == e->trace_cap) {
		e->trace_cap = e->trace_cap ? e->trace_cap * 2 : 1024;
		r->trace_path = (uint32_t *)realloc(r->trace_path, e->trace_cap * sizeof(uint32_t));
		r->trace_tuples = (uint32_t *)realloc(r->trace_tuples, e->trace_cap * sizeof(uint32_t));
	}
	r->trace_path[r->n_trace] = (uint32_t)path;
	r->trace_tuples[r->n_trace] = (uint32_t)tuples;
	r->n_trace++;
}

/* POLARPipelineExecutor::Execute(input, result, initial_idx), polar_pipeline_executor.cpp:255-425,
 * for a pipeline whose operator list is [MULTIPLEXER, <n_trailing pass-through operators>]
 * (multiplexer_idx = 0, initial_idx = 0).  Operator index 1 is the multiplexer; indices 2.. are the
 * operators after the join run (a PROJECTION in every plan the fixtures come from), restated as
 * identity operators that return NEED_MORE_INPUT and need no cache (RequiresCache() false).
 * `join_out` is intermediate_chunks[multiplexer_idx + 1] when trailing operators exist, else it
 * aliases `result` (:261-263).  With NO trailing operator the reference returns straight out of
 * the join-output branch (:282-290) without looking at in_process_joins -- restated as is. */
static int polar_execute(exec_t *e, const chunk_t *input, chunk_t *result) {
	orc_mpx_t *mpx = e->mpx;
	const idx_t n_ops = 1 + (idx_t)e->cfg->trailing_operators;
	chunk_t *join_out = e->cfg->trailing_operators ? &e->join_out_chunk : result;
	int did_flush = 0;
	int op_result = flush_in_process_joins(e, join_out, &did_flush);
	if (op_result == OP_FINISHED) {
		op_result = flush_join_caches(e, join_out, &did_flush);
	}
	idx_t current_idx;
	if (op_result == OP_FINISHED) {
		if (did_flush && !e->in_process_operators) {
			return OP_NEED_MORE_INPUT;
		}
		if (input->count == 0) {
			return OP_NEED_MORE_INPUT;
		}
		/* GoToSource (pipeline_executor.cpp:299-310) */
		current_idx = 0;
		if (e->in_process_operators) {
			current_idx = 1;
			e->in_process_operators = 0;
		}
		if (current_idx == 0) {
			current_idx++;
		}
	} else {
		/* join output: jump to the operator after the join sequence (:282-290) */
		current_idx = 2;
		if (current_idx > n_ops) {
			return e->in_process_operators ? OP_HAVE_MORE_OUTPUT : op_result;
		}
	}
	idx_t current_path = mpx->current_path_idx;
	while (1) {
		chunk_t *current_chunk = (current_idx == 1 && e->cfg->trailing_operators) ? join_out : result;
		if (current_idx == 1 && !e->cfg->trailing_operators) {
			current_chunk = result;
		}
		current_chunk->count = 0;
		if (current_idx == 0) {
			break;
		}
		if (current_idx == 1) {
			/* the multiplexer (:320-366) */
			if (mpx->num_cache_flushing_skips > 0) {
				chunk_t *m = &e->mpx_output_chunk;
				m->count = input->count;
				memcpy(m->cols[0], input->cols[0], input->count * sizeof(uint32_t));
				orc_mpx_increase_input(mpx, m->count);
				trace_push(e, mpx->current_path_idx, m->count);
				run_path(e, m, current_chunk, 0);
				mpx->num_cache_flushing_skips--;
			} else {
				idx_t off, cnt, path, skips;
				int more = orc_mpx_execute(mpx, input->count, &off, &cnt, &path, &skips);
				chunk_t *m = &e->mpx_output_chunk;
				m->count = cnt;
				memcpy(m->cols[0], input->cols[0] + off, cnt * sizeof(uint32_t));
				trace_push(e, path, cnt);
				if (more) {
					e->in_process_operators = 1;
				}
				current_path = mpx->current_path_idx;
				run_path(e, m, current_chunk, 0);
				if (current_chunk->count == 0) {
					if (mpx->num_cache_flushing_skips == 0) {
						for (int i = 0; i < e->k; i++) {
							if (e->cached_join_chunks[current_path][i].count > 0) {
								return OP_HAVE_MORE_OUTPUT;
							}
						}
					}
				}
			}
		} else {
			/* pass-through operator after the joins (:367-386): Execute(prev -> current), NEED_MORE_INPUT */
			const chunk_t *prev = current_idx == 2 ? join_out : result;
			if (prev != current_chunk) {
				for (int c = 0; c < 1 + e->k; c++) {
					memcpy(current_chunk->cols[c], prev->cols[c], prev->count * sizeof(uint32_t));
				}
				current_chunk->count = prev->count;
			}
		}
		if (current_chunk->count == 0) {
			/* GoToSource */
			current_idx = 0;
			if (e->in_process_operators) {
				current_idx = 1;
				e->in_process_operators = 0;
			}
			continue;
		}
		current_idx++;
		if (current_idx > n_ops) {
			break;
		}
	}
	if (e->in_process_operators || e->n_in_process_joins > 0 || op_result == OP_HAVE_MORE_OUTPUT) {
		return OP_HAVE_MORE_OUTPUT;
	}
	if (mpx->num_cache_flushing_skips == 0 || e->finalized) {
		for (int i = 0; i < e->k; i++) {
			if (e->cached_join_chunks[current_path][i].count > 0) {
				return OP_HAVE_MORE_OUTPUT;
			}
		}
	}
	return OP_NEED_MORE_INPUT;
}

/* PipelineExecutor::ExecutePushInternal, pipeline_executor.cpp:134-167 */
static void execute_push_internal(exec_t *e, const chunk_t *input, chunk_t *final_chunk) {
	if (input->count == 0) {
		return;
	}
	while (1) {
		final_chunk->count = 0;
		int result = polar_execute(e, input, final_chunk);
		if (final_chunk->count > 0) {
			sink_chunk(e, final_chunk);
		}
		if (result == OP_NEED_MORE_INPUT) {
			return;
		}
	}
}

int orc_run_pipeline(const orc_col_t *probe_cols, int n_probe_cols, idx_t n_probe_rows, const uint32_t *sel,
                     idx_t n_sel, const idx_t *chunk_offsets, idx_t n_chunks, const orc_join_t *joins, int k,
                     const int32_t *paths, int n_paths, const orc_config_t *cfg, orc_result_t *res) {
	(void)n_probe_cols;
	if (k < 1 || k > ORC_MAX_JOINS || n_paths < 1 || n_paths > ORC_MAX_PATHS || cfg->vector_size > ORC_MAX_VECTOR) {
		return -1;
	}
	memset(res, 0, sizeof(*res));
	exec_t *e = (exec_t *)calloc(1, sizeof(exec_t));
	e->probe_cols = probe_cols;
	e->joins = joins;
	e->k = k;
	e->P = n_paths;
	e->paths = paths;
	e->cfg = cfg;
	e->V = cfg->vector_size;
	e->res = res;
	e->mpx = orc_mpx_create(n_paths, cfg->routing, cfg->regret_budget, cfg->init_tuple_count, cfg->atc_multiplier);
	e->mpx->log_tuples_routed = cfg->log_tuples_routed;
	idx_t V = e->V;
	for (int p = 0; p < n_paths; p++) {
		for (int j = 0; j < k; j++) {
			chunk_init(&e->join_intermediate_chunks[p][j], 2 + j, V);
			chunk_init(&e->cached_join_chunks[p][j], 2 + j, V);
			e->join_states[p][j].sel_vector = (uint32_t *)malloc(V * sizeof(uint32_t));
			e->join_states[p][j].pointers = (uint8_t **)malloc(V * sizeof(uint8_t *));
		}
	}
	chunk_init(&e->mpx_output_chunk, 1, V);
	chunk_init(&e->join_out_chunk, 1 + k, V);
	chunk_t source_chunk, final_chunk;
	chunk_init(&source_chunk, 1, V);
	chunk_init(&final_chunk, 1 + k, V);

	/* NO_OUTPUT_POSSIBLE (polar_pipeline_executor.cpp:63-68): an empty build side finishes the
	 * pipeline before any tuple is routed */
	int finished = 0;
	for (int j = 0; j < k; j++) {
		if (joins[j].ht->count == 0) {
			finished = 1;
		}
	}

	idx_t total = sel ? n_sel : n_probe_rows;
	if (!chunk_offsets) {
		n_chunks = (total + V - 1) / V;
	}
	/* PipelineExecutor::Execute(max_chunks), pipeline_executor.cpp:88-115 */
	for (idx_t c = 0; c < n_chunks && !finished; c++) {
		idx_t begin = chunk_offsets ? chunk_offsets[c] : c * V;
		idx_t end = chunk_offsets ? chunk_offsets[c + 1] : (begin + V < total ? begin + V : total);
		if (end == begin) {
			continue; /* the scan never emits an empty chunk mid-table */
		}
		source_chunk.count = end - begin;
		for (idx_t i = begin; i < end; i++) {
			source_chunk.cols[0][i - begin] = sel ? sel[i] : (uint32_t)i;
		}
		execute_push_internal(e, &source_chunk, &final_chunk);
	}
	if (!finished) {
		/* POLARPipelineExecutor::PushFinalize, polar_pipeline_executor.cpp:111-164 */
		e->finalized = 1;
		idx_t current_path = e->mpx->current_path_idx;
		for (int i = 0; i < k; i++) {
			if (e->cached_join_chunks[current_path][i].count > 0) {
				execute_push_internal(e, &e->mpx_output_chunk, &final_chunk);
			}
		}
		if (!e->mpx->first_mpx_run) {
			orc_mpx_finalize_path_run(e->mpx);
		}
	}
	res->num_intermediates = e->num_intermediates_produced;
	for (int p = 0; p < n_paths; p++) {
		res->input_tuple_count_per_path[p] = e->mpx->input_tuple_count_per_path[p];
	}
	res->n_rounds = e->mpx->n_rounds;
	res->intermediates_per_round = e->mpx->intermediates_per_round;
	e->mpx->intermediates_per_round = NULL;
	if (e->mpx->alternate_mode_active) {
		/* WriteLogToFile (physical_multiplexer.cpp:194-208): rows = front().size() */
		idx_t rows = e->mpx->n_alt[0];
		res->n_alt_rows = rows;
		res->alt_matrix = (idx_t *)calloc(rows * (idx_t)n_paths + 1, sizeof(idx_t));
		for (idx_t i = 0; i < rows; i++) {
			for (int p = 0; p < n_paths; p++) {
				res->alt_matrix[i * (idx_t)n_paths + (idx_t)p] = i < e->mpx->n_alt[p] ? e->mpx->alt[p][i] : 0;
			}
		}
	}
	for (int p = 0; p < n_paths; p++) {
		for (int j = 0; j < k; j++) {
			chunk_destroy(&e->join_intermediate_chunks[p][j]);
			chunk_destroy(&e->cached_join_chunks[p][j]);
			free(e->join_states[p][j].sel_vector);
			free(e->join_states[p][j].pointers);
		}
	}
	chunk_destroy(&e->mpx_output_chunk);
	chunk_destroy(&e->join_out_chunk);
	chunk_destroy(&source_chunk);
	chunk_destroy(&final_chunk);
	orc_mpx_free(e->mpx);
	free(e);
	return 0;
}

void orc_result_free(orc_result_t *res) {
	free(res->intermediates_per_round);
	free(res->alt_matrix);
	free(res->trace_path);
	free(res->trace_tuples);
	free(res->out_rows);
	memset(res, 0, sizeof(*res));
}

/* Gather of one output column by row id.  src_join = -1: the probe column sliced by the probe row
 * ids (result.Slice, join_hashtable.cpp:555); otherwise build column `col` of join src_join
 * indexed by build-table row (values identical to RowOperations::Gather from the HT row,
 * row_gather.cpp:16-45; NULL -> validity 0, cell zeroed). */
int orc_materialize_column(const uint32_t *out_rows, idx_t n_out, int k, int src_join, const orc_col_t *col,
                           uint8_t *dst_data, uint8_t *dst_valid) {
	idx_t w = (idx_t)(1 + k);
	int slot = src_join < 0 ? 0 : 1 + src_join;
	for (idx_t i = 0; i < n_out; i++) {
		idx_t row = out_rows[i * w + (idx_t)slot];
		int valid = col->valid ? col->valid[row] : 1;
		if (dst_valid) {
			dst_valid[i] = (uint8_t)valid;
		}
		if (valid) {
			memcpy(dst_data + i * (idx_t)col->width, (const uint8_t *)col->data + row * (idx_t)col->width,
			       (size_t)col->width);
		} else {
			memset(dst_data + i * (idx_t)col->width, 0, (size_t)col->width);
		}
	}
	return 0;
}

/* ===================================================================================== */
/* Join-order enumeration + bindings                                                     */
/* ===================================================================================== */

static int can_join(const int32_t *r, int nr, int s, int k, const uint8_t *deps) {
	/* JoinEnumerationAlgo::CanJoin, polar_enumeration_algo.cpp:126-135 */
	for (int q = 0; q < k; q++) {
		if (deps[s * k + q]) {
			int found = 0;
			for (int i = 0; i < nr; i++) {
				if (r[i] == q) {
					found = 1;
				}
			}
			if (!found) {
				return 0;
			}
		}
	}
	return 1;
}

/* MinCardinalitySelector::SelectNextCandidate, polar_enumeration_algo.cpp:18-30 */
static int select_min_card(const int32_t *cand, int n, const idx_t *card) {
	idx_t min_card = IDX_MAX;
	int selected = 0;
	for (int i = 0; i < n; i++) {
		if (card[cand[i]] < min_card) {
			min_card = card[cand[i]];
			selected = cand[i];
		}
	}
	return selected;
}

static void erase_value(int32_t *v, int *n, int value) {
	for (int i = 0; i < *n; i++) {
		if (v[i] == value) {
			memmove(v + i, v + i + 1, (size_t)(*n - i - 1) * sizeof(int32_t));
			(*n)--;
			return;
		}
	}
}

typedef struct {
	int k, max_orders, n;
	const uint8_t *deps;
	const idx_t *card;
	int32_t *out;
} enum_ctx_t;

/* DFSEnumeration::GeneratePathsRecursive, polar_enumeration_algo.cpp:152-189 */
static void dfs_recursive(enum_ctx_t *c, const int32_t *seq, int nseq, const int32_t *left, int nleft) {
	if (c->n >= c->max_orders) {
		return;
	}
	int32_t cand[ORC_MAX_JOINS];
	int ncand = 0;
	for (int i = 0; i < nleft; i++) {
		if (can_join(seq, nseq, left[i], c->k, c->deps)) {
			cand[ncand++] = left[i];
		}
	}
	int num_relations = ncand;
	for (int i = 0; i < num_relations; i++) {
		int join_idx = select_min_card(cand, ncand, c->card);
		erase_value(cand, &ncand, join_idx);
		int32_t seq_new[ORC_MAX_JOINS];
		memcpy(seq_new, seq, (size_t)nseq * sizeof(int32_t));
		seq_new[nseq] = join_idx;
		if (nleft == 1) {
			/* result.push_back is unconditional here (:180-181): the cap is only checked on entry */
			memcpy(c->out + c->n * c->k, seq_new, (size_t)c->k * sizeof(int32_t));
			c->n++;
		} else {
			int32_t left_new[ORC_MAX_JOINS];
			int nl = nleft;
			memcpy(left_new, left, (size_t)nleft * sizeof(int32_t));
			erase_value(left_new, &nl, join_idx);
			dfs_recursive(c, seq_new, nseq + 1, left_new, nl);
		}
	}
}

/* the "original join order first" fix-up shared by DFS and BFS (:573-608, :717-747) */
static int move_original_first(int32_t *out, int n, int k, int max_orders) {
	int orig = -1;
	for (int i = 0; i < n; i++) {
		int is_orig = 1;
		for (int j = 0; j < k; j++) {
			if (out[i * k + j] != j) {
				is_orig = 0;
				break;
			}
		}
		if (is_orig) {
			orig = i;
			break;
		}
	}
	if (orig < 0) {
		memmove(out + k, out, (size_t)n * (size_t)k * sizeof(int32_t));
		for (int j = 0; j < k; j++) {
			out[j] = j;
		}
		n++;
		if (n > max_orders) {
			n--;
		}
	} else if (orig != 0) {
		memmove(out + k, out, (size_t)orig * (size_t)k * sizeof(int32_t));
		for (int j = 0; j < k; j++) {
			out[j] = j;
		}
	}
	return n;
}

typedef struct {
	idx_t level, candidate_idx, step;
	int32_t pred[ORC_MAX_JOINS];
	int npred;
	int candidate;
} bfs_entry_t;

static int bfs_find_candidates(int k, const int32_t *pred, int npred, const uint8_t *deps, int32_t *out) {
	/* BFSEnumeration::FindJoinCandidates, :669-685 */
	int n = 0;
	for (int i = 0; i < k; i++) {
		int found = 0;
		for (int q = 0; q < npred; q++) {
			if (pred[q] == i) {
				found = 1;
			}
		}
		if (!found && can_join(pred, npred, i, k, deps)) {
			out[n++] = i;
		}
	}
	return n;
}

int orc_enumerate(int enumerator, int k, const uint8_t *deps, const idx_t *card, int max_join_orders,
                  int32_t *paths) {
	int n = 0;
	/* scratch large enough for the one-past-the-cap pushes the reference allows */
	int32_t *out = (int32_t *)calloc((size_t)(max_join_orders + 2 + k) * (size_t)k * 4, sizeof(int32_t));
	switch (enumerator) {
	case ORC_ENUM_EACH_LAST_ONCE: { /* :610-638 */
		for (int j = 0; j < k; j++) {
			out[j] = j;
		}
		n = 1;
		for (int i = 0; i < k - 1; i++) {
			int32_t gen[ORC_MAX_JOINS];
			int ng = 0;
			for (int j = 0; j < k; j++) {
				if (j == i) {
					continue;
				}
				if (!can_join(gen, ng, j, k, deps)) {
					break;
				}
				gen[ng++] = j;
			}
			if (ng == k - 1 && can_join(gen, ng, i, k, deps)) {
				gen[ng++] = i;
				memcpy(out + n * k, gen, (size_t)k * sizeof(int32_t));
				n++;
			}
		}
		break;
	}
	case ORC_ENUM_EACH_FIRST_ONCE: { /* :640-667 */
		for (int j = 0; j < k; j++) {
			out[j] = j;
		}
		n = 1;
		for (int i = 1; i < k; i++) {
			int32_t gen[ORC_MAX_JOINS];
			int ng = 0;
			if (!can_join(gen, 0, i, k, deps)) {
				continue;
			}
			gen[ng++] = i;
			for (int j = 0; j < k; j++) {
				if (j == i) {
					continue;
				}
				if (!can_join(gen, ng, j, k, deps)) {
					break;
				}
				gen[ng++] = j;
			}
			if (ng == k) {
				memcpy(out + n * k, gen, (size_t)k * sizeof(int32_t));
				n++;
			}
		}
		break;
	}
	case ORC_ENUM_DFS_MIN_CARD: { /* :558-608 */
		enum_ctx_t c = {k, max_join_orders, 0, deps, card, out};
		int32_t left[ORC_MAX_JOINS];
		for (int i = 0; i < k; i++) {
			left[i] = i;
		}
		dfs_recursive(&c, NULL, 0, left, k);
		n = move_original_first(out, c.n, k, max_join_orders);
		break;
	}
	case ORC_ENUM_BFS_MIN_CARD: { /* :687-747 */
		int cap = 4096, nq = 0;
		bfs_entry_t *q = (bfs_entry_t *)malloc((size_t)cap * sizeof(bfs_entry_t));
		int32_t first[ORC_MAX_JOINS];
		int nfirst = bfs_find_candidates(k, NULL, 0, deps, first);
		idx_t step = 0;
		int num_initial = nfirst < 4 ? nfirst : 4;
		for (int i = 0; i < num_initial; i++) {
			int next = select_min_card(first, nfirst, card);
			bfs_entry_t en;
			memset(&en, 0, sizeof(en));
			en.level = 0;
			en.candidate_idx = (idx_t)i;
			en.step = step;
			en.candidate = next;
			q[nq++] = en;
			erase_value(first, &nfirst, next);
			step++;
		}
		while (n <= max_join_orders && nq > 0) {
			/* priority_queue top(): smallest (level, candidate_idx, step) given operator< (:651-661) */
			int best = 0;
			for (int i = 1; i < nq; i++) {
				bfs_entry_t *a = &q[i], *b = &q[best];
				if (a->level < b->level || (a->level == b->level && (a->candidate_idx < b->candidate_idx ||
				                                                      (a->candidate_idx == b->candidate_idx &&
				                                                       a->step < b->step)))) {
					best = i;
				}
			}
			bfs_entry_t entry = q[best];
			q[best] = q[--nq];
			entry.pred[entry.npred++] = entry.candidate;
			int32_t cands[ORC_MAX_JOINS];
			int nc = bfs_find_candidates(k, entry.pred, entry.npred, deps, cands);
			if (entry.npred == k - 1 && nc == 1) {
				entry.pred[entry.npred++] = cands[0];
				memcpy(out + n * k, entry.pred, (size_t)k * sizeof(int32_t));
				n++;
			} else {
				int num = 4 - entry.npred;
				if (num < 1) {
					num = 1;
				}
				if (num > nc) {
					num = nc;
				}
				for (int i = 0; i < num; i++) {
					int cand = select_min_card(cands, nc, card);
					erase_value(cands, &nc, cand);
					bfs_entry_t en = entry;
					en.level = (idx_t)entry.npred;
					en.candidate_idx = (idx_t)i;
					en.step = step;
					en.candidate = cand;
					if (nq == cap) {
						cap *= 2;
						q = (bfs_entry_t *)realloc(q, (size_t)cap * sizeof(bfs_entry_t));
					}
					q[nq++] = en;
					step++;
				}
			}
		}
		free(q);
		n = move_original_first(out, n, k, max_join_orders);
		break;
	}
	default:
		free(out);
		return -1;
	}
	memcpy(paths, out, (size_t)n * (size_t)k * sizeof(int32_t));
	free(out);
	return n;
}

void orc_bindings(int k, int n_probe_cols, const int32_t *num_build_cols, const orc_join_t *joins,
                  const int32_t *paths, int n_paths, int32_t *bindings) {
	/* polar_config.cpp:152-229: relative binding (source join, relative column) is path-independent;
	 * per path it is turned into the absolute column index of that path's layout. */
	for (int p = 0; p < n_paths; p++) {
		const int32_t *path = paths + p * k;
		int32_t offsets[ORC_MAX_JOINS + 1];
		offsets[0] = n_probe_cols;
		for (int j = 0; j < k; j++) {
			int join_idx = path[j];
			for (int c = 0; c < ORC_MAX_KEYS; c++) {
				int32_t *dst = &bindings[(p * k + j) * ORC_MAX_KEYS + c];
				*dst = -1;
				if (c >= joins[join_idx].n_keys || joins[join_idx].key_src_join[c] < 0) {
					continue;
				}
				int src = joins[join_idx].key_src_join[c];
				/* the reference scans current_join_path_column_offsets.size() entries = j+1 (:198-203) */
				for (int i = 0; i < j + 1; i++) {
					if (path[i] == src) {
						*dst = offsets[i] + joins[join_idx].key_src_col[c];
					}
				}
			}
			offsets[j + 1] = offsets[j] + num_build_cols[join_idx];
		}
	}
}

/* ---- table scan with pushed-down filters (row_group.cpp:316-452, column_segment.cpp:194-475) ---- */
static int64_t orc_cell_i64(const orc_col_t *c, idx_t row) {
	const uint8_t *p = (const uint8_t *)c->data + row * (idx_t)c->width;
	switch (c->width) {
	case 1:
		return c->is_signed ? (int64_t) * (const int8_t *)p : (int64_t) * (const uint8_t *)p;
	case 2: {
		uint16_t v;
		memcpy(&v, p, 2);
		return c->is_signed ? (int64_t)(int16_t)v : (int64_t)v;
	}
	case 4: {
		uint32_t v;
		memcpy(&v, p, 4);
		return c->is_signed ? (int64_t)(int32_t)v : (int64_t)v;
	}
	default: {
		int64_t v;
		memcpy(&v, p, 8);
		return v;
	}
	}
}

static int orc_compare(const orc_col_t *c, idx_t row, int op, int64_t constant) {
	/* OP::Operation(vec[idx], *predicate) on the column's physical type (FilterSelectionSwitch :208-300);
	 * an unsigned 64-bit column compares unsigned */
	if (c->width == 8 && !c->is_signed) {
		uint64_t v;
		memcpy(&v, (const uint8_t *)c->data + row * 8, 8);
		const uint64_t k = (uint64_t)constant;
		switch (op) {
		case ORC_CMP_EQ:
			return v == k;
		case ORC_CMP_NE:
			return v != k;
		case ORC_CMP_LT:
			return v < k;
		case ORC_CMP_GT:
			return v > k;
		case ORC_CMP_LE:
			return v <= k;
		default:
			return v >= k;
		}
	}
	const int64_t v = orc_cell_i64(c, row);
	switch (op) {
	case ORC_CMP_EQ:
		return v == constant;
	case ORC_CMP_NE:
		return v != constant;
	case ORC_CMP_LT:
		return v < constant;
	case ORC_CMP_GT:
		return v > constant;
	case ORC_CMP_LE:
		return v <= constant;
	default:
		return v >= constant;
	}
}

idx_t orc_scan_filter(const orc_col_t *cols, idx_t n_rows, const orc_filter_t *filters, int n_filters,
                      idx_t vector_size, uint32_t *sel, idx_t *n_sel, idx_t *chunk_offsets) {
	idx_t n_chunks = 0, out = 0;
	uint32_t *approved = (uint32_t *)malloc(sizeof(uint32_t) * (vector_size ? vector_size : 1));
	for (idx_t current_row = 0; current_row < n_rows; current_row += vector_size) { /* row_group.cpp:323-329 */
		const idx_t max_count = n_rows - current_row < vector_size ? n_rows - current_row : vector_size;
		idx_t approved_tuple_count = max_count; /* :375 (no deletions: count == max_count) */
		for (idx_t i = 0; i < max_count; i++) {
			approved[i] = (uint32_t)i;
		}
		for (int f = 0; f < n_filters; f++) { /* :388-394: one Select per filter column, each thins `sel` */
			const orc_col_t *c = &cols[filters[f].col];
			idx_t result_count = 0;
			for (idx_t i = 0; i < approved_tuple_count; i++) { /* column_segment.cpp:198-204 */
				const idx_t idx = current_row + approved[i];
				const int valid = !(c->valid && !c->valid[idx]);
				int keep;
				if (filters[f].op == ORC_CMP_IS_NULL) {
					keep = !valid;
				} else if (filters[f].op == ORC_CMP_IS_NOT_NULL) {
					keep = valid;
				} else {
					keep = valid && orc_compare(c, idx, filters[f].op, filters[f].constant);
				}
				if (keep) {
					approved[result_count++] = approved[i];
				}
			}
			approved_tuple_count = result_count;
		}
		if (approved_tuple_count == 0) {
			continue; /* :399-416: all rows filtered out, skip this vector */
		}
		chunk_offsets[n_chunks++] = out;
		for (idx_t i = 0; i < approved_tuple_count; i++) {
			sel[out++] = (uint32_t)(current_row + approved[i]);
		}
	}
	chunk_offsets[n_chunks] = out;
	*n_sel = out;
	free(approved);
	return n_chunks;
}
