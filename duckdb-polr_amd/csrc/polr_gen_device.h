// duckdb-polr_amd/csrc/polr_gen_device.h -- the GENERIC probe pipeline of the pool launch (device code, gfx950).
//
// What it computes is RunPath (src/parallel/polar_pipeline_executor.cpp:427-538) for one unit of routed tuples: the k
// hash joins of the unit's join order, fused -- PhysicalHashJoin::Execute (physical_hash_join.cpp:637-681),
// JoinHashTable::Probe / ScanStructure::NextInnerJoin (join_hashtable.cpp:396-565), ProbePerfectHashTable
// (perfect_hash_join_executor.cpp:177-291), RowOperations::Match for non-equality conditions (row_match.cpp:59-119) --
// for ANY pipeline: keys that are build columns of earlier joins, repeated build keys (fan-out), composite keys, row-id
// output.  (Banks of single-key unique-match joins that only count take the flat pipeline, polr_flat_device.h.)
//
// Round 1/2's version of this pipeline (polr_probe_device.h, still behind the per-round path kernel) compiled every
// stage POSITION separately: 160-316 KB of machine code per kernel against a 64 KB instruction cache, generic (FLAT)
// memory instructions throughout, 33 G tuples/s on the JOB 18a shape's best join order with the device otherwise idle
// (profiles/r03_*).  This one is built the other way round:
//
//   * ONE stage routine, the stage position a wave-uniform runtime value: per-stage scalars (queue fill, counters,
//     pending run) live in the LANES of four vector registers (lane p = stage p; v_readlane / v_writelane with a scalar
//     lane index), the stage's descriptor is read with scalar loads from the constant address space when the step
//     begins.  The whole probe side is a few KB of code.
//   * every access names its address space (global / LDS / constant): no FLAT instruction, so LDS traffic and the
//     key / table loads in flight wait on separate counters.
//   * a step takes up to 64 x GEN_F tuples, GEN_F per lane: the (sel ->) key -> slot-group chains of a lane are in
//     flight together, at every stage (round 2: stage 0 and sometimes the last one).  The library is built with
//     GEN_F = 2 (Makefile: GENF): measured against 1 and 4 on the JOB 18a / JOB-light shapes and the 113-pipeline pass,
//     4 costs registers for nothing on queues that rarely hold 256 tuples, 1 loses 10 % on table-sized sources.
//   * fan-out without per-stage pending areas.  A step looks its candidates up, prefix-sums the run lengths and
//     consumes the longest PREFIX of candidates whose outputs fit the next queue -- the rest stay where they are (in
//     the queue, or in the source) and are looked up again later; a single candidate whose run alone exceeds the room
//     becomes the stage's PENDING RUN (start, remaining: two scalars) and is emitted piecewise.  Deepest stage first,
//     like the reference's in_process_joins stack (polar_pipeline_executor.cpp:296-420): LDS use is bounded for any
//     fan-out, and the only per-wave scratch is one 2 KB area for the prefix sums of the step in progress.
//   * a counting sink never expands the last join: its run lengths are summed.
//   * MULTIPLICITIES (counting runs): a join whose build rows nobody reads downstream -- its build id is not a later
//     join's key source, it has no non-equality condition, nothing is materialised -- does not have to hand its matches
//     on one by one: the tuple goes on ONCE with its multiplicity multiplied by the run length, and every counter adds
//     multiplicities instead of ones.  Counts (per-join intermediates, COUNT(*)) are exactly those of the expanded
//     stream -- the reference's NextInnerJoin would have produced run-length copies that are indistinguishable to every
//     later join (join_hashtable.cpp:531-565) -- at none of its work; a skewed build key (JOB's movie_keyword /
//     movie_info runs of thousands of rows) no longer leaves one wave expanding millions of pairs alone.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "polr_device.h"

#ifndef POLR_EXT
#define POLR_EXT 1
#endif

#define POLR_CONST __attribute__((address_space(4)))

#ifndef GEN_F
#define GEN_F 2                 // candidates per lane and step
#endif
#define GEN_STEP (64 * GEN_F)   // candidates per step
#define GEN_SCRATCH_DWORDS (2 * GEN_STEP) // run starts, inclusive prefix of run lengths
#define GEN_NO_CHUNK 0xFFFFFFFFu

__device__ __forceinline__ uint32_t gen_uni(uint32_t v) {
	return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint32_t gen_rank(uint64_t m) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
__device__ __forceinline__ uint64_t gen_wave_sum64(uint64_t v) { // the same sum in every lane
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) {
		v += __shfl_xor(v, d, 64);
	}
	return v;
}
__device__ __forceinline__ uint32_t gen_lane_get(uint32_t v, uint32_t lane_uniform) {
	return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane_uniform);
}
__device__ __forceinline__ void gen_lane_set(uint32_t &v, uint32_t lane_uniform, uint32_t value_uniform) {
	// (v_writelane_b32 has no builtin in this compiler: a compare against the lane id and a select)
	v = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == lane_uniform) ? value_uniform : v;
}

__device__ __forceinline__ uint64_t gen_load_cell(const POLR_GLOBAL uint8_t *p, uint32_t width, bool sign_extend) {
	switch (width) {
	case 1: {
		const uint8_t v = *p;
		return sign_extend ? (uint64_t)(int64_t)(int8_t)v : (uint64_t)v;
	}
	case 2: {
		const uint16_t v = *(const POLR_GLOBAL uint16_t *)p;
		return sign_extend ? (uint64_t)(int64_t)(int16_t)v : (uint64_t)v;
	}
	case 4: {
		const uint32_t v = *(const POLR_GLOBAL uint32_t *)p;
		return sign_extend ? (uint64_t)(int64_t)(int32_t)v : (uint64_t)v;
	}
	default:
		return *(const POLR_GLOBAL uint64_t *)p;
	}
}

// the part of a stage descriptor a step keeps in scalar registers
struct GenStage {
	uint32_t kind, n_keys, key_width0, key_width1, key_signed, unique, xflags; // xflags: bit 0 packed key, bits 8.. n_preds
	int32_t key_slot0, key_slot1, out_slot;
	const POLR_GLOBAL uint8_t *key_data0, *key_valid0, *key_data1, *key_valid1;
	const POLR_GLOBAL uint32_t *table;
	const POLR_GLOBAL uint32_t *rowids;
	uint64_t mask;
	int64_t min_value;
	uint64_t range;
	uint32_t sentinel_start, sentinel_count;
	const POLR_GLOBAL StageExt *ext;
};

__device__ __forceinline__ GenStage gen_load_stage(const POLR_CONST StageDesc *d) {
	GenStage s;
	s.kind = d->kind;
	s.n_keys = d->n_keys;
	s.key_width0 = d->key_width[0];
	s.key_width1 = d->key_width[1];
	s.key_signed = d->key_signed;
	s.unique = d->unique;
#if POLR_EXT
	s.xflags = d->packed | (d->n_preds << 8);
#else
	s.xflags = 0; // (pipelines with packed keys or conditions run on the POLR_EXT build)
#endif
	s.key_slot0 = d->key_slot[0];
	s.key_slot1 = d->key_slot[1];
	s.out_slot = d->out_slot;
	s.key_data0 = (const POLR_GLOBAL uint8_t *)d->key_data[0];
	s.key_valid0 = (const POLR_GLOBAL uint8_t *)d->key_valid[0];
	s.key_data1 = (const POLR_GLOBAL uint8_t *)d->key_data[1];
	s.key_valid1 = (const POLR_GLOBAL uint8_t *)d->key_valid[1];
	s.table = (const POLR_GLOBAL uint32_t *)d->table;
	s.rowids = (const POLR_GLOBAL uint32_t *)d->rowids;
	s.mask = d->mask;
	s.min_value = d->min_value;
	s.range = d->range;
	s.sentinel_start = d->sentinel_start;
	s.sentinel_count = d->sentinel_count;
	s.ext = (const POLR_GLOBAL StageExt *)d->ext;
	return s;
}

template <int W>
struct GenTuple {
	uint32_t s[W];
};

template <int W>
__device__ __forceinline__ uint32_t gen_slot(const GenTuple<W> &t, int32_t slot) {
	uint32_t v = t.s[0];
#pragma unroll
	for (int q = 1; q < W; q++) {
		v = (q == slot) ? t.s[q] : v;
	}
	return v;
}

template <int W>
struct GenCtx {
	uint32_t k, lane, qcap, full; // qcap: entries per queue; full = qcap / 2: a deeper stage runs when it holds this many
	POLR_LDS uint32_t *q;         // queue that FEEDS stage p (p >= 1): q + (p - 1) * W * qcap, slot-major [W][qcap]
	POLR_LDS uint32_t *scratch;   // [GEN_SCRATCH_DWORDS]
	const POLR_GLOBAL uint32_t *sel;
	const POLR_CONST StageDesc *stages; // the k descriptors of the current join order
	// lane p: state of stage p
	uint32_t v_qsize;  // entries waiting in the queue that feeds stage p
	uint32_t v_cnt_lo, v_cnt_hi; // tuples stage p has produced in this unit (64 bits: multiplicities)
	uint32_t v_gstart; // pending run of stage p: next position in rowids[] ...
	uint32_t v_grem;   // ... and how many build rows of it are still to be emitted (0: none pending)
	uint64_t in_pos, in_end;
	bool mult; // tuple slot W - 1 carries the tuple's multiplicity (counting runs of pipelines with foldable joins)
	// output (row ids)
	DevOut out;
	bool emit, overflow;
	uint32_t cur_chunk, fill;
};

// ---- keys --------------------------------------------------------------------------------------------------------------
#if POLR_EXT
// composite key in packed form (KeyPack): per column (value - min) << shift; a value outside the build side's
// [min, min + range] cannot match
template <int W>
__device__ __attribute__((noinline)) bool gen_fetch_key_packed(const POLR_GLOBAL StageExt *d, uint32_t n_keys, const GenTuple<W> &t,
                                                               uint64_t &key) {
	bool valid = true;
	key = 0;
	for (uint32_t c = 0; c < n_keys; c++) {
		const uint32_t row = gen_slot<W>(t, d->key_slot[c]);
		const POLR_GLOBAL uint8_t *kv = (const POLR_GLOBAL uint8_t *)d->key_valid[c];
		const uint32_t w = d->key_width[c];
		// (by VALUE: the probe column's own width and signedness -- a CAST'ed key; a value outside the build side's range
		// cannot match, whichever type could or could not hold it)
		const uint64_t v = gen_load_cell((const POLR_GLOBAL uint8_t *)d->key_data[c] + (uint64_t)row * w, w, d->key_sx[c] != 0);
		uint64_t off = v - (uint64_t)d->pack.min[c];
		if (kv && !kv[row]) {
			if ((d->pack.null_eq >> c) & 1u) {
				off = d->pack.range[c] + 1u; // IS NOT DISTINCT FROM: NULL is a key value of its own
			} else {
				valid = false;
			}
		} else if (off > d->pack.range[c]) {
			valid = false;
		}
		key |= off << d->pack.shift[c];
	}
	if (!valid) {
		key = 0;
	}
	return valid;
}

// the join's non-equality conditions on one (tuple, build row) pair: both sides valid and `left OP right`
template <int W>
__device__ __attribute__((noinline)) bool gen_preds_hold(const POLR_GLOBAL StageExt *d, uint32_t n_preds, const GenTuple<W> &t,
                                                         uint32_t id) {
	bool ok = true;
	for (uint32_t c = 0; c < n_preds; c++) {
		const uint32_t row = gen_slot<W>(t, d->pred_slot[c]);
		const uint32_t w = d->pred_width[c];
		const bool sx = d->pred_sx[c] != 0;
		const POLR_GLOBAL uint8_t *lv = (const POLR_GLOBAL uint8_t *)d->pred_valid[c];
		const POLR_GLOBAL uint8_t *rv = (const POLR_GLOBAL uint8_t *)d->pred_bvalid[c];
		if ((lv && !lv[row]) || (rv && !rv[id])) {
			ok = false;
		}
		if (d->pred_op[c] == POLR_PRED_STR_EQ) { // the strings behind a VARCHAR key's hash
			ok = ok && polr_str_cells_equal(d->pred_data[c] + (uint64_t)row * 16u, d->pred_bdata[c] + (uint64_t)id * 16u);
			continue;
		}
		const uint64_t l = gen_load_cell((const POLR_GLOBAL uint8_t *)d->pred_data[c] + (uint64_t)row * w, w, sx);
		const uint64_t r = gen_load_cell((const POLR_GLOBAL uint8_t *)d->pred_bdata[c] + (uint64_t)id * w, w, sx);
		bool h;
		if (w == 8 && !sx) {
			switch (d->pred_op[c]) {
			case 0: h = l == r; break; // (POLR_CMP_EQ: the verifying comparison behind a hashed composite key)
			case 1: h = l != r; break;
			case 2: h = l < r; break;
			case 3: h = l > r; break;
			case 4: h = l <= r; break;
			default: h = l >= r; break;
			}
		} else {
			const int64_t a = (int64_t)l, b = (int64_t)r; // (narrow unsigned values are zero-extended: same order)
			switch (d->pred_op[c]) {
			case 0: h = a == b; break;
			case 1: h = a != b; break;
			case 2: h = a < b; break;
			case 3: h = a > b; break;
			case 4: h = a <= b; break;
			default: h = a >= b; break;
			}
		}
		ok = ok && h;
	}
	return ok;
}
#endif

// key of one candidate; false for NULL (NULL never matches: join_hashtable.cpp:170-192,
// perfect_hash_join_executor.cpp:272-277) and for inactive lanes
template <int W>
__device__ __forceinline__ bool gen_fetch_key(const GenStage &s, const GenTuple<W> &t, bool active, uint64_t &key) {
	key = 0;
	if (!active) {
		return false;
	}
#if POLR_EXT
	if (s.xflags & 1u) {
		return gen_fetch_key_packed<W>(s.ext, s.n_keys, t, key);
	}
#endif
	const bool sx = s.kind == KIND_PERFECT && s.key_signed != 0;
	const uint32_t row0 = gen_slot<W>(t, s.key_slot0);
	bool valid = !(s.key_valid0 && !s.key_valid0[row0]);
	if (s.key_width0 == 4) { // (the common case: one aligned dword per key)
		const uint32_t v = ((const POLR_GLOBAL uint32_t *)s.key_data0)[row0];
		key = sx ? (uint64_t)(int64_t)(int32_t)v : (uint64_t)v;
	} else {
		key = gen_load_cell(s.key_data0 + (uint64_t)row0 * s.key_width0, s.key_width0, sx);
	}
	if (s.n_keys > 1) {
		const uint32_t row1 = gen_slot<W>(t, s.key_slot1);
		if (s.key_valid1 && !s.key_valid1[row1]) {
			valid = false;
		}
		key |= gen_load_cell(s.key_data1 + (uint64_t)row1 * s.key_width1, s.key_width1, false) << 32;
	}
	return valid;
}

// ---- index lookups, GEN_F per lane in flight ---------------------------------------------------------------------------
// every lookup yields (start, count): perfect / unique-key tables count <= 1 with start = the build id itself;
// repeated-key tables the run [start, start + count) of rowids[]
__device__ __forceinline__ void gen_lookup_perfect(const GenStage &s, const uint64_t (&key)[GEN_F], const bool (&valid)[GEN_F],
                                                   uint32_t (&start)[GEN_F], uint32_t (&count)[GEN_F]) {
	uint32_t w[GEN_F];
	uint64_t idx[GEN_F];
#pragma unroll
	for (int i = 0; i < GEN_F; i++) {
		bool in_range;
		if (s.key_signed) {
			const int64_t v = (int64_t)key[i];
			in_range = v >= s.min_value && (uint64_t)(v - s.min_value) <= s.range;
			idx[i] = (uint64_t)(v - s.min_value);
		} else {
			in_range = key[i] >= (uint64_t)s.min_value && key[i] - (uint64_t)s.min_value <= s.range;
			idx[i] = key[i] - (uint64_t)s.min_value;
		}
		w[i] = 0;
		if (valid[i] && in_range) {
			w[i] = s.table[idx[i] >> 5];
		}
	}
#pragma unroll
	for (int i = 0; i < GEN_F; i++) {
		count[i] = (w[i] >> (idx[i] & 31)) & 1u;
		start[i] = (uint32_t)idx[i];
	}
}

// unique 32-bit keys: open addressing, linear probing, one aligned 32-byte group of 4 {key,row} slots per round trip
__device__ __forceinline__ void gen_lookup_s8(const GenStage &s, const uint64_t (&key)[GEN_F], const bool (&valid)[GEN_F],
                                              uint32_t (&start)[GEN_F], uint32_t (&count)[GEN_F]) {
	uint32_t group[GEN_F], first[GEN_F];
	bool searching[GEN_F];
	bool any = false;
	const uint32_t gmask = (uint32_t)(s.mask >> 2);
#pragma unroll
	for (int i = 0; i < GEN_F; i++) {
		const uint64_t h = polr_murmurhash64((uint64_t)(uint32_t)key[i]) & s.mask;
		group[i] = (uint32_t)(h >> 2);
		first[i] = (uint32_t)(h & 3u);
		searching[i] = valid[i];
		start[i] = 0;
		count[i] = 0;
		any = any || searching[i];
	}
	while (__ballot(any) != 0ull) {
		uint4 a[GEN_F], b[GEN_F];
#pragma unroll
		for (int i = 0; i < GEN_F; i++) {
			if (searching[i]) {
				a[i] = load_global_x4(s.table + (uint64_t)group[i] * 8);
				b[i] = load_global_x4(s.table + (uint64_t)group[i] * 8 + 4);
			}
		}
		any = false;
#pragma unroll
		for (int i = 0; i < GEN_F; i++) {
			if (searching[i]) {
				const uint32_t kk[4] = {a[i].x, a[i].z, b[i].x, b[i].z};
				const uint32_t rr[4] = {a[i].y, a[i].w, b[i].y, b[i].w};
#pragma unroll
				for (int j = 0; j < 4; j++) {
					if (searching[i] && (uint32_t)j >= first[i]) {
						if (rr[j] == S8_EMPTY_ROW) {
							searching[i] = false;
						} else if (kk[j] == (uint32_t)key[i]) {
							count[i] = 1;
							start[i] = rr[j];
							searching[i] = false;
						}
					}
				}
				first[i] = 0;
				group[i] = (group[i] + 1) & gmask;
			}
			any = any || searching[i];
		}
	}
}

// repeated / 64-bit / composite keys: 16-byte {key64, start, count} slots, two per group
__device__ __forceinline__ void gen_lookup_s16(const GenStage &s, const uint64_t (&key)[GEN_F], const bool (&valid)[GEN_F],
                                               uint32_t (&start)[GEN_F], uint32_t (&count)[GEN_F]) {
	uint64_t group[GEN_F];
	uint32_t first[GEN_F];
	bool searching[GEN_F];
	bool any = false;
	const uint64_t gmask = s.mask >> 1;
#pragma unroll
	for (int i = 0; i < GEN_F; i++) {
		start[i] = 0;
		count[i] = 0;
		searching[i] = valid[i];
		if (valid[i] && key[i] == S16_EMPTY_KEY) {
			start[i] = s.sentinel_start;
			count[i] = s.sentinel_count;
			searching[i] = false;
		}
		const uint64_t h = polr_murmurhash64(key[i]) & s.mask;
		group[i] = h >> 1;
		first[i] = (uint32_t)(h & 1u);
		any = any || searching[i];
	}
	while (__ballot(any) != 0ull) {
		uint4 e0[GEN_F], e1[GEN_F];
#pragma unroll
		for (int i = 0; i < GEN_F; i++) {
			if (searching[i]) {
				e0[i] = load_global_x4(s.table + group[i] * 8);
				e1[i] = load_global_x4(s.table + group[i] * 8 + 4);
			}
		}
		any = false;
#pragma unroll
		for (int i = 0; i < GEN_F; i++) {
			if (searching[i]) {
				if (first[i] == 0) {
					const uint64_t k0 = ((uint64_t)e0[i].y << 32) | e0[i].x;
					if (k0 == S16_EMPTY_KEY) {
						searching[i] = false;
					} else if (k0 == key[i]) {
						start[i] = e0[i].z;
						count[i] = e0[i].w;
						searching[i] = false;
					}
				}
				if (searching[i]) {
					const uint64_t k1 = ((uint64_t)e1[i].y << 32) | e1[i].x;
					if (k1 == S16_EMPTY_KEY) {
						searching[i] = false;
					} else if (k1 == key[i]) {
						start[i] = e1[i].z;
						count[i] = e1[i].w;
						searching[i] = false;
					}
				}
				first[i] = 0;
				group[i] = (group[i] + 1) & gmask;
			}
			any = any || searching[i];
		}
	}
}

// ---- output --------------------------------------------------------------------------------------------------------------
// the final tuples of the wave's current unit go to its current output chunk (a DataChunk stream of row ids)
template <int W>
__device__ __forceinline__ void gen_out_write(GenCtx<W> &c, const GenTuple<W> &t, bool valid) {
	if (!c.emit || c.out.ids == nullptr) {
		return;
	}
	const uint64_t m = __ballot(valid);
	const uint32_t n = (uint32_t)__popcll(m);
	const uint32_t rank = gen_rank(m);
	POLR_GLOBAL uint32_t *ids = as_global(c.out.ids);
	POLR_GLOBAL uint32_t *chunk_count = as_global(c.out.chunk_count);
	uint32_t done = 0;
	while (done < n) {
		if (c.cur_chunk == GEN_NO_CHUNK || c.fill == c.out.chunk_capacity) {
			if (c.cur_chunk != GEN_NO_CHUNK && c.lane == 0) {
				chunk_count[c.cur_chunk] = c.fill;
			}
			uint32_t nc = 0;
			if (c.lane == 0) {
				nc = atomicAdd(&c.out.cursor[0], 1u);
			}
			nc = gen_uni(nc);
			if (nc >= c.out.max_chunks) {
				if (c.lane == 0) {
					atomicExch(&c.out.cursor[1], 1u);
				}
				c.overflow = true;
				c.cur_chunk = GEN_NO_CHUNK;
				c.emit = false;
				return;
			}
			c.cur_chunk = nc;
			c.fill = 0;
		}
		const uint32_t room = c.out.chunk_capacity - c.fill;
		const uint32_t take = (n - done) < room ? (n - done) : room;
		if (valid && rank >= done && rank < done + take) {
			const uint64_t base = (uint64_t)c.cur_chunk * c.out.chunk_capacity + c.fill + (rank - done);
#pragma unroll
			for (int i = 0; i < W; i++) {
				ids[(uint64_t)i * c.out.slot_stride + base] = t.s[i];
			}
		}
		c.fill += take;
		done += take;
	}
}

// one sub-batch of (tuple, build id) pairs leaves stage `pos`: conditions, counter, next queue or output.
// qs_next: fill of the next queue (updated); returns the number of tuples that passed (the sum of their multiplicities)
template <int W>
__device__ __forceinline__ uint64_t gen_emit(GenCtx<W> &c, const GenStage &s, uint32_t pos, bool last, GenTuple<W> t, uint32_t id,
                                             bool valid, uint32_t &qs_next) {
#if POLR_EXT
	if (s.xflags >> 8) {
		// (inactive lanes carry arbitrary ids: evaluate on the pairs only)
		valid = valid && gen_preds_hold<W>(s.ext, s.xflags >> 8, t, id);
	}
#endif
#pragma unroll
	for (int i = 1; i < W; i++) {
		t.s[i] = (i == s.out_slot) ? id : t.s[i];
	}
	const uint64_t m = __ballot(valid);
	const uint32_t n = (uint32_t)__popcll(m);
	if (last) {
		gen_out_write<W>(c, t, valid);
	} else {
		if (valid) {
			POLR_LDS uint32_t *qq = c.q + (size_t)pos * W * c.qcap; // (the queue that feeds stage pos + 1)
			const uint32_t idx = qs_next + gen_rank(m);
#pragma unroll
			for (int i = 0; i < W; i++) {
				qq[i * c.qcap + idx] = t.s[i];
			}
		}
		qs_next += n;
	}
	if (c.mult) {
		return gen_wave_sum64(valid ? (uint64_t)t.s[W - 1] : 0ull);
	}
	return n;
}

__device__ __forceinline__ uint32_t gen_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(v, d, 64);
		if ((int)lane >= d) {
			v += o;
		}
	}
	return v;
}

// candidate ci of the step that stage `pos` is working on: the tuple at the source position / queue entry it stands for
// (candidates are numbered in the order they are consumed: the source front to back, a queue top to bottom)
template <int W>
__device__ __forceinline__ GenTuple<W> gen_candidate(const GenCtx<W> &c, uint32_t pos, uint32_t qs, uint32_t ci, bool active) {
	GenTuple<W> t;
#pragma unroll
	for (int q = 0; q < W; q++) {
		t.s[q] = 0;
	}
	if (active) {
		if (pos == 0) {
			const uint64_t tp = c.in_pos + ci;
			t.s[0] = c.sel ? c.sel[tp] : (uint32_t)tp;
			if (c.mult) {
				t.s[W - 1] = 1u;
			}
		} else {
			const POLR_LDS uint32_t *qq = c.q + (size_t)(pos - 1) * W * c.qcap;
			const uint32_t qi = qs - 1u - ci;
#pragma unroll
			for (int q = 0; q < W; q++) {
				t.s[q] = qq[q * c.qcap + qi];
			}
		}
	}
	return t;
}

// ---- one step of stage `pos` (wave-uniform) -----------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void gen_step(GenCtx<W> &c, const uint32_t pos) {
	const GenStage s = gen_load_stage(c.stages + pos);
	const bool last = pos + 1 == c.k;
	const uint32_t qs = pos ? gen_lane_get(c.v_qsize, pos) : 0u;
	uint32_t qs_next = last ? 0u : gen_lane_get(c.v_qsize, pos + 1);
	const uint32_t room = last ? 0xFFFFFFFFu : c.qcap - qs_next;
	const bool need_id = s.out_slot >= 0 || (s.xflags >> 8) != 0;
	uint64_t produced = 0;
	const uint32_t grem = gen_lane_get(c.v_grem, pos);
	if (grem) {
		// the pending run of this stage: candidate 0 (still at the source front / the queue top) x its next build rows
		const uint32_t gstart = gen_lane_get(c.v_gstart, pos);
		const uint32_t n_emit = grem < room ? grem : room;
		const GenTuple<W> t0 = gen_candidate<W>(c, pos, qs, 0u, true);
		for (uint32_t o0 = 0; o0 < n_emit; o0 += GEN_STEP) {
			uint32_t id[GEN_F];
#pragma unroll
			for (int j = 0; j < GEN_F; j++) {
				const uint32_t o = o0 + 64u * j + c.lane;
				id[j] = (need_id && o < n_emit) ? s.rowids[gstart + o] : 0u;
			}
#pragma unroll
			for (int j = 0; j < GEN_F; j++) {
				if (o0 + 64u * j < n_emit) {
					produced += gen_emit<W>(c, s, pos, last, t0, id[j], o0 + 64u * j + c.lane < n_emit, qs_next);
				}
			}
		}
		gen_lane_set(c.v_gstart, pos, gstart + n_emit);
		gen_lane_set(c.v_grem, pos, grem - n_emit);
		if (grem == n_emit) { // the candidate is used up
			if (pos == 0) {
				c.in_pos += 1;
			} else {
				gen_lane_set(c.v_qsize, pos, qs - 1u);
			}
		}
	} else {
		uint32_t n;
		if (pos == 0) {
			const uint64_t left = c.in_end - c.in_pos;
			n = left < GEN_STEP ? (uint32_t)left : (uint32_t)GEN_STEP;
		} else {
			n = qs < GEN_STEP ? qs : (uint32_t)GEN_STEP;
		}
		if (s.unique == 1 && n > room) {
			n = room; // (at most one output per candidate: take what the next queue has room for)
		}
		GenTuple<W> t[GEN_F];
		bool act[GEN_F], valid[GEN_F];
		uint64_t key[GEN_F];
		uint32_t start[GEN_F], count[GEN_F];
#pragma unroll
		for (int i = 0; i < GEN_F; i++) {
			const uint32_t ci = 64u * i + c.lane;
			act[i] = ci < n;
			t[i] = gen_candidate<W>(c, pos, qs, ci, act[i]);
		}
#pragma unroll
		for (int i = 0; i < GEN_F; i++) {
			valid[i] = gen_fetch_key<W>(s, t[i], act[i], key[i]);
		}
		if (s.kind == KIND_PERFECT) {
			gen_lookup_perfect(s, key, valid, start, count);
		} else if (s.kind == KIND_S8) {
			gen_lookup_s8(s, key, valid, start, count);
		} else {
			gen_lookup_s16(s, key, valid, start, count);
		}
		bool multi = false;
		uint32_t hits = 0;
#pragma unroll
		for (int i = 0; i < GEN_F; i++) {
			multi = multi || count[i] > 1u;
			hits += (uint32_t)__popcll(__ballot(count[i] != 0u));
		}
		multi = __ballot(multi) != 0ull;
		if (multi && c.mult && !need_id) {
			// nobody downstream reads this join's build rows: a run of r rows multiplies the tuple's multiplicity by r
			// instead of making r copies (unless a product leaves 32 bits: then this step expands, copies keep theirs)
			bool wide = false;
#pragma unroll
			for (int i = 0; i < GEN_F; i++) {
				wide = wide || ((uint64_t)t[i].s[W - 1] * (uint64_t)count[i]) > 0xFFFFFFFFull;
			}
			if (__ballot(wide) == 0ull) {
#pragma unroll
				for (int i = 0; i < GEN_F; i++) {
					if (count[i]) {
						t[i].s[W - 1] *= count[i];
						count[i] = 1u;
					}
				}
				multi = false;
			}
		}
		uint32_t consumed = n;
		if (!multi) {
			// at most one tuple goes on per candidate (one build row, or a run folded into the multiplicity): compact and
			// push, straight from the registers.  More matches than the next queue takes: only the longest prefix of
			// candidates whose matches fit is consumed, the others stay where they are
			bool keep[GEN_F];
#pragma unroll
			for (int i = 0; i < GEN_F; i++) {
				keep[i] = act[i];
			}
			if (hits > room) {
				uint32_t carry = 0;
				consumed = 0;
#pragma unroll
				for (int i = 0; i < GEN_F; i++) {
					const uint32_t incl = carry + gen_inclusive_scan(count[i] != 0u ? 1u : 0u, c.lane);
					carry = gen_lane_get(incl, 63);
					keep[i] = act[i] && incl <= room;
					consumed += (uint32_t)__popcll(__ballot(keep[i]));
				}
			}
			uint32_t id[GEN_F];
#pragma unroll
			for (int i = 0; i < GEN_F; i++) {
				id[i] = start[i];
				if (s.kind == KIND_S16) {
					id[i] = (count[i] && keep[i] && need_id) ? s.rowids[start[i]] : 0u;
				}
			}
#pragma unroll
			for (int i = 0; i < GEN_F; i++) {
				if (64u * i < n) {
					produced += gen_emit<W>(c, s, pos, last, t[i], id[i], count[i] != 0u && keep[i], qs_next);
				}
			}
		} else {
			// runs of build rows whose rows are needed one by one (no fold happened in this step: the tuples are read
			// back from where they wait): inclusive prefix of the run lengths over the candidates, in consumption order
			POLR_LDS uint32_t *sc_start = c.scratch;
			POLR_LDS uint32_t *sc_pref = c.scratch + GEN_STEP;
			uint32_t carry = 0;
			uint32_t incl[GEN_F];
#pragma unroll
			for (int i = 0; i < GEN_F; i++) {
				incl[i] = carry + gen_inclusive_scan(count[i], c.lane);
				sc_start[64 * i + c.lane] = start[i];
				sc_pref[64 * i + c.lane] = incl[i];
				carry = gen_lane_get(incl[i], 63);
			}
			const uint32_t total = carry;
			if (last && !c.emit && (s.xflags >> 8) == 0) {
				// a counting sink: the run lengths are all it needs
				if (c.mult) {
					uint64_t w = 0;
#pragma unroll
					for (int i = 0; i < GEN_F; i++) {
						w += (uint64_t)t[i].s[W - 1] * (uint64_t)count[i];
					}
					produced = gen_wave_sum64(w);
				} else {
					produced = total;
				}
			} else {
				uint32_t m = n, total_m = total;
				if (total > room) {
					m = 0;
#pragma unroll
					for (int i = 0; i < GEN_F; i++) {
						m += (uint32_t)__popcll(__ballot(act[i] && incl[i] <= room));
					}
					total_m = m ? sc_pref[m - 1u] : 0u;
				}
				consumed = m;
				if (m == 0) {
					// candidate 0's run alone exceeds the room: it becomes the stage's pending run (it stays where it is)
					gen_lane_set(c.v_gstart, pos, gen_lane_get(start[0], 0));
					gen_lane_set(c.v_grem, pos, gen_lane_get(count[0], 0));
				}
				for (uint32_t o0 = 0; o0 < total_m; o0 += GEN_STEP) {
					uint32_t src[GEN_F], rpos[GEN_F], id[GEN_F];
					bool ov[GEN_F];
#pragma unroll
					for (int j = 0; j < GEN_F; j++) {
						const uint32_t o = o0 + 64u * j + c.lane;
						ov[j] = o < total_m;
						uint32_t lo = 0, hi = GEN_STEP - 1; // smallest candidate whose inclusive prefix exceeds o
						if (ov[j]) {
#pragma unroll
							for (int it = 0; it < 8; it++) {
								const uint32_t mid = (lo + hi) >> 1;
								if (sc_pref[mid] > o) {
									hi = mid;
								} else {
									lo = mid + 1;
								}
							}
						}
						src[j] = ov[j] ? lo : 0u;
						const uint32_t excl = src[j] > 0 ? sc_pref[src[j] - 1u] : 0u;
						rpos[j] = sc_start[src[j]] + (o - excl);
					}
#pragma unroll
					for (int j = 0; j < GEN_F; j++) {
						id[j] = rpos[j];
						if (s.kind == KIND_S16) {
							id[j] = (ov[j] && need_id) ? s.rowids[rpos[j]] : 0u;
						}
					}
#pragma unroll
					for (int j = 0; j < GEN_F; j++) {
						if (o0 + 64u * j < total_m) {
							const GenTuple<W> tj = gen_candidate<W>(c, pos, qs, src[j], ov[j]);
							produced += gen_emit<W>(c, s, pos, last, tj, id[j], ov[j], qs_next);
						}
					}
				}
			}
		}
		if (pos == 0) {
			c.in_pos += consumed;
		} else {
			gen_lane_set(c.v_qsize, pos, qs - consumed);
		}
	}
	if (!last) {
		gen_lane_set(c.v_qsize, pos + 1, qs_next);
	}
	const uint64_t so_far = (((uint64_t)gen_lane_get(c.v_cnt_hi, pos) << 32) | gen_lane_get(c.v_cnt_lo, pos)) + produced;
	gen_lane_set(c.v_cnt_lo, pos, (uint32_t)so_far);
	gen_lane_set(c.v_cnt_hi, pos, (uint32_t)(so_far >> 32));
}

// ---- scheduler: one unit [in_pos, in_end), until nothing is left anywhere (a unit leaves nothing behind: its counters
// are final when it arrives) -------------------------------------------------------------------------------------------
// share(): called after every `share_after` steps spent on this unit (work sharing: polr_poolg.hip) -- it may take work
// out of c (the back of the source range, part of a pending run, the bottom entries of a queue) between two steps
template <int W, class ShareFn>
__device__ __forceinline__ void gen_run_unit(GenCtx<W> &c, uint32_t share_after, ShareFn &&share) {
	uint32_t steps = 0;
	while (true) {
		// deepest stage (>= 1) that has a pending run or a full step waiting
		const bool stage_lane = c.lane >= 1u && c.lane < c.k;
		const uint64_t ready = __ballot(stage_lane && (c.v_grem != 0u || c.v_qsize >= c.full));
		uint32_t pick;
		if (ready) {
			pick = 63u - (uint32_t)__clzll(ready);
		} else if (c.in_pos < c.in_end || gen_lane_get(c.v_grem, 0) != 0u) {
			pick = 0;
		} else {
			// the source is used up: drain, shallowest queue first (what it pushes merges with what waits below)
			const uint64_t waiting = __ballot(stage_lane && c.v_qsize != 0u);
			if (!waiting) {
				return;
			}
			pick = (uint32_t)__builtin_ctzll(waiting);
		}
		gen_step<W>(c, gen_uni(pick));
		if (++steps >= share_after) {
			steps = 0;
			share();
		}
	}
}
