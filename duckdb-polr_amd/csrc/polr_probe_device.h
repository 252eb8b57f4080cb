// duckdb-polr_amd/csrc/polr_probe_device.h -- the per-wave probe pipeline (device code) shared by the path kernel
// (polr_probe.hip: one launch per routed round) and the pool kernel (polr_pool.hip: the whole run in one launch).
// See polr_probe.hip for the execution model.  Needs POLR_K (compiled stage count) defined by the including file.
#pragma once

// POLR_EXT = 1 builds the generic pipeline with its uncommon parts (composite keys in packed form, non-equality join
// conditions); pipelines that need neither run on the POLR_EXT = 0 build of the same kernels, whose stages carry no trace
// of them (inlined into every place a stage fetches keys or emits matches they doubled the scalar-register spills)
#ifndef POLR_EXT
#define POLR_EXT 1
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "polr_device.h"
#include "polr_mpx_device.h"

// Per-stage input queues (LDS, per wave).  Stage 1 is fed by the wide stage-0 step (up to 256 matches at
// once): capacity 63 residual + 256; deeper queues take one batch of 64 at a time: 63 + 64.
#define QCAP1 320
#define QCAPN 128
#define WIDE 4 // tuples per lane of a wide stage-0 step
static_assert(64 * WIDE <= 256, "the pinned-step expansion searches 256 prefix entries in 8 halvings");
template <int POS>
__device__ __host__ constexpr int qcap() {
	return POS == 1 ? QCAP1 : QCAPN;
}
template <int W, int POS>
__device__ __host__ constexpr int qoff() { // dword offset of the queue that feeds stage POS (POS >= 1)
	return POS <= 1 ? 0 : W * QCAP1 + (POS - 2) * W * QCAPN;
}
template <int W, int K>
__device__ __host__ constexpr int qtotal() {
	return K <= 1 ? 0 : W * QCAP1 + (K - 2) * W * QCAPN;
}
// dwords of LDS one wave owns: descriptors, queues, narrow pending areas [K][64] x 2, pinned stage-0 rows
// [256], wide pending areas [2][256] x 2
// wide pending areas: stage 0 always, the last stage only where it takes wide steps (K <= 4, W <= 4)
template <int W, int K>
__device__ __host__ constexpr int wide_pend_slots() {
	return (K <= 4 && W <= 4) ? 2 : 1;
}
template <int W, int K>
__device__ __host__ constexpr int per_wave_dwords() {
	return K * STAGE_DESC_DWORDS + qtotal<W, K>() + K * 64 * 2 + 64 * WIDE + wide_pend_slots<W, K>() * 64 * WIDE * 2;
}
#define NO_CHUNK 0xFFFFFFFFu

// diagnostic build only (-DPOLR_DIAG_STAMPS): wall-clock stamps of workgroup 0 / wave 0 at the phase
// boundaries of a self-routing launch, written to sr.stamps[iter*8 + i] (never compiled into the product)
#ifdef POLR_DIAG_STAMPS
#define STAMP(i)                                                                                                       \
	if (sr.stamps && blockIdx.x == 0 && threadIdx.x == 0) {                                                            \
		sr.stamps[(uint64_t)sr.iter * 8 + (i)] = wall_clock64();                                                       \
	}
#else
#define STAMP(i)
#endif

__device__ __forceinline__ uint32_t uni(uint32_t v) {
	return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
	uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
	uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
	return ((uint64_t)hi << 32) | lo;
}
template <class T>
__device__ __forceinline__ const T *uniptr(const T *p) {
	return (const T *)uni64((uint64_t)p);
}
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__device__ __forceinline__ uint64_t load_cell(const uint8_t *p, uint32_t width, bool sign_extend) {
	switch (width) {
	case 1: {
		uint8_t v = *p;
		return sign_extend ? (uint64_t)(int64_t)(int8_t)v : (uint64_t)v;
	}
	case 2: {
		uint16_t v = *(const uint16_t *)p;
		return sign_extend ? (uint64_t)(int64_t)(int16_t)v : (uint64_t)v;
	}
	case 4: {
		uint32_t v = *(const uint32_t *)p;
		return sign_extend ? (uint64_t)(int64_t)(int32_t)v : (uint64_t)v;
	}
	default:
		return *(const uint64_t *)p;
	}
}

template <int W>
struct Tuple {
	uint32_t s[W];
};

// per-stage descriptor pulled from LDS into wave-uniform registers
struct Stage {
	uint32_t kind, n_keys, key_width0, key_width1, key_signed;
	uint32_t xflags; // POLR_EXT builds: bit 0 = composite key in packed form, bits 8.. = number of non-equality
	                 // conditions (both rare: their descriptors are read from the stage's extension record at use)
	int32_t key_slot0, key_slot1, out_slot;
	const uint8_t *key_data0, *key_valid0, *key_data1, *key_valid1;
	const void *table;
	const uint32_t *rowids;
	uint64_t mask;
	int64_t min_value;
	uint64_t range;
	uint32_t sentinel_start, sentinel_count;
};

__device__ __forceinline__ Stage load_stage(const StageDesc *d) {
	Stage s;
	s.kind = uni(d->kind);
	s.n_keys = uni(d->n_keys);
#if POLR_EXT
	s.xflags = uni(d->packed | (d->n_preds << 8));
#else
	s.xflags = 0; // (pipelines with packed keys or conditions are launched on the POLR_EXT build of these kernels)
#endif
	s.key_width0 = uni(d->key_width[0]);
	s.key_width1 = uni(d->key_width[1]);
	s.key_signed = uni(d->key_signed);
	s.key_slot0 = (int32_t)uni((uint32_t)d->key_slot[0]);
	s.key_slot1 = (int32_t)uni((uint32_t)d->key_slot[1]);
	s.out_slot = (int32_t)uni((uint32_t)d->out_slot);
	s.key_data0 = uniptr(d->key_data[0]);
	s.key_valid0 = uniptr(d->key_valid[0]);
	s.key_data1 = uniptr(d->key_data[1]);
	s.key_valid1 = uniptr(d->key_valid[1]);
	s.table = uniptr((const uint8_t *)d->table);
	s.rowids = uniptr(d->rowids);
	s.mask = uni64(d->mask);
	s.min_value = (int64_t)uni64((uint64_t)d->min_value);
	s.range = uni64(d->range);
	s.sentinel_start = uni(d->sentinel_start);
	s.sentinel_count = uni(d->sentinel_count);
	return s;
}

template <int W>
__device__ __forceinline__ uint32_t tuple_slot(const Tuple<W> &t, int32_t slot) {
	uint32_t v = t.s[0];
#pragma unroll
	for (int q = 1; q < W; q++) {
		v = (q == slot) ? t.s[q] : v;
	}
	return v;
}

#if POLR_EXT
// composite key in packed form (KeyPack): per column (value - min) << shift; a value outside the build side's
// [min, min + range] cannot match.  Everything comes from the extension record in global memory.  A function of its own
// (like preds_hold below): inlined into every place a stage fetches keys, the uncommon paths made the POLR_EXT objects
// the slowest of the build by minutes -- the pipelines that take them can afford a call.
template <int W>
__device__ __attribute__((noinline)) bool fetch_key_packed(const StageExt *d, uint32_t n_keys, const Tuple<W> &t,
                                                           uint64_t &key) {
	bool valid = true;
	key = 0;
	for (uint32_t c = 0; c < n_keys; c++) {
		const uint32_t row = tuple_slot<W>(t, d->key_slot[c]);
		const uint8_t *kv = d->key_valid[c];
		const uint32_t w = d->key_width[c];
		const uint64_t v = load_cell(d->key_data[c] + (uint64_t)row * w, w, d->key_sx[c] != 0); // (the probe column's own type)
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
#endif

// key of this lane's tuple; false for NULL (NULL never matches: join_hashtable.cpp:170-192,
// perfect_hash_join_executor.cpp:272-277)
template <int W>
__device__ __forceinline__ bool fetch_key(const Stage &s, const StageDesc *desc, const Tuple<W> &t, bool active,
                                          uint64_t &key) {
	key = 0;
	if (!active) {
		return false;
	}
#if POLR_EXT
	if (s.xflags & 1u) {
		return fetch_key_packed<W>(uniptr(desc->ext), s.n_keys, t, key);
	}
#endif
	const bool sx = s.kind == KIND_PERFECT && s.key_signed != 0;
	const uint32_t row0 = tuple_slot<W>(t, s.key_slot0);
	bool valid = !(s.key_valid0 && !s.key_valid0[row0]);
	key = load_cell(s.key_data0 + (uint64_t)row0 * s.key_width0, s.key_width0, sx);
	if (s.n_keys > 1) {
		const uint32_t row1 = tuple_slot<W>(t, s.key_slot1);
		if (s.key_valid1 && !s.key_valid1[row1]) {
			valid = false;
		}
		key |= load_cell(s.key_data1 + (uint64_t)row1 * s.key_width1, s.key_width1, false) << 32;
	}
	return valid;
}

// ---- bucket probes ---------------------------------------------------------------------------
__device__ __forceinline__ bool lookup_perfect(const Stage &s, uint64_t key, bool valid, uint32_t &id) {
	uint64_t idx;
	bool in_range;
	if (s.key_signed) {
		const int64_t v = (int64_t)key;
		in_range = v >= s.min_value && (uint64_t)(v - s.min_value) <= s.range;
		idx = (uint64_t)(v - s.min_value);
	} else {
		in_range = key >= (uint64_t)s.min_value && key - (uint64_t)s.min_value <= s.range;
		idx = key - (uint64_t)s.min_value;
	}
	bool hit = false;
	if (valid && in_range) {
		const uint32_t *bits = (const uint32_t *)s.table;
		hit = (bits[idx >> 5] >> (idx & 31)) & 1u;
	}
	id = (uint32_t)idx;
	return hit;
}

// Linear probing, but one round trip inspects an aligned group of slots (32 bytes): a wave waits for its
// slowest lane, so what counts is the number of DEPENDENT loads of the unluckiest of 64 lanes; at load
// factor <= 0.5 a group of 4 (2) slots almost always holds the end of the probe sequence.
struct S8Probe { // one in-flight {key,row} probe
	uint64_t group;
	uint32_t first;
	uint32_t k32;
	bool searching;
	bool hit;
	uint32_t id;
};

__device__ __forceinline__ void s8_begin(const Stage &s, uint64_t key, bool valid, S8Probe &p) {
	p.k32 = (uint32_t)key;
	const uint64_t h = polr_murmurhash64((uint64_t)p.k32) & s.mask;
	p.group = h >> 2;
	p.first = (uint32_t)(h & 3);
	p.searching = valid;
	p.hit = false;
	p.id = 0;
}

__device__ __forceinline__ void s8_check(const Stage &s, S8Probe &p, const uint4 a, const uint4 b) {
	const uint32_t kk[4] = {a.x, a.z, b.x, b.z};
	const uint32_t rr[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
	for (int i = 0; i < 4; i++) {
		if (p.searching && (uint32_t)i >= p.first) {
			if (rr[i] == S8_EMPTY_ROW) {
				p.searching = false;
			} else if (kk[i] == p.k32) {
				p.hit = true;
				p.id = rr[i];
				p.searching = false;
			}
		}
	}
	p.first = 0;
	p.group = (p.group + 1) & (s.mask >> 2);
}

__device__ __forceinline__ bool lookup_s8(const Stage &s, uint64_t key, bool valid, uint32_t &id) {
	const uint4 *tab = (const uint4 *)s.table; // 2 slots {key,row} per uint4
	S8Probe p;
	s8_begin(s, key, valid, p);
	while (p.searching) {
		const uint4 a = tab[p.group * 2];
		const uint4 b = tab[p.group * 2 + 1];
		s8_check(s, p, a, b);
	}
	id = p.id;
	return p.hit;
}

struct S16Probe { // one in-flight {key64,start,count} probe
	uint64_t group;
	uint64_t key;
	uint32_t first;
	bool searching;
	uint32_t start, count;
};

__device__ __forceinline__ void s16_begin(const Stage &s, uint64_t key, bool valid, S16Probe &p) {
	p.key = key;
	p.start = 0;
	p.count = 0;
	p.searching = valid;
	if (valid && key == S16_EMPTY_KEY) {
		p.start = s.sentinel_start;
		p.count = s.sentinel_count;
		p.searching = false;
	}
	const uint64_t h = polr_murmurhash64(key) & s.mask;
	p.group = h >> 1; // 2 slots per group
	p.first = (uint32_t)(h & 1);
}

__device__ __forceinline__ void s16_check(const Stage &s, S16Probe &p, const uint4 e0, const uint4 e1) {
	if (p.first == 0) {
		const uint64_t k0 = ((uint64_t)e0.y << 32) | e0.x;
		if (k0 == S16_EMPTY_KEY) {
			p.searching = false;
		} else if (k0 == p.key) {
			p.start = e0.z;
			p.count = e0.w;
			p.searching = false;
		}
	}
	if (p.searching) {
		const uint64_t k1 = ((uint64_t)e1.y << 32) | e1.x;
		if (k1 == S16_EMPTY_KEY) {
			p.searching = false;
		} else if (k1 == p.key) {
			p.start = e1.z;
			p.count = e1.w;
			p.searching = false;
		}
	}
	p.first = 0;
	p.group = (p.group + 1) & (s.mask >> 1);
}

__device__ __forceinline__ void lookup_s16(const Stage &s, uint64_t key, bool valid, uint32_t &start,
                                           uint32_t &count) {
	const uint4 *tab = (const uint4 *)s.table;
	S16Probe p;
	s16_begin(s, key, valid, p);
	while (p.searching) {
		const uint4 e0 = tab[p.group * 2];
		const uint4 e1 = tab[p.group * 2 + 1];
		s16_check(s, p, e0, e1);
	}
	start = p.start;
	count = p.count;
}

// ---- per-wave state ----------------------------------------------------------------------------
template <int W, int K>
struct WaveCtx {
	uint32_t k;
	uint32_t lane;
	// LDS regions of this wave
	StageDesc *desc;      // [K] descriptors of the current join order
	uint32_t *q;          // (K-1) queues, slot-major: q[qoff<W,pos>() + slot*qcap<pos>() + idx]
	uint32_t *pend_start; // [K][64]
	uint32_t *pend_pref;  // [K][64] inclusive prefix of run lengths
	uint32_t *batch0;     // [256] probe rows of the pinned stage-0 step (a narrow batch uses the first 64)
	// wave-uniform scalars, statically indexed
	uint32_t qsize[K], pend_T[K], pend_cur[K], pend_base[K], cnt[K];
	const uint32_t *sel;
	uint64_t in_pos, in_end;
	uint32_t flush_token; // depends on the returned values of the counter atomics (ordering only)
	// bit p: stage p of the current join order takes wide steps (256 tuples, 4 lookups in flight per lane);
	// pend_wide bit p: its pending expansion is a pinned wide step (wide pending area)
	uint32_t wide_mask, pend_wide;
	uint32_t *wpend_start, *wpend_pref; // [2][256]: slot 0 = stage 0, slot 1 = the last stage
	// output
	DevOut out;
	bool emit;
	uint32_t cur_chunk, fill;
	bool overflow;
};

template <int W, int K>
__device__ __forceinline__ void out_write(WaveCtx<W, K> &c, const Tuple<W> &t, bool valid) {
	if (!c.emit || c.out.ids == nullptr) {
		return;
	}
	const uint64_t m = __ballot(valid);
	const uint32_t n = (uint32_t)__popcll(m);
	const uint32_t rank = lane_rank(m);
	uint32_t done = 0;
	while (done < n) {
		if (c.cur_chunk == NO_CHUNK || c.fill == c.out.chunk_capacity) {
			if (c.cur_chunk != NO_CHUNK && c.lane == 0) {
				c.out.chunk_count[c.cur_chunk] = c.fill;
			}
			uint32_t nc = 0;
			if (c.lane == 0) {
				nc = atomicAdd(&c.out.cursor[0], 1u);
			}
			nc = uni(nc);
			if (nc >= c.out.max_chunks) {
				if (c.lane == 0) {
					atomicExch(&c.out.cursor[1], 1u);
				}
				c.overflow = true;
				c.cur_chunk = NO_CHUNK;
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
				c.out.ids[(uint64_t)i * c.out.slot_stride + base] = t.s[i];
			}
		}
		c.fill += take;
		done += take;
	}
}

// the join's non-equality conditions on one (tuple, build row) pair (RowOperations::Match, row_match.cpp:59-119:
// both sides valid and `left OP right`); descriptors come from the extension record -- joins that have any are rare
template <int W>
__device__ __attribute__((noinline)) bool preds_hold(const StageExt *d, uint32_t n_preds, const Tuple<W> &t, uint32_t id) {
	bool ok = true;
	for (uint32_t c = 0; c < n_preds; c++) {
		const uint32_t row = tuple_slot<W>(t, d->pred_slot[c]);
		const uint32_t w = d->pred_width[c];
		const bool sx = d->pred_sx[c] != 0;
		const uint8_t *lv = d->pred_valid[c], *rv = d->pred_bvalid[c];
		if ((lv && !lv[row]) || (rv && !rv[id])) {
			ok = false;
		}
		if (d->pred_op[c] == POLR_PRED_STR_EQ) { // the strings behind a VARCHAR key's hash
			ok = ok && polr_str_cells_equal(d->pred_data[c] + (uint64_t)row * 16u, d->pred_bdata[c] + (uint64_t)id * 16u);
			continue;
		}
		const uint64_t l = load_cell(d->pred_data[c] + (uint64_t)row * w, w, sx);
		const uint64_t r = load_cell(d->pred_bdata[c] + (uint64_t)id * w, w, sx);
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

// push the matches of stage POS to the next stage (or to the output when POS is the last join)
template <int W, int K, int POS>
__device__ __forceinline__ void emit_tuples(WaveCtx<W, K> &c, const Stage &s, Tuple<W> t, uint32_t id, bool valid) {
#if POLR_EXT
	if (s.xflags >> 8) {
		// (inactive lanes carry arbitrary ids: evaluate on the matches only)
		valid = valid && preds_hold<W>(uniptr(c.desc[POS].ext), s.xflags >> 8, t, id);
	}
#endif
#pragma unroll
	for (int i = 1; i < W; i++) {
		t.s[i] = (i == s.out_slot) ? id : t.s[i];
	}
	const uint64_t m = __ballot(valid);
	const uint32_t n = (uint32_t)__popcll(m);
	c.cnt[POS] += n;
	if (POS + 1 >= K || POS + 1 == (int)c.k) {
		out_write(c, t, valid);
		return;
	}
	if constexpr (POS + 1 < K) {
		const uint32_t qs = c.qsize[POS + 1];
		if (valid) {
			const uint32_t idx = qs + lane_rank(m);
			uint32_t *qq = c.q + qoff<W, POS + 1>();
#pragma unroll
			for (int i = 0; i < W; i++) {
				qq[i * qcap<POS + 1>() + idx] = t.s[i];
			}
		}
		c.qsize[POS + 1] = qs + n;
	}
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(v, d, 64);
		if ((int)lane >= d) {
			v += o;
		}
	}
	return v;
}

// continue the pending expansion of stage POS: emit the next <= 64 (tuple, build row) pairs
template <int W, int K, int POS>
__device__ __forceinline__ void resume_expansion(WaveCtx<W, K> &c) {
	const Stage s = load_stage(&c.desc[POS]);
	const uint32_t T = c.pend_T[POS];
	const uint32_t cur = c.pend_cur[POS];
	const uint32_t o = cur + c.lane;
	const bool valid = o < T;
	const uint32_t *pref = c.pend_pref + POS * 64;
	uint32_t lo = 0, hi = 63; // smallest index with pref[index] > o
	if (valid) {
#pragma unroll
		for (int it = 0; it < 6; it++) {
			const uint32_t mid = (lo + hi) >> 1;
			if (pref[mid] > o) {
				hi = mid;
			} else {
				lo = mid + 1;
			}
		}
	}
	const uint32_t src = valid ? lo : 0;
	const uint32_t excl = src > 0 ? pref[src - 1] : 0;
	const uint32_t r = o - excl;
	const uint32_t st = c.pend_start[POS * 64 + src];
	Tuple<W> t;
#pragma unroll
	for (int i = 0; i < W; i++) {
		t.s[i] = 0;
	}
	if (POS == 0) {
		t.s[0] = c.batch0[src];
	} else {
		const uint32_t *qq = c.q + qoff<W, POS>();
#pragma unroll
		for (int i = 0; i < W; i++) {
			t.s[i] = qq[i * qcap<POS>() + c.pend_base[POS] + src];
		}
	}
	uint32_t id = 0;
	if (valid && (s.out_slot >= 0 || (s.xflags >> 8))) { // (the build id is needed: carried on, or read by a condition)
		id = s.rowids[st + r];
	}
	emit_tuples<W, K, POS>(c, s, t, id, valid);
	if (cur + 64 >= T) {
		c.pend_T[POS] = 0;
		c.pend_cur[POS] = 0;
	} else {
		c.pend_cur[POS] = cur + 64;
	}
}

// take one batch of <= 64 tuples into stage POS and probe
template <int W, int K, int POS>
__device__ __forceinline__ void run_stage(WaveCtx<W, K> &c) {
	const Stage s = load_stage(&c.desc[POS]);
	Tuple<W> t;
#pragma unroll
	for (int i = 0; i < W; i++) {
		t.s[i] = 0;
	}
	bool active;
	uint32_t base = 0;
	if (POS == 0) {
		const uint64_t left = c.in_end - c.in_pos;
		const uint32_t n = left < 64 ? (uint32_t)left : 64u;
		active = c.lane < n;
		if (active) {
			const uint64_t tp = c.in_pos + c.lane;
			t.s[0] = c.sel ? c.sel[tp] : (uint32_t)tp;
		}
		c.in_pos += n;
	} else {
		const uint32_t qs = c.qsize[POS];
		const uint32_t n = qs < 64 ? qs : 64u;
		base = qs - n;
		active = c.lane < n;
		if (active) {
			const uint32_t *qq = c.q + qoff<W, POS>();
#pragma unroll
			for (int i = 0; i < W; i++) {
				t.s[i] = qq[i * qcap<POS>() + base + c.lane];
			}
		}
		// popped; if the batch has to be expanded its cells stay in place: nothing pushes into this
		// queue while the stage has a pending expansion (deepest stage runs first)
		c.qsize[POS] = base;
	}
	uint64_t key;
	const bool valid = fetch_key<W>(s, &c.desc[POS], t, active, key);
	if (s.kind == KIND_PERFECT) {
		uint32_t id;
		const bool hit = lookup_perfect(s, key, valid, id);
		emit_tuples<W, K, POS>(c, s, t, id, hit);
	} else if (s.kind == KIND_S8) {
		uint32_t id;
		const bool hit = lookup_s8(s, key, valid, id);
		emit_tuples<W, K, POS>(c, s, t, id, hit);
	} else {
		uint32_t start, count;
		lookup_s16(s, key, valid, start, count);
		const bool multi = __ballot(count > 1) != 0ull;
		if (!multi) {
			uint32_t id = 0;
			if (count && (s.out_slot >= 0 || (s.xflags >> 8))) {
				id = s.rowids[start];
			}
			emit_tuples<W, K, POS>(c, s, t, id, count != 0);
		} else {
			const uint32_t pref = wave_inclusive_scan(count, c.lane);
			const uint32_t T = uni(__shfl(pref, 63, 64));
			c.pend_start[POS * 64 + c.lane] = start;
			c.pend_pref[POS * 64 + c.lane] = pref;
			if (POS == 0) {
				c.batch0[c.lane] = t.s[0];
			} else {
				c.pend_base[POS] = base;
			}
			c.pend_T[POS] = T;
			c.pend_cur[POS] = 0;
			resume_expansion<W, K, POS>(c);
		}
	}
}

// Wide step: 4 tuples per lane (256 per wave) with all loads of one kind issued back to back -- 4 rows, 4 keys,
// 4 slot-group probes in flight per lane -- so a wave pays the (sel ->) key -> bucket latency chain once per 256
// tuples instead of once per 64.  Stage 0 reads the source, a deeper stage the top of its queue (sub-batch i =
// the i-th 64 entries from the top, so whatever is not consumed stays in place at the bottom).  Allowed where
// the matches have somewhere to go: stage 0 (queue 1 holds 63 + 256) and the last stage (output chunks).
// Tables with repeated keys: if some key of the step repeats, the whole step (up to 256 tuples) is pinned --
// stage 0: its rows in batch0; deeper: its queue cells stay in place -- with start / prefix-summed run lengths
// in the wide pending area, and expanded 64 outputs at a time by resume_expansion_wide, deepest stage first.
template <int W, int K, int POS>
__device__ __forceinline__ void resume_expansion_wide(WaveCtx<W, K> &c);

template <int W, int K, int POS>
__device__ __forceinline__ void run_stage_wide(WaveCtx<W, K> &c) {
	const Stage s = load_stage(&c.desc[POS]);
	uint32_t n, qs = 0, base = 0;
	if (POS == 0) {
		const uint64_t left = c.in_end - c.in_pos;
		n = left < 64 * WIDE ? (uint32_t)left : 64u * WIDE;
	} else {
		qs = c.qsize[POS];
		n = qs < 64 * WIDE ? qs : 64u * WIDE;
		base = qs - n;
	}
	Tuple<W> t[WIDE];
	bool act[WIDE];
#pragma unroll
	for (int i = 0; i < WIDE; i++) {
#pragma unroll
		for (int q = 0; q < W; q++) {
			t[i].s[q] = 0;
		}
		if (POS == 0) {
			const uint32_t off = c.lane + 64u * i;
			act[i] = off < n;
			const uint64_t tp = c.in_pos + off;
			t[i].s[0] = act[i] ? (c.sel ? c.sel[tp] : (uint32_t)tp) : 0u;
		} else {
			const int32_t idx = (int32_t)qs - 64 * (i + 1) + (int32_t)c.lane;
			act[i] = idx >= (int32_t)base;
			if (act[i]) {
				const uint32_t *qq = c.q + qoff<W, POS>();
#pragma unroll
				for (int q = 0; q < W; q++) {
					t[i].s[q] = qq[q * qcap<POS>() + idx];
				}
			}
		}
	}
	uint64_t key[WIDE];
	bool valid[WIDE];
#pragma unroll
	for (int i = 0; i < WIDE; i++) {
		valid[i] = fetch_key<W>(s, &c.desc[POS], t[i], act[i], key[i]);
	}
	uint32_t id[WIDE];
	bool hit[WIDE];
	bool multi[WIDE];
	uint32_t start[WIDE], cnt[WIDE];
#pragma unroll
	for (int i = 0; i < WIDE; i++) {
		multi[i] = false;
		start[i] = cnt[i] = 0;
	}
	if (s.kind == KIND_PERFECT) {
#pragma unroll
		for (int i = 0; i < WIDE; i++) {
			hit[i] = lookup_perfect(s, key[i], valid[i], id[i]);
		}
	} else if (s.kind == KIND_S8) {
		const uint4 *tab = (const uint4 *)s.table;
		S8Probe p[WIDE];
		bool any = false;
#pragma unroll
		for (int i = 0; i < WIDE; i++) {
			s8_begin(s, key[i], valid[i], p[i]);
			any = any || p[i].searching;
		}
		while (any) {
			uint4 a[WIDE], b[WIDE];
#pragma unroll
			for (int i = 0; i < WIDE; i++) {
				if (p[i].searching) {
					a[i] = tab[p[i].group * 2];
					b[i] = tab[p[i].group * 2 + 1];
				}
			}
			any = false;
#pragma unroll
			for (int i = 0; i < WIDE; i++) {
				if (p[i].searching) {
					s8_check(s, p[i], a[i], b[i]);
				}
				any = any || p[i].searching;
			}
		}
#pragma unroll
		for (int i = 0; i < WIDE; i++) {
			hit[i] = p[i].hit;
			id[i] = p[i].id;
		}
	} else {
		const uint4 *tab = (const uint4 *)s.table;
		S16Probe p[WIDE];
		bool any = false;
#pragma unroll
		for (int i = 0; i < WIDE; i++) {
			s16_begin(s, key[i], valid[i], p[i]);
			any = any || p[i].searching;
		}
		while (any) {
			uint4 e0[WIDE], e1[WIDE];
#pragma unroll
			for (int i = 0; i < WIDE; i++) {
				if (p[i].searching) {
					e0[i] = tab[p[i].group * 2];
					e1[i] = tab[p[i].group * 2 + 1];
				}
			}
			any = false;
#pragma unroll
			for (int i = 0; i < WIDE; i++) {
				if (p[i].searching) {
					s16_check(s, p[i], e0[i], e1[i]);
				}
				any = any || p[i].searching;
			}
		}
		// row ids of the single matches: one more hop, all sub-batches at once
#pragma unroll
		for (int i = 0; i < WIDE; i++) {
			hit[i] = p[i].count != 0;
			start[i] = p[i].start;
			cnt[i] = p[i].count;
			multi[i] = __ballot(p[i].count > 1) != 0ull;
			id[i] = (hit[i] && (s.out_slot >= 0 || (s.xflags >> 8))) ? s.rowids[p[i].start] : 0u;
		}
	}
	bool any_multi = false;
#pragma unroll
	for (int i = 0; i < WIDE; i++) {
		any_multi = any_multi || multi[i];
	}
	if (POS == 0) {
		c.in_pos += n;
	} else {
		c.qsize[POS] = base; // popped (a pinned step's cells stay in place below the new top)
	}
	if (!any_multi) {
#pragma unroll
		for (int i = 0; i < WIDE; i++) {
			if (64u * i < n) {
				emit_tuples<W, K, POS>(c, s, t[i], id[i], hit[i]);
			}
		}
		return;
	}
	// pin the step and start its expansion
	constexpr int SLOT = POS == 0 ? 0 : 1;
	uint32_t *wst = c.wpend_start + SLOT * 64 * WIDE;
	uint32_t *wpf = c.wpend_pref + SLOT * 64 * WIDE;
	uint32_t carry = 0;
#pragma unroll
	for (int i = 0; i < WIDE; i++) {
		const uint32_t pref = carry + wave_inclusive_scan(cnt[i], c.lane);
		wst[i * 64 + c.lane] = start[i];
		wpf[i * 64 + c.lane] = pref;
		carry = uni(__shfl(pref, 63, 64));
		if (POS == 0) {
			c.batch0[i * 64 + c.lane] = t[i].s[0];
		}
	}
	if (POS != 0) {
		c.pend_base[POS] = qs;
	}
	c.pend_T[POS] = carry;
	c.pend_cur[POS] = 0;
	c.pend_wide |= 1u << POS;
	resume_expansion_wide<W, K, POS>(c);
}

// continue the pending expansion of a pinned wide step of stage POS: the next <= 256 (tuple, build row)
// pairs, 4 per lane, their row-id loads in flight together (one hop per 256 outputs instead of per 64)
template <int W, int K, int POS>
__device__ __forceinline__ void resume_expansion_wide(WaveCtx<W, K> &c) {
	const Stage s = load_stage(&c.desc[POS]);
	constexpr int SLOT = POS == 0 ? 0 : 1;
	const uint32_t *wst = c.wpend_start + SLOT * 64 * WIDE;
	const uint32_t *wpf = c.wpend_pref + SLOT * 64 * WIDE;
	const uint32_t T = c.pend_T[POS];
	const uint32_t cur = c.pend_cur[POS];
	uint32_t src[WIDE], pos[WIDE];
	bool valid[WIDE];
#pragma unroll
	for (int j = 0; j < WIDE; j++) {
		const uint32_t o = cur + 64u * j + c.lane;
		valid[j] = o < T;
		uint32_t lo = 0, hi = 64 * WIDE - 1; // smallest entry with prefix > o
		if (valid[j]) {
#pragma unroll
			for (int it = 0; it < 8; it++) {
				const uint32_t mid = (lo + hi) >> 1;
				if (wpf[mid] > o) {
					hi = mid;
				} else {
					lo = mid + 1;
				}
			}
		}
		src[j] = valid[j] ? lo : 0;
		const uint32_t excl = src[j] > 0 ? wpf[src[j] - 1] : 0;
		pos[j] = wst[src[j]] + (o - excl);
	}
	uint32_t id[WIDE];
#pragma unroll
	for (int j = 0; j < WIDE; j++) {
		id[j] = (valid[j] && (s.out_slot >= 0 || (s.xflags >> 8))) ? s.rowids[pos[j]] : 0u;
	}
#pragma unroll
	for (int j = 0; j < WIDE; j++) {
		if (cur + 64u * j < T) { // (wave-uniform)
			Tuple<W> t;
#pragma unroll
			for (int i = 0; i < W; i++) {
				t.s[i] = 0;
			}
			if (POS == 0) {
				t.s[0] = c.batch0[src[j]];
			} else if (valid[j]) {
				// entry (sub-batch i, lane l) of the step sits at queue index top - 64 (i + 1) + l
				const uint32_t idx = c.pend_base[POS] - 64u * ((src[j] >> 6) + 1) + (src[j] & 63u);
				const uint32_t *qq = c.q + qoff<W, POS>();
#pragma unroll
				for (int i = 0; i < W; i++) {
					t.s[i] = qq[i * qcap<POS>() + idx];
				}
			}
			emit_tuples<W, K, POS>(c, s, t, id[j], valid[j]);
		}
	}
	if (cur + 64u * WIDE >= T) {
		c.pend_T[POS] = 0;
		c.pend_cur[POS] = 0;
		c.pend_wide &= ~(1u << POS);
	} else {
		c.pend_cur[POS] = cur + 64u * WIDE;
	}
}

template <int W, int K, int POS>
__device__ __forceinline__ void dispatch_resume_wide(WaveCtx<W, K> &c, int pick) {
	if (pick == POS) {
		if constexpr (POS == 0 || (K <= 4 && W <= 4)) {
			resume_expansion_wide<W, K, POS>(c);
		}
		return;
	}
	if constexpr (POS + 1 < K) {
		dispatch_resume_wide<W, K, POS + 1>(c, pick);
	}
}

// wide steps exist for stage 0 and -- for pipelines of up to 4 stages carrying up to 4 ids -- for the last stage
template <int W, int K, int POS>
__device__ __forceinline__ void dispatch_wide(WaveCtx<W, K> &c, int pick) {
	if (pick == POS) {
		if constexpr (POS == 0 || (K <= 4 && W <= 4)) {
			run_stage_wide<W, K, POS>(c);
		}
		return;
	}
	if constexpr (POS + 1 < K) {
		dispatch_wide<W, K, POS + 1>(c, pick);
	}
}

template <int W, int K, int POS>
__device__ __forceinline__ void dispatch(WaveCtx<W, K> &c, int pick, bool resume) {
	if (pick == POS) {
		if (resume) {
			resume_expansion<W, K, POS>(c);
		} else {
			run_stage<W, K, POS>(c);
		}
		return;
	}
	if constexpr (POS + 1 < K) {
		dispatch<W, K, POS + 1>(c, pick, resume);
	}
}

// which stages of the join order whose descriptors start at `src` may take wide steps
template <int W, int K>
__device__ __forceinline__ uint32_t stage_wide_mask(const uint32_t *src, uint32_t k) {
	uint32_t m = uni(src[offsetof(StageDesc, unique) / 4]) != 0 ? 1u : 0u;
	if constexpr (K <= 4 && W <= 4) {
		if (k > 1 && uni(src[(k - 1) * STAGE_DESC_DWORDS + offsetof(StageDesc, unique) / 4]) != 0) {
			m |= 1u << (k - 1);
		}
	}
	return m;
}

// scheduler: run until the unit's input is consumed and no stage holds a full batch or a pending
// expansion; with `flushing` also drain partial batches, shallowest first.
template <int W, int K>
__device__ __forceinline__ void run_until_idle(WaveCtx<W, K> &c, bool flushing) {
	while (true) {
		int pick = -1;
		bool resume = false;
#pragma unroll
		for (int p = K - 1; p >= 0; p--) {
			if (pick < 0 && p < (int)c.k) {
				if (c.pend_T[p] != 0) {
					pick = p;
					resume = true;
				} else if (p > 0 && c.qsize[p] >= 64) {
					pick = p;
				}
			}
		}
		if (pick < 0) {
			if (c.in_pos < c.in_end) {
				pick = 0;
			} else if (flushing) {
#pragma unroll
				for (int p = K - 1; p >= 1; p--) {
					if (p < (int)c.k && c.qsize[p] > 0) {
						pick = p; // ends on the shallowest non-empty queue
					}
				}
			}
		}
		if (pick < 0) {
			return;
		}
		if (resume && ((c.pend_wide >> pick) & 1u)) {
			dispatch_resume_wide<W, K, 0>(c, pick);
		} else if (!resume && ((c.wide_mask >> pick) & 1u)) {
			dispatch_wide<W, K, 0>(c, pick);
		} else {
			dispatch<W, K, 0>(c, pick, resume);
		}
	}
}

template <int W, int K>
__device__ __forceinline__ void flush_counts(WaveCtx<W, K> &c, unsigned long long *counts, int64_t round) {
	if (c.lane == 0) {
		unsigned long long seen = 0;
#pragma unroll
		for (int p = 0; p < K; p++) {
			if (p < (int)c.k && c.cnt[p]) {
				// returning form: the wave cannot run past the point where `seen` is consumed before the
				// add has been performed at device scope -- that is what orders it before the arrival ticket
				seen |= atomicAdd(&counts[((uint64_t)round * POLR_NSHARD + (blockIdx.x % POLR_NSHARD)) * c.k + p],
				                  (unsigned long long)c.cnt[p]);
			}
		}
		c.flush_token = (uint32_t)(seen >> 63);
	}
#pragma unroll
	for (int p = 0; p < K; p++) {
		c.cnt[p] = 0;
	}
}

