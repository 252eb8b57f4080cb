// duckdb-polr_amd/csrc/polr_flat_device.h -- the probe pipeline for FLAT join banks (device code, gfx950).
//
// A pipeline is flat when every multiplexed join (a) is keyed by ONE 4-byte column of the probe table and (b) has
// at most one build row per key (KIND_PERFECT bit table or KIND_S8 unique-key hash table) -- the star joins of
// SSB / SSB-skew and the FK -> PK joins of JOB-light -- and only counters leave the pipeline (COUNT(*) sink, or
// the exploration / ALTERNATE rounds of any sink).  Then a tuple between two joins is just its position in the
// source, the reference's RunPath (src/parallel/polar_pipeline_executor.cpp:427-538) degenerates to "AND the k
// membership tests in path order, count the survivors of every prefix" (:486-487), and the generic pipeline's
// descriptors, tuple slots and expansion machinery (polr_probe_device.h) are dead weight.  What this version does:
//
//   * stage 0 streams its key column with 16-byte loads (8 tuples per lane, 512 per wave step), the next step's
//     keys requested one step ahead;
//   * bit tables small enough live in LDS for the whole run (copied once per workgroup): a lookup there is free
//     next to the key stream (measured: 4 LDS lookups per tuple at the 5.4 TB/s streaming rate, against
//     190-350 G lookups/s for an L2-resident table, tools/micro/gather_bench.hip);
//   * between stages the survivors wait in per-wave LDS queues (what CacheJoinChunk does for 1024-row chunks,
//     polar_pipeline_executor.cpp:166-195) as 16-bit positions inside the unit (a unit is at most 65 536 tuples);
//   * the deeper stages run as ONE SWEEP: when some deeper stage has a full step, up to 512 (256 beyond four joins)
//     entries are popped from
//     EVERY deeper queue at once, all their key gathers are issued together (8 / 4 per lane and stage), then all their
//     table lookups, then the survivors are pushed on, deepest stage first.  A wave is bound by dependent memory
//     round trips (measured: 67 % of its cycles waiting with one stage at a time), and a sweep pays two of them for
//     all stages instead of two per stage;
//   * the k stage descriptors of the current join order sit in scalar registers (loaded when the order changes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "polr_device.h"

#define FLAT_STEP0 512       // tuples per stage-0 step
#define FLAT_UNIT_MAX 65536u // queue entries are 16-bit positions inside the unit

// entries a sweep takes from each deeper queue: 8 per lane for pipelines of up to 4 joins, 4 per lane beyond (a sweep
// holds position + key + table word of every deeper stage in registers)
template <int K>
__device__ __host__ constexpr int flat_sweep_f() {
	return K <= 2 ? 8 : 4;
}
template <int K>
__device__ __host__ constexpr int flat_qcap() { // a queue holds < one sweep when a step pushes <= 512 more
	return 64 * flat_sweep_f<K>() - 1 + 512;
}
template <int K>
__device__ __host__ constexpr int flat_qoff(int pos) { // uint16 offset of the queue that FEEDS stage pos (pos >= 1)
	return (pos - 1) * (flat_qcap<K>() + 1);
}
template <int K>
__device__ __host__ constexpr int flat_per_wave_dwords() {
	return K <= 1 ? 0 : ((K - 1) * (flat_qcap<K>() + 1) * 2 + 3) / 4;
}

// scalar view of one stage (SGPRs; loaded from the StageDesc array in global memory through the scalar cache)
struct FlatStage {
	const POLR_GLOBAL uint32_t *keys;
	const POLR_GLOBAL uint8_t *valid;
	const POLR_GLOBAL uint32_t *table; // bit words or {key,row} slots in HBM
	uint32_t kind_lds;     // kind | lds_off1 << 8 (lds_off1: 1 + dword offset inside the workgroup's LDS table area, 0 = HBM)
	uint32_t a, b;         // perfect: min, range (32-bit modular); hash: slot mask (capacity <= 2^31), unused
	uint32_t out_slot;     // emitting runs: the output slot of this join's build id (1 + its index in the original order)
};

__device__ __forceinline__ uint32_t flat_uni(uint32_t v) {
	return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint64_t flat_uni64(uint64_t v) {
	return ((uint64_t)flat_uni((uint32_t)(v >> 32)) << 32) | flat_uni((uint32_t)v);
}

__device__ __forceinline__ FlatStage flat_load_stage(const StageDesc *d_generic, uint32_t join_index) {
	const POLR_GLOBAL StageDesc *d = as_global(d_generic);
	FlatStage s;
	s.out_slot = 1u + join_index;
	s.keys = as_global((const uint32_t *)flat_uni64((uint64_t)d->key_data[0]));
	s.valid = as_global((const uint8_t *)flat_uni64((uint64_t)d->key_valid[0]));
	s.table = as_global((const uint32_t *)flat_uni64((uint64_t)d->table));
	const uint32_t kind = flat_uni(d->kind);
	s.kind_lds = kind | (flat_uni(d->lds_off1) << 8);
	s.a = kind == KIND_PERFECT ? flat_uni((uint32_t)d->min_value) : flat_uni((uint32_t)d->mask);
	s.b = flat_uni((uint32_t)d->range);
	return s;
}

template <int K>
struct FlatCtx {
	uint32_t k, lane;
	FlatStage st[K]; // the current join order, statically indexed (SGPRs)
	const POLR_GLOBAL uint32_t *sel;
	const POLR_LDS uint32_t *lds_tables; // the workgroup's LDS-resident bit tables
	POLR_LDS uint16_t *q;                // this wave's queues
	uint32_t qsize[K], cnt[K];
	uint64_t unit_begin, in_pos, in_end;
	// emitting runs (a materialising sink behind a bank of perfect tables): the final tuples as row ids, chunked like a
	// DataChunk stream -- slot 0 the probe row, slot 1 + j the build id of join j = its key's offset in the perfect table
	DevOut out;
	bool emit, overflow;
	uint32_t cur_chunk, fill;
	POLR_LDS unsigned long long *fused_lds; // the workgroup's group cells of a fused GROUP BY sink, when they fit in LDS
	// stage 0's next step, requested one step ahead (the key stream's HBM round trip overlaps the current step)
	uint4 pf0, pf1;
	uint64_t pf_pos; // source position the prefetched keys belong to; ~0: none
};

__device__ __forceinline__ uint32_t flat_rank(uint64_t m) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

// ---- membership test in two halves: issue (addresses + loads in flight), then resolve -------------------------
// perfect table: 32-bit modular arithmetic, exact for signed and unsigned 4-byte keys while [min, max] lies inside the
// key type's domain (checked on the host).  w = the loaded bit word (0 where the key is out of range / inactive).
template <int F>
__device__ __forceinline__ void flat_perfect_issue(const FlatStage &s, const POLR_LDS uint32_t *lds_tables, const uint32_t (&key)[F],
                                                   const bool (&act)[F], uint32_t (&idx)[F], uint32_t (&w)[F]) {
	const uint32_t lds_off1 = s.kind_lds >> 8;
	bool in[F];
#pragma unroll
	for (int i = 0; i < F; i++) {
		idx[i] = key[i] - s.a;
		in[i] = act[i] && idx[i] <= s.b;
		w[i] = 0;
	}
	if (lds_off1) {
		const POLR_LDS uint32_t *bits = lds_tables + (lds_off1 - 1u);
#pragma unroll
		for (int i = 0; i < F; i++) {
			if (in[i]) {
				w[i] = bits[idx[i] >> 5];
			}
		}
	} else {
#pragma unroll
		for (int i = 0; i < F; i++) {
			if (in[i]) {
				w[i] = s.table[idx[i] >> 5];
			}
		}
	}
}

// unique-key hash table: linear probing, one aligned 32-byte group of 4 {key,row} slots per round trip, four probes in
// flight per lane (eight would hold 64 VGPRs of slot data)
template <int F>
__device__ __forceinline__ void flat_hash_lookup(const FlatStage &s, const uint32_t (&key)[F], const bool (&act)[F],
                                                 bool (&hit)[F]) {
	const POLR_GLOBAL uint32_t *tab = s.table; // (groups of 8 dwords)
#pragma unroll
	for (int h0 = 0; h0 < F; h0 += 4) {
		uint32_t group[4], first[4];
		bool searching[4];
		bool any = false;
		const uint32_t gmask = s.a >> 2;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const uint64_t h = polr_murmurhash64((uint64_t)key[h0 + i]) & (uint64_t)s.a;
			group[i] = (uint32_t)(h >> 2);
			first[i] = (uint32_t)(h & 3u);
			searching[i] = act[h0 + i];
			hit[h0 + i] = false;
			any = any || searching[i];
		}
		while (__ballot(any) != 0ull) {
			uint4 a[4], b[4];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				if (searching[i]) {
					a[i] = load_global_x4(tab + (uint64_t)group[i] * 8);
					b[i] = load_global_x4(tab + (uint64_t)group[i] * 8 + 4);
				}
			}
			any = false;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				if (searching[i]) {
					const uint32_t kk[4] = {a[i].x, a[i].z, b[i].x, b[i].z};
					const uint32_t rr[4] = {a[i].y, a[i].w, b[i].y, b[i].w};
#pragma unroll
					for (int j = 0; j < 4; j++) {
						if (searching[i] && (uint32_t)j >= first[i]) {
							if (rr[j] == S8_EMPTY_ROW) {
								searching[i] = false;
							} else if (kk[j] == key[h0 + i]) {
								hit[h0 + i] = true;
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
}

template <int F>
__device__ __forceinline__ void flat_lookup(const FlatStage &s, const POLR_LDS uint32_t *lds_tables, const uint32_t (&key)[F],
                                            const bool (&act)[F], bool (&hit)[F]) {
	if ((s.kind_lds & 0xFFu) == KIND_PERFECT) {
		uint32_t idx[F], w[F];
		flat_perfect_issue<F>(s, lds_tables, key, act, idx, w);
#pragma unroll
		for (int i = 0; i < F; i++) {
			hit[i] = (w[i] >> (idx[i] & 31u)) & 1u;
		}
	} else {
		flat_hash_lookup<F>(s, key, act, hit);
	}
}

#ifndef POLR_FLAT_EMIT
#define POLR_FLAT_EMIT 0 // 1: the build whose last join can write row ids (a counting run never pays for that code:
                         // inlined at every place a stage emits it cost the SF100 headline 6 % -- 96 more spilled SGPRs)
#endif
#define FLAT_NO_CHUNK 0xFFFFFFFFu
// one cell of a column of the aggregate sinks, as a signed 64-bit value (narrow unsigned values zero-extended)
__device__ __forceinline__ long long flat_sink_cell(const DevCol &c, uint32_t row) {
	const POLR_GLOBAL uint8_t *p = as_global(c.data) + (uint64_t)row * c.width;
	const bool sx = (c.flags & 1u) != 0;
	switch (c.width) {
	case 1:
		return sx ? (long long)*(const POLR_GLOBAL int8_t *)p : (long long)*p;
	case 2:
		return sx ? (long long)*(const POLR_GLOBAL int16_t *)p : (long long)*(const POLR_GLOBAL uint16_t *)p;
	case 4:
		return sx ? (long long)*(const POLR_GLOBAL int32_t *)p : (long long)*(const POLR_GLOBAL uint32_t *)p;
	default:
		return *(const POLR_GLOBAL long long *)p;
	}
}

// the fused GROUP BY sink (FusedSink, polr_device.h): slot 0 = the probe row, slot 1 + j = the build id of join j, which
// the stage that probed join j holds (id[p], out_slot = 1 + j).  One lane = one surviving tuple.
template <int K>
__device__ __forceinline__ void flat_fused_accumulate(FlatCtx<K> &c, uint32_t row) {
	const FusedSink *f = c.out.fused; // (a generic pointer: the descriptor is read a few times per surviving tuple)
	auto slot_row = [&](uint32_t slot) { // (slot is wave-uniform: one stage's key column is read, if any)
		uint32_t r = row;
#pragma unroll
		for (int p = 0; p < K; p++) {
			if (p < (int)c.k && c.st[p].out_slot == slot) {
				r = c.st[p].keys[row] - c.st[p].a;
			}
		}
		return r;
	};
	uint64_t g = 0;
	bool ok = true;
	for (uint32_t q = 0; q < f->groups.n; q++) {
		const DevGroupKey gk = f->groups.k[q];
		const uint32_t r = slot_row(gk.slot);
		if (gk.src.valid && !as_global(gk.src.valid)[r]) {
			ok = false;
			break;
		}
		const unsigned long long off = (unsigned long long)(flat_sink_cell(gk.src, r) - gk.min_value);
		if (off >= gk.n_values) {
			ok = false;
			break;
		}
		g = g * gk.n_values + off;
	}
	if (!ok) {
		__hip_atomic_fetch_add(f->dropped, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return;
	}
	const uint32_t n_aggs = f->aggs.n;
	// the cells of this workgroup: in LDS when they fit (flushed when the workgroup leaves: 25 M adds to a few hundred
	// global addresses cost the SF100 run 1.9 ms), else its table in global memory; adds nobody waits for either way
	const size_t at = g * (1u + 2u * n_aggs);
	unsigned long long *gcell = f->cells + (size_t)(blockIdx.x % f->n_tables) * f->words_per_table + at;
	POLR_LDS unsigned long long *lcell = c.fused_lds + at;
	const bool in_lds = c.fused_lds != nullptr;
	auto add = [&](uint32_t w, unsigned long long v) {
		if (in_lds) {
			__hip_atomic_fetch_add(&lcell[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else {
			__hip_atomic_fetch_add(&gcell[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	};
	add(0, 1ull);
	for (uint32_t a = 0; a < n_aggs; a++) {
		const DevAgg ag = f->aggs.a[a];
		if (ag.fn == POLR_DEV_AGG_COUNT_STAR) {
			continue; // (= the rows of the group)
		}
		const uint32_t r = slot_row(ag.slot);
		if (ag.src.valid) {
			if (!as_global(ag.src.valid)[r]) {
				continue;
			}
			add(2u + 2u * a, 1ull);
		}
		if (ag.fn == POLR_DEV_AGG_SUM) {
			add(1u + 2u * a, (unsigned long long)flat_sink_cell(ag.src, r));
		}
	}
}

// the tuples that survived the last join leave as row ids (emitting runs): the probe row, and for every join the build id
// its key stands for -- every join of an emitting flat pipeline is a perfect table, whose build id IS key - min
// (RowOperations::Gather reads payload columns re-ordered that way: polr_gather.hip) -- re-read from the probe columns
// for the few tuples that get this far
template <int K>
__device__ __forceinline__ void flat_out_write(FlatCtx<K> &c, uint32_t pos, bool valid) {
	const uint64_t m = __ballot(valid);
	const uint32_t n = (uint32_t)__popcll(m);
	if (n == 0) {
		return;
	}
	const uint32_t rank = flat_rank(m);
	POLR_GLOBAL uint32_t *ids = as_global(c.out.ids);
	POLR_GLOBAL uint32_t *chunk_count = as_global(c.out.chunk_count);
	uint32_t row = 0, id[K];
	if (valid) {
		const uint64_t tp = c.unit_begin + pos;
		row = c.sel ? c.sel[tp] : (uint32_t)tp;
	}
	if (c.out.fused) {
		// a fused GROUP BY sink: the tuple is folded into the group cells of its workgroup, nothing is written (and only
		// the build ids the sink's columns hang on are re-read)
		if (valid) {
			flat_fused_accumulate<K>(c, row);
		}
		return;
	}
	if (valid) {
#pragma unroll
		for (int p = 0; p < K; p++) {
			id[p] = p < (int)c.k ? c.st[p].keys[row] - c.st[p].a : 0u;
		}
	}
	uint32_t done = 0;
	while (done < n) {
		if (c.cur_chunk == FLAT_NO_CHUNK || c.fill == c.out.chunk_capacity) {
			if (c.cur_chunk != FLAT_NO_CHUNK && c.lane == 0) {
				chunk_count[c.cur_chunk] = c.fill;
			}
			uint32_t nc = 0;
			if (c.lane == 0) {
				nc = atomicAdd(&c.out.cursor[0], 1u);
			}
			nc = flat_uni(nc);
			if (nc >= c.out.max_chunks) {
				if (c.lane == 0) {
					atomicExch(&c.out.cursor[1], 1u);
				}
				c.overflow = true;
				c.cur_chunk = FLAT_NO_CHUNK;
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
			ids[base] = row;
#pragma unroll
			for (int p = 0; p < K; p++) {
				if (p < (int)c.k) {
					ids[(uint64_t)c.st[p].out_slot * c.out.slot_stride + base] = id[p];
				}
			}
		}
		c.fill += take;
		done += take;
	}
}

// survivors of stage POS: count them; push their unit positions to the next stage's queue unless POS is the last join
template <int K, int POS, int F>
__device__ __forceinline__ void flat_emit(FlatCtx<K> &c, const uint32_t (&pos)[F], const bool (&hit)[F]) {
	const bool last = POS + 1 >= K || POS + 1 == (int)c.k;
	uint32_t total = 0;
#if POLR_FLAT_EMIT
	if (last && c.emit) {
		// (ONE copy of the sink's code per place a stage emits, not F: the loop stays a loop -- the element is picked
		// with selects -- or the emitting build is 2.4 ms where the counting build is 1.5)
#pragma unroll 1
		for (int i = 0; i < F; i++) {
			uint32_t p_i = pos[0];
			bool h_i = hit[0];
#pragma unroll
			for (int j = 1; j < F; j++) {
				p_i = j == i ? pos[j] : p_i;
				h_i = j == i ? hit[j] : h_i;
			}
			flat_out_write<K>(c, p_i, h_i);
		}
	}
#endif
	if constexpr (POS + 1 < K) {
		POLR_LDS uint16_t *qq = c.q + flat_qoff<K>(POS + 1);
		uint32_t qs = c.qsize[POS + 1];
#pragma unroll
		for (int i = 0; i < F; i++) {
			const uint64_t m = __ballot(hit[i]);
			const uint32_t n = (uint32_t)__popcll(m);
			if (!last && hit[i]) {
				qq[qs + flat_rank(m)] = (uint16_t)pos[i];
			}
			qs += n;
			total += n;
		}
		if (!last) {
			c.qsize[POS + 1] = qs;
		}
	} else {
#pragma unroll
		for (int i = 0; i < F; i++) {
			total += (uint32_t)__popcll(__ballot(hit[i]));
		}
	}
	c.cnt[POS] += total;
}

// one step of stage 0: up to 512 tuples straight from the source, looked up and pushed on in two halves of 256 (four
// keys per lane live at a time: the step's registers are its 8 keys + the 8 prefetched ones)
template <int K>
__device__ __forceinline__ void flat_stage0(FlatCtx<K> &c) {
	constexpr int F = 4;
	const FlatStage &s = c.st[0];
	const uint64_t left = c.in_end - c.in_pos;
	const uint32_t n = left < FLAT_STEP0 ? (uint32_t)left : (uint32_t)FLAT_STEP0;
	const uint32_t upos = (uint32_t)(c.in_pos - c.unit_begin); // position of the step inside the unit
	const bool streamable = c.sel == nullptr && s.valid == nullptr && (c.in_pos & 3ull) == 0 && (((uint64_t)s.keys) & 15ull) == 0;
	if (n == FLAT_STEP0 && streamable) {
		// lane l holds tuples in_pos + 256 h + 4 l + {0..3}: two 16-byte loads per lane
		const uint32_t base = (uint32_t)c.in_pos + 4u * c.lane;
		uint4 kk[2];
		if (c.pf_pos == c.in_pos) {
			kk[0] = c.pf0;
			kk[1] = c.pf1;
		} else {
			kk[0] = load_global_x4(s.keys + base);
			kk[1] = load_global_x4(s.keys + base + 256u);
		}
		if (c.in_end - c.in_pos >= 2ull * FLAT_STEP0) {
			c.pf0 = load_global_x4(s.keys + base + FLAT_STEP0);
			c.pf1 = load_global_x4(s.keys + base + FLAT_STEP0 + 256u);
			c.pf_pos = c.in_pos + FLAT_STEP0;
		} else {
			c.pf_pos = ~0ull;
		}
		c.in_pos += n;
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const uint32_t key[F] = {kk[h].x, kk[h].y, kk[h].z, kk[h].w};
			uint32_t pos[F];
			bool act[F], hit[F];
#pragma unroll
			for (int i = 0; i < F; i++) {
				pos[i] = upos + 4u * c.lane + h * 256u + i;
				act[i] = true;
			}
			flat_lookup<F>(s, c.lds_tables, key, act, hit);
			flat_emit<K, 0, F>(c, pos, hit);
		}
		return;
	}
	c.pf_pos = ~0ull;
	const uint64_t pos0 = c.in_pos;
	c.in_pos += n;
#pragma unroll 1
	for (uint32_t h = 0; h * 256u < n; h++) {
		uint32_t pos[F], row[F], key[F];
		bool act[F], hit[F];
#pragma unroll
		for (int i = 0; i < F; i++) {
			const uint32_t off = h * 256u + c.lane + 64u * i;
			act[i] = off < n;
			pos[i] = upos + off;
			const uint64_t tp = pos0 + off;
			row[i] = act[i] ? (c.sel ? c.sel[tp] : (uint32_t)tp) : 0u;
		}
#pragma unroll
		for (int i = 0; i < F; i++) {
			key[i] = act[i] ? s.keys[row[i]] : 0u;
		}
		if (s.valid) {
#pragma unroll
			for (int i = 0; i < F; i++) {
				act[i] = act[i] && s.valid[row[i]] != 0; // NULL never matches (join_hashtable.cpp:170-192)
			}
		}
		flat_lookup<F>(s, c.lds_tables, key, act, hit);
		flat_emit<K, 0, F>(c, pos, hit);
	}
}

// ---- the sweep over the deeper stages ------------------------------------------------------------------------------
template <int I, int N, class Fn>
__device__ __forceinline__ void flat_static_for(Fn &&fn) {
	if constexpr (I < N) {
		fn(std::integral_constant<int, I> {});
		flat_static_for<I + 1, N>(fn);
	}
}
template <int I, int N, class Fn>
__device__ __forceinline__ void flat_static_for_down(Fn &&fn) { // I = N-1 ... 1
	if constexpr (I >= 1) {
		fn(std::integral_constant<int, I> {});
		flat_static_for_down<I - 1, N>(fn);
	}
}

// All deeper stages at once.  Per stage three registers per tuple live across the memory round trips: its position in
// the unit, its key (after the lookup: the bit to test), the table word (hash stages: the hit flag).
//   1. pop up to one sweep from EVERY deeper queue and request the keys    -> all key gathers in flight together
//   2. bit-table stages: request the bit words                             -> all table loads in flight together
//      hash stages: probe
//   3. push the survivors on, deepest stage first (the queue a stage pushes into has just been popped: it has room)
template <int K>
__device__ __forceinline__ void flat_sweep(FlatCtx<K> &c) {
	constexpr int F = flat_sweep_f<K>();
	if constexpr (K > 1) {
		uint32_t pos[K - 1][F], key[K - 1][F], w[K - 1][F];
		flat_static_for<1, K>([&](auto P) {
			constexpr int p = decltype(P)::value;
			if (p < (int)c.k) {
				const FlatStage &s = c.st[p];
				const uint32_t qs = c.qsize[p];
				const uint32_t n = qs < (uint32_t)(64 * F) ? qs : (uint32_t)(64 * F);
				const uint32_t base = qs - n;
				const POLR_LDS uint16_t *qq = c.q + flat_qoff<K>(p);
				uint32_t row[F];
#pragma unroll
				for (int i = 0; i < F; i++) {
					const uint32_t off = c.lane + 64u * i;
					// (an inactive slot keeps an impossible position: its key is never loaded, its word stays 0)
					pos[p - 1][i] = off < n ? (uint32_t)qq[base + off] : 0xFFFFFFFFu;
				}
				c.qsize[p] = base;
				if (c.sel) {
#pragma unroll
					for (int i = 0; i < F; i++) {
						row[i] = pos[p - 1][i] != 0xFFFFFFFFu ? c.sel[c.unit_begin + pos[p - 1][i]] : 0u;
					}
				} else {
#pragma unroll
					for (int i = 0; i < F; i++) {
						row[i] = (uint32_t)c.unit_begin + pos[p - 1][i];
					}
				}
#pragma unroll
				for (int i = 0; i < F; i++) {
					key[p - 1][i] = pos[p - 1][i] != 0xFFFFFFFFu ? s.keys[row[i]] : 0u;
				}
				if (s.valid) {
#pragma unroll
					for (int i = 0; i < F; i++) {
						if (pos[p - 1][i] != 0xFFFFFFFFu && s.valid[row[i]] == 0) {
							pos[p - 1][i] = 0xFFFFFFFFu; // NULL never matches
						}
					}
				}
			}
		});
		flat_static_for<1, K>([&](auto P) {
			constexpr int p = decltype(P)::value;
			if (p < (int)c.k && (c.st[p].kind_lds & 0xFFu) == KIND_PERFECT) {
				const FlatStage &s = c.st[p];
				const uint32_t lds_off1 = s.kind_lds >> 8;
				bool in[F];
#pragma unroll
				for (int i = 0; i < F; i++) {
					const uint32_t idx = key[p - 1][i] - s.a;
					in[i] = pos[p - 1][i] != 0xFFFFFFFFu && idx <= s.b;
					key[p - 1][i] = idx; // (the bit to test is idx & 31, taken when the word is there)
					w[p - 1][i] = 0;
				}
				// (one address space per stage, wave-uniform: DS_READ or GLOBAL_LOAD, never a generic access)
				if (lds_off1) {
					const POLR_LDS uint32_t *bits = c.lds_tables + (lds_off1 - 1u);
#pragma unroll
					for (int i = 0; i < F; i++) {
						if (in[i]) {
							w[p - 1][i] = bits[key[p - 1][i] >> 5];
						}
					}
				} else {
#pragma unroll
					for (int i = 0; i < F; i++) {
						if (in[i]) {
							w[p - 1][i] = s.table[key[p - 1][i] >> 5];
						}
					}
				}
#pragma unroll
				for (int i = 0; i < F; i++) {
					key[p - 1][i] &= 31u;
				}
			}
		});
		flat_static_for<1, K>([&](auto P) {
			constexpr int p = decltype(P)::value;
			if (p < (int)c.k && (c.st[p].kind_lds & 0xFFu) != KIND_PERFECT) {
				bool act[F], hit[F];
#pragma unroll
				for (int i = 0; i < F; i++) {
					act[i] = pos[p - 1][i] != 0xFFFFFFFFu;
				}
				flat_hash_lookup<F>(c.st[p], key[p - 1], act, hit);
#pragma unroll
				for (int i = 0; i < F; i++) {
					w[p - 1][i] = hit[i] ? 1u : 0u;
					key[p - 1][i] = 0;
				}
			}
		});
		flat_static_for_down<K - 1, K>([&](auto P) {
			constexpr int p = decltype(P)::value;
			if (p < (int)c.k) {
				bool hit[F];
#pragma unroll
				for (int i = 0; i < F; i++) {
					hit[i] = (w[p - 1][i] >> key[p - 1][i]) & 1u;
				}
				flat_emit<K, p, F>(c, pos[p - 1], hit);
			}
		});
	}
}

// One unit [in_pos, in_end) in two phases, so that the caller can look for its next unit in between:
//   flat_run_source -- a sweep whenever a deeper stage holds a full step, else the source, until the source is used up;
//   flat_run_drain  -- sweeps until every queue is empty (a unit leaves nothing behind: its counters are final when it
//                      arrives).
// (Measured and dropped: sending a small unit, or the queue remainders at the end of a unit, down ALL remaining stages
// at once -- two round trips instead of two per stage, but keys and table words are then loaded for tuples that are
// already dead, and on a device whose memory system is the bottleneck that costs more than the chain it saves: SF100
// run 1.58 -> 1.77 ms.)
template <int K>
__device__ __forceinline__ void flat_run_source(FlatCtx<K> &c) {
	while (true) {
		bool deep_full = false;
#pragma unroll
		for (int p = 1; p < K; p++) {
			if (p < (int)c.k) {
				deep_full = deep_full || c.qsize[p] >= (uint32_t)(64 * flat_sweep_f<K>());
			}
		}
		if (deep_full) {
			flat_sweep<K>(c);
		} else if (c.in_pos < c.in_end) {
			flat_stage0<K>(c);
		} else {
			return;
		}
	}
}

template <int K>
__device__ __forceinline__ void flat_run_drain(FlatCtx<K> &c) {
	while (true) {
		bool deep_any = false;
#pragma unroll
		for (int p = 1; p < K; p++) {
			if (p < (int)c.k) {
				deep_any = deep_any || c.qsize[p] > 0;
			}
		}
		if (!deep_any) {
			return;
		}
		flat_sweep<K>(c);
	}
}

// the first source step of the unit AFTER the current one, requested while the current one drains (same join order,
// streamable source): flat_stage0 finds the keys in pf0 / pf1 when that unit starts
template <int K>
__device__ __forceinline__ void flat_prefetch_unit(FlatCtx<K> &c, uint64_t begin, uint32_t count) {
	const FlatStage &s = c.st[0];
	const bool streamable = c.sel == nullptr && s.valid == nullptr && (begin & 3ull) == 0 && (((uint64_t)s.keys) & 15ull) == 0;
	if (count >= (uint32_t)FLAT_STEP0 && streamable) {
		const uint32_t base = (uint32_t)begin + 4u * c.lane;
		c.pf0 = load_global_x4(s.keys + base);
		c.pf1 = load_global_x4(s.keys + base + 256u);
		c.pf_pos = begin;
	}
}
