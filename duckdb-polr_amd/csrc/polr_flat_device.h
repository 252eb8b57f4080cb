// duckdb-polr_amd/csrc/polr_flat_device.h -- the probe pipeline for FLAT join banks (device code, gfx950).
//
// A pipeline is flat when every multiplexed join (a) is keyed by ONE 4-byte column of the probe table and (b) has
// at most one build row per key (KIND_PERFECT bit table or KIND_S8 unique-key hash table) -- the star joins of
// SSB / SSB-skew and the FK -> PK joins of JOB-light -- and only counters leave the pipeline (COUNT(*) sink, or
// the exploration / ALTERNATE rounds of any sink).  Then a tuple between two joins is just its probe row, the
// reference's RunPath (src/parallel/polar_pipeline_executor.cpp:427-538) degenerates to "AND the k membership
// tests in path order, count the survivors of every prefix" (:486-487), and the generic pipeline's descriptors,
// tuple slots and expansion machinery (polr_probe_device.h) are dead weight.  This version keeps what matters:
//
//   * stage 0 streams its key column with 16-byte loads (8 tuples per lane, 512 per wave step);
//   * bit tables small enough live in LDS for the whole run (copied once per workgroup): a lookup there is free
//     next to the key stream (measured: 4 LDS lookups per tuple at the 5.4 TB/s streaming rate, against
//     190-350 G lookups/s for an L2-resident table, tools/micro/gather_bench.hip);
//   * between stages the survivors' rows wait in per-wave LDS queues until a full 256-tuple step is there
//     (4 independent key gathers + 4 lookups in flight per lane), deepest stage first -- lanes stay full after a
//     selective join, what CacheJoinChunk does for 1024-row chunks (polar_pipeline_executor.cpp:166-195);
//   * the k stage descriptors of the current join order sit in scalar registers (loaded when the order changes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "polr_device.h"

#define FLAT_STEP0 512 // tuples per stage-0 step
#define FLAT_STEPN 256 // tuples per step of a deeper stage
#define FLAT_Q1 (FLAT_STEPN - 1 + FLAT_STEP0) // capacity of the queue behind stage 0
#define FLAT_QN (FLAT_STEPN - 1 + FLAT_STEPN) // ... behind a deeper stage

template <int K>
__device__ __host__ constexpr int flat_qoff(int pos) { // dword offset of the queue that FEEDS stage pos (pos >= 1)
	return pos <= 1 ? 0 : FLAT_Q1 + 1 + (pos - 2) * (FLAT_QN + 1);
}
template <int K>
__device__ __host__ constexpr int flat_per_wave_dwords() {
	return K <= 1 ? 0 : FLAT_Q1 + 1 + (K - 2) * (FLAT_QN + 1);
}

// scalar view of one stage (SGPRs; loaded from the StageDesc array in global memory through the scalar cache)
struct FlatStage {
	const uint32_t *keys;
	const uint8_t *valid;
	const uint32_t *table; // bit words or {key,row} slots in HBM
	uint32_t kind, min32, range32, lds_off1; // lds_off1: 1 + dword offset inside the workgroup's LDS table area, 0 = HBM
	uint64_t mask;
};

__device__ __forceinline__ uint32_t flat_uni(uint32_t v) {
	return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint64_t flat_uni64(uint64_t v) {
	return ((uint64_t)flat_uni((uint32_t)(v >> 32)) << 32) | flat_uni((uint32_t)v);
}

__device__ __forceinline__ FlatStage flat_load_stage(const StageDesc *d) {
	FlatStage s;
	s.keys = (const uint32_t *)flat_uni64((uint64_t)d->key_data[0]);
	s.valid = (const uint8_t *)flat_uni64((uint64_t)d->key_valid[0]);
	s.table = (const uint32_t *)flat_uni64((uint64_t)d->table);
	s.kind = flat_uni(d->kind);
	s.min32 = flat_uni((uint32_t)d->min_value);
	s.range32 = flat_uni((uint32_t)d->range);
	s.lds_off1 = flat_uni(d->lds_off1);
	s.mask = flat_uni64(d->mask);
	return s;
}

template <int K>
struct FlatCtx {
	uint32_t k, lane;
	FlatStage st[K]; // the current join order, statically indexed (SGPRs)
	const uint32_t *sel;
	const uint32_t *lds_tables; // the workgroup's LDS-resident bit tables
	uint32_t *q;                // this wave's queues
	uint32_t qsize[K], cnt[K];
	uint64_t in_pos, in_end;
};

__device__ __forceinline__ uint32_t flat_rank(uint64_t m) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

// membership of F keys per lane: bit table (LDS or HBM) or unique-key hash table.  All loads of one kind are issued
// back to back so that F of them are in flight per lane.
template <int F>
__device__ __forceinline__ void flat_lookup(const FlatStage &s, const uint32_t *lds_tables, const uint32_t (&key)[F],
                                            const bool (&act)[F], bool (&hit)[F]) {
	if (s.kind == KIND_PERFECT) {
		uint32_t idx[F], w[F];
		bool in[F];
#pragma unroll
		for (int i = 0; i < F; i++) {
			// 32-bit modular arithmetic: exact for signed and unsigned 4-byte keys while [min, max] lies inside the key
			// type's domain (checked on the host)
			idx[i] = key[i] - s.min32;
			in[i] = act[i] && idx[i] <= s.range32;
			w[i] = 0;
		}
		if (s.lds_off1) {
			const uint32_t *bits = lds_tables + (s.lds_off1 - 1u);
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
#pragma unroll
		for (int i = 0; i < F; i++) {
			hit[i] = in[i] && ((w[i] >> (idx[i] & 31u)) & 1u);
		}
	} else { // KIND_S8: linear probing, one aligned 32-byte group of 4 {key,row} slots per round trip
		const uint4 *tab = (const uint4 *)s.table;
		// four probes in flight per lane (eight would hold 64 VGPRs of slot data)
#pragma unroll
		for (int h0 = 0; h0 < F; h0 += 4) {
			uint32_t group[4], first[4];
			bool searching[4];
			bool any = false;
			const uint32_t gmask = (uint32_t)(s.mask >> 2);
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint64_t h = polr_murmurhash64((uint64_t)key[h0 + i]) & s.mask;
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
						a[i] = tab[(uint64_t)group[i] * 2];
						b[i] = tab[(uint64_t)group[i] * 2 + 1];
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
}

// survivors of stage POS: count them; push their rows to the next stage's queue unless POS is the last join
template <int K, int POS, int F>
__device__ __forceinline__ void flat_emit(FlatCtx<K> &c, const uint32_t (&row)[F], const bool (&hit)[F]) {
	const bool last = POS + 1 >= K || POS + 1 == (int)c.k;
	uint32_t total = 0;
	if constexpr (POS + 1 < K) {
		uint32_t *qq = c.q + flat_qoff<K>(POS + 1);
		uint32_t qs = c.qsize[POS + 1];
#pragma unroll
		for (int i = 0; i < F; i++) {
			const uint64_t m = __ballot(hit[i]);
			const uint32_t n = (uint32_t)__popcll(m);
			if (!last && hit[i]) {
				qq[qs + flat_rank(m)] = row[i];
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

// one step of stage 0: up to 512 tuples straight from the source
template <int K>
__device__ __forceinline__ void flat_stage0(FlatCtx<K> &c) {
	constexpr int F = FLAT_STEP0 / 64;
	const FlatStage &s = c.st[0];
	const uint64_t left = c.in_end - c.in_pos;
	const uint32_t n = left < FLAT_STEP0 ? (uint32_t)left : (uint32_t)FLAT_STEP0;
	uint32_t row[F], key[F];
	bool act[F], hit[F];
	const bool fast = n == FLAT_STEP0 && c.sel == nullptr && s.valid == nullptr && (c.in_pos & 3ull) == 0 &&
	                  (((uint64_t)s.keys) & 15ull) == 0;
	if (fast) {
		// lane l holds tuples in_pos + 256 h + 4 l + {0..3}: two 16-byte loads per lane
		const uint32_t base = (uint32_t)c.in_pos + 4u * c.lane;
		const uint4 k0 = *(const uint4 *)(s.keys + base);
		const uint4 k1 = *(const uint4 *)(s.keys + base + 256u);
		key[0] = k0.x, key[1] = k0.y, key[2] = k0.z, key[3] = k0.w;
		key[4] = k1.x, key[5] = k1.y, key[6] = k1.z, key[7] = k1.w;
#pragma unroll
		for (int i = 0; i < F; i++) {
			row[i] = base + (i >> 2) * 256u + (i & 3);
			act[i] = true;
		}
	} else {
#pragma unroll
		for (int i = 0; i < F; i++) {
			const uint32_t off = c.lane + 64u * i;
			act[i] = off < n;
			const uint64_t tp = c.in_pos + off;
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
	}
	c.in_pos += n;
	flat_lookup<F>(s, c.lds_tables, key, act, hit);
	flat_emit<K, 0, F>(c, row, hit);
}

// one step of a deeper stage: up to 256 rows from the top of its queue
template <int K, int POS>
__device__ __forceinline__ void flat_stage(FlatCtx<K> &c) {
	constexpr int F = FLAT_STEPN / 64;
	const FlatStage &s = c.st[POS];
	const uint32_t qs = c.qsize[POS];
	const uint32_t n = qs < FLAT_STEPN ? qs : (uint32_t)FLAT_STEPN;
	const uint32_t base = qs - n;
	const uint32_t *qq = c.q + flat_qoff<K>(POS);
	uint32_t row[F], key[F];
	bool act[F], hit[F];
#pragma unroll
	for (int i = 0; i < F; i++) {
		const uint32_t off = c.lane + 64u * i;
		act[i] = off < n;
		row[i] = act[i] ? qq[base + off] : 0u;
	}
	c.qsize[POS] = base;
#pragma unroll
	for (int i = 0; i < F; i++) {
		key[i] = act[i] ? s.keys[row[i]] : 0u;
	}
	if (s.valid) {
#pragma unroll
		for (int i = 0; i < F; i++) {
			act[i] = act[i] && s.valid[row[i]] != 0;
		}
	}
	flat_lookup<F>(s, c.lds_tables, key, act, hit);
	flat_emit<K, POS, F>(c, row, hit);
}

template <int K, int POS>
__device__ __forceinline__ void flat_dispatch(FlatCtx<K> &c, int pick) {
	if (pick == POS) {
		if constexpr (POS == 0) {
			flat_stage0<K>(c);
		} else {
			flat_stage<K, POS>(c);
		}
		return;
	}
	if constexpr (POS + 1 < K) {
		flat_dispatch<K, POS + 1>(c, pick);
	}
}

// run one unit [in_pos, in_end) to completion: deepest stage with a full step first, then the source, then drain
// the partial steps shallowest first (a unit leaves nothing behind: its counters are final when it arrives)
template <int K>
__device__ __forceinline__ void flat_run_unit(FlatCtx<K> &c) {
	while (true) {
		int pick = -1;
#pragma unroll
		for (int p = K - 1; p >= 1; p--) {
			if (pick < 0 && p < (int)c.k && c.qsize[p] >= FLAT_STEPN) {
				pick = p;
			}
		}
		if (pick < 0) {
			if (c.in_pos < c.in_end) {
				pick = 0;
			} else {
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
		flat_dispatch<K, 0>(c, pick);
	}
}
