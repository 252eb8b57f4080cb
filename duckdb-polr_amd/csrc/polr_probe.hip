// duckdb-polr_amd/csrc/polr_probe.hip -- the POLAR path kernel for gfx950 (MI355X, wave64).
//
// One launch executes RunPath (reference: src/parallel/polar_pipeline_executor.cpp:427-538) for a
// list of routed slices ("rounds"): every tuple of a round is sent through the k hash joins of the
// round's join order, fused in one kernel:
//
//   stage j:  key fetch (probe column, or a build column of an earlier join)        <- global gather
//             -> hash / range check -> bucket probe in HBM (PhysicalHashJoin::Execute,
//                JoinHashTable::Probe, ScanStructure::NextInnerJoin, ProbePerfectHashTable)
//             -> wave-ballot + mbcnt prefix compaction of the matches (the selection vector)
//             -> survivors pushed as row-id tuples into the next stage's LDS queue
//   last stage: row ids appended to the wave's current output chunk (adaptive union = slot map)
//   per stage:  wave-uniform counter += matches; flushed with one atomicAdd per (wave, round, stage)
//               -> the multiplexer's AddNumIntermediates input.
//
// Execution model: every WAVE is an independent pipeline (no workgroup barriers).  Between stages
// tuples wait in per-wave LDS queues until a full batch of 64 is available, so lanes stay busy after
// a selective join (what the reference's CacheJoinChunk does for 1024-row chunks,
// polar_pipeline_executor.cpp:166-195).  A join whose key repeats on the build side expands matches
// 64 outputs at a time (prefix sum over the run lengths + binary search), deepest stage first: the
// same depth-first order as the reference's in_process_joins stack, so LDS use is bounded for any
// fan-out.  Nothing but final row ids and k counters is written to HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "polr_device.h"

#define QCAP 128 // per-stage queue capacity in tuples: 63 residual + one batch of 64
#define NO_CHUNK 0xFFFFFFFFu

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
	// number of set bits of mask below this lane
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// wave-uniform small arrays with dynamic index, kept in SGPRs (no scratch): select chains
struct UArr {
	uint32_t v[POLR_KMAX];
	__device__ __forceinline__ uint32_t get(uint32_t i) const {
		uint32_t r = 0;
#pragma unroll
		for (uint32_t q = 0; q < POLR_KMAX; q++) {
			r = (q == i) ? v[q] : r;
		}
		return r;
	}
	__device__ __forceinline__ void set(uint32_t i, uint32_t x) {
#pragma unroll
		for (uint32_t q = 0; q < POLR_KMAX; q++) {
			v[q] = (q == i) ? x : v[q];
		}
	}
	__device__ __forceinline__ void clear() {
#pragma unroll
		for (uint32_t q = 0; q < POLR_KMAX; q++) {
			v[q] = 0;
		}
	}
};

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

// key of join `j` for this lane's tuple; returns false for NULL (NULL never matches:
// join_hashtable.cpp:170-192, perfect_hash_join_executor.cpp:272-277)
template <int W>
__device__ __forceinline__ bool fetch_key(const DevPipeline *pipe, const DevJoin *j, const Tuple<W> &t, bool active,
                                          bool sign_extend, uint64_t &key) {
	key = 0;
	if (!active) {
		return false;
	}
	const uint32_t nk = uni(j->n_keys);
	bool valid = true;
#pragma unroll
	for (uint32_t c = 0; c < 2; c++) {
		if (c < nk) {
			const int32_t sj = (int32_t)uni((uint32_t)j->key_src_join[c]);
			const uint32_t sc = uni((uint32_t)j->key_src_col[c]);
			const DevCol *col;
			uint32_t row;
			if (sj < 0) {
				col = uniptr(pipe->probe_cols) + sc;
				row = t.s[0];
			} else {
				col = uniptr(pipe->joins[sj].payload) + sc;
				const uint32_t slot = uni((uint32_t)pipe->slot_of_join[sj]);
				uint32_t id = 0;
#pragma unroll
				for (int q = 0; q < W; q++) {
					id = ((uint32_t)q == slot) ? t.s[q] : id;
				}
				row = id;
			}
			const uint8_t *data = uniptr(col->data);
			const uint8_t *vld = uniptr(col->valid);
			const uint32_t width = uni(col->width);
			if (vld && !vld[row]) {
				valid = false;
			}
			uint64_t v = load_cell(data + (uint64_t)row * width, width, sign_extend && nk == 1);
			key = (c == 0) ? v : (key | (v << 32));
		}
	}
	return valid;
}

// ---- bucket probes ---------------------------------------------------------------------------
__device__ __forceinline__ bool lookup_perfect(const DevJoin *j, uint64_t key, bool valid, uint32_t &id) {
	const bool is_signed = uni(j->key_signed) != 0;
	const int64_t mn = (int64_t)uni64((uint64_t)j->min_value);
	const uint64_t range = uni64(j->range);
	uint64_t idx;
	bool in_range;
	if (is_signed) {
		const int64_t v = (int64_t)key;
		in_range = v >= mn && (uint64_t)(v - mn) <= range;
		idx = (uint64_t)(v - mn);
	} else {
		in_range = key >= (uint64_t)mn && key - (uint64_t)mn <= range;
		idx = key - (uint64_t)mn;
	}
	bool hit = false;
	if (valid && in_range) {
		const uint32_t *bits = (const uint32_t *)uniptr((const uint8_t *)j->table);
		hit = (bits[idx >> 5] >> (idx & 31)) & 1u;
	}
	id = (uint32_t)idx;
	return hit;
}

__device__ __forceinline__ bool lookup_s8(const DevJoin *j, uint64_t key, bool valid, uint32_t &id) {
	const uint2 *tab = (const uint2 *)uniptr((const uint8_t *)j->table);
	const uint64_t mask = uni64(j->mask);
	const uint32_t k32 = (uint32_t)key;
	uint64_t s = polr_murmurhash64((uint64_t)k32) & mask;
	bool hit = false;
	bool searching = valid;
	id = 0;
	while (searching) {
		const uint2 e = tab[s];
		if (e.y == S8_EMPTY_ROW) {
			searching = false;
		} else if (e.x == k32) {
			hit = true;
			id = e.y;
			searching = false;
		} else {
			s = (s + 1) & mask;
		}
	}
	return hit;
}

__device__ __forceinline__ bool lookup_s16(const DevJoin *j, uint64_t key, bool valid, uint32_t &start,
                                           uint32_t &count) {
	const uint4 *tab = (const uint4 *)uniptr((const uint8_t *)j->table);
	const uint64_t mask = uni64(j->mask);
	start = 0;
	count = 0;
	if (!valid) {
		return false;
	}
	if (key == S16_EMPTY_KEY) {
		start = j->sentinel_start;
		count = j->sentinel_count;
		return count > 0;
	}
	uint64_t s = polr_murmurhash64(key) & mask;
	while (true) {
		const uint4 e = tab[s];
		const uint64_t ek = ((uint64_t)e.y << 32) | e.x;
		if (ek == S16_EMPTY_KEY) {
			return false;
		}
		if (ek == key) {
			start = e.z;
			count = e.w;
			return true;
		}
		s = (s + 1) & mask;
	}
}

// ---- per-wave state ----------------------------------------------------------------------------
template <int W>
struct WaveCtx {
	const DevPipeline *pipe;
	uint32_t k;
	uint32_t lane;
	uint32_t *q;          // LDS: (k-1) queues, slot-major: q[(pos-1)*W*QCAP + slot*QCAP + idx]
	uint32_t *pend_start; // LDS: [k][64]
	uint32_t *pend_pref;  // LDS: [k][64] inclusive prefix of run lengths
	uint32_t *batch0;     // LDS: [64] probe rows of the pinned stage-0 batch
	UArr qsize, pend_T, pend_cur, pend_base, cnt, order;
	// output
	DevOut out;
	bool emit;
	uint32_t cur_chunk, fill;
	bool overflow;
};

template <int W>
__device__ __forceinline__ void out_write(WaveCtx<W> &c, const Tuple<W> &t, bool valid) {
	if (!c.emit || c.out.ids == nullptr) {
		return;
	}
	const uint64_t m = __ballot(valid);
	uint32_t n = (uint32_t)__popcll(m);
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

// push the matches of stage `pos` to the next stage (or to the output when pos is the last join)
template <int W>
__device__ __forceinline__ void emit_tuples(WaveCtx<W> &c, uint32_t pos, Tuple<W> t, uint32_t id, bool valid) {
	const uint32_t jidx = c.order.get(pos);
	const int32_t slot = (int32_t)uni((uint32_t)c.pipe->slot_of_join[jidx]);
#pragma unroll
	for (int i = 0; i < W; i++) {
		t.s[i] = (i == slot) ? id : t.s[i];
	}
	const uint64_t m = __ballot(valid);
	const uint32_t n = (uint32_t)__popcll(m);
	c.cnt.set(pos, c.cnt.get(pos) + n);
	if (pos + 1 == c.k) {
		out_write(c, t, valid);
		return;
	}
	const uint32_t qs = c.qsize.get(pos + 1);
	if (valid) {
		const uint32_t idx = qs + lane_rank(m);
		uint32_t *qq = c.q + (uint64_t)pos * (W * QCAP);
#pragma unroll
		for (int i = 0; i < W; i++) {
			qq[i * QCAP + idx] = t.s[i];
		}
	}
	c.qsize.set(pos + 1, qs + n);
}

// inclusive wave scan of a per-lane u32 via DPP-free shuffles (6 steps)
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

// continue the pending expansion of stage pos: emit the next <= 64 (tuple, build row) pairs
template <int W>
__device__ __forceinline__ void resume_expansion(WaveCtx<W> &c, uint32_t pos) {
	const uint32_t T = c.pend_T.get(pos);
	const uint32_t cur = c.pend_cur.get(pos);
	const uint32_t o = cur + c.lane;
	const bool valid = o < T;
	const uint32_t *pref = c.pend_pref + pos * 64;
	// smallest s with pref[s] > o
	uint32_t lo = 0, hi = 63;
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
	const uint32_t s = valid ? lo : 0;
	const uint32_t excl = s > 0 ? pref[s - 1] : 0;
	const uint32_t r = o - excl;
	const uint32_t st = c.pend_start[pos * 64 + s];
	Tuple<W> t;
	if (pos == 0) {
#pragma unroll
		for (int i = 0; i < W; i++) {
			t.s[i] = 0;
		}
		t.s[0] = c.batch0[s];
	} else {
		const uint32_t base = c.pend_base.get(pos);
		const uint32_t *qq = c.q + (uint64_t)(pos - 1) * (W * QCAP);
#pragma unroll
		for (int i = 0; i < W; i++) {
			t.s[i] = qq[i * QCAP + base + s];
		}
	}
	const uint32_t jidx = c.order.get(pos);
	const DevJoin *j = &c.pipe->joins[jidx];
	uint32_t id = 0;
	if (valid && (int32_t)uni((uint32_t)c.pipe->slot_of_join[jidx]) >= 0) {
		id = uniptr(j->rowids)[st + r];
	}
	emit_tuples(c, pos, t, id, valid);
	const uint32_t ncur = cur + 64;
	if (ncur >= T) {
		c.pend_T.set(pos, 0);
		c.pend_cur.set(pos, 0);
	} else {
		c.pend_cur.set(pos, ncur);
	}
}

// take one batch of <= 64 tuples into stage pos and probe
template <int W>
__device__ __forceinline__ void run_stage(WaveCtx<W> &c, uint32_t pos, uint64_t &in_pos, uint64_t in_end) {
	Tuple<W> t;
#pragma unroll
	for (int i = 0; i < W; i++) {
		t.s[i] = 0;
	}
	bool active;
	uint32_t base = 0;
	if (pos == 0) {
		const uint64_t left = in_end - in_pos;
		const uint32_t n = left < 64 ? (uint32_t)left : 64u;
		active = c.lane < n;
		if (active) {
			const uint32_t *sel = uniptr(c.pipe->sel);
			const uint64_t tp = in_pos + c.lane;
			t.s[0] = sel ? sel[tp] : (uint32_t)tp;
		}
		in_pos += n;
	} else {
		const uint32_t qs = c.qsize.get(pos);
		const uint32_t n = qs < 64 ? qs : 64u;
		base = qs - n;
		active = c.lane < n;
		if (active) {
			const uint32_t *qq = c.q + (uint64_t)(pos - 1) * (W * QCAP);
#pragma unroll
			for (int i = 0; i < W; i++) {
				t.s[i] = qq[i * QCAP + base + c.lane];
			}
		}
		c.qsize.set(pos, base);
	}
	const uint32_t jidx = c.order.get(pos);
	const DevJoin *j = &c.pipe->joins[jidx];
	const uint32_t kind = uni(j->kind);
	uint64_t key;
	const bool valid = fetch_key<W>(c.pipe, j, t, active, kind == KIND_PERFECT && uni(j->key_signed) != 0, key);
	if (kind == KIND_PERFECT) {
		uint32_t id;
		const bool hit = lookup_perfect(j, key, valid, id);
		emit_tuples(c, pos, t, id, hit);
	} else if (kind == KIND_S8) {
		uint32_t id;
		const bool hit = lookup_s8(j, key, valid, id);
		emit_tuples(c, pos, t, id, hit);
	} else {
		uint32_t start, count;
		lookup_s16(j, key, valid, start, count);
		const bool multi = __ballot(count > 1) != 0ull;
		if (!multi) {
			uint32_t id = 0;
			if (count && (int32_t)uni((uint32_t)c.pipe->slot_of_join[jidx]) >= 0) {
				id = uniptr(j->rowids)[start];
			}
			emit_tuples(c, pos, t, id, count != 0);
		} else {
			// pin the batch and expand its runs 64 outputs at a time
			const uint32_t pref = wave_inclusive_scan(count, c.lane);
			const uint32_t T = uni(__shfl(pref, 63, 64));
			c.pend_start[pos * 64 + c.lane] = start;
			c.pend_pref[pos * 64 + c.lane] = pref;
			if (pos == 0) {
				c.batch0[c.lane] = t.s[0];
			} else {
				// the batch was popped above but its cells stay in place in the queue: nothing pushes
				// into this queue while the stage has a pending expansion (deepest stage runs first)
				c.pend_base.set(pos, base);
			}
			c.pend_T.set(pos, T);
			c.pend_cur.set(pos, 0);
			resume_expansion(c, pos);
		}
	}
}

// scheduler: run until the unit's input is consumed and no stage holds a full batch or a pending
// expansion; with `flushing` also drain partial batches, shallowest first.
template <int W>
__device__ __forceinline__ void run_until_idle(WaveCtx<W> &c, uint64_t &in_pos, uint64_t in_end, bool flushing) {
	while (true) {
		int pick = -1;
		bool resume = false;
		for (int p = (int)c.k - 1; p >= 0; p--) {
			if (c.pend_T.get(p) != 0) {
				pick = p;
				resume = true;
				break;
			}
			if (p > 0 && c.qsize.get(p) >= 64) {
				pick = p;
				break;
			}
		}
		if (pick < 0) {
			if (in_pos < in_end) {
				pick = 0;
			} else if (flushing) {
				for (int p = 1; p < (int)c.k; p++) {
					if (c.qsize.get(p) > 0) {
						pick = p;
						break;
					}
				}
			}
		}
		if (pick < 0) {
			return;
		}
		if (resume) {
			resume_expansion(c, (uint32_t)pick);
		} else {
			run_stage(c, (uint32_t)pick, in_pos, in_end);
		}
	}
}

template <int W>
__global__ __launch_bounds__(256) void polr_path_kernel(const DevPipeline *__restrict__ pipe,
                                                        const DevRound *__restrict__ rounds,
                                                        const uint64_t *__restrict__ unit_prefix, uint32_t n_rounds,
                                                        uint32_t unit_size, DevOut out,
                                                        unsigned long long *__restrict__ counts) {
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wave_in_block = threadIdx.x >> 6;
	const uint32_t k = uni(pipe->k);
	const uint32_t per_wave = (k - 1) * W * QCAP + k * 64 * 2 + 64;
	WaveCtx<W> c;
	c.pipe = pipe;
	c.k = k;
	c.lane = threadIdx.x & 63;
	c.q = lds + wave_in_block * per_wave;
	c.pend_start = c.q + (k - 1) * W * QCAP;
	c.pend_pref = c.pend_start + k * 64;
	c.batch0 = c.pend_pref + k * 64;
	c.qsize.clear();
	c.pend_T.clear();
	c.pend_cur.clear();
	c.pend_base.clear();
	c.cnt.clear();
	c.out = out;
	c.emit = false;
	c.cur_chunk = NO_CHUNK;
	c.fill = 0;
	c.overflow = false;

	const uint64_t total_units = unit_prefix[n_rounds];
	const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
	const uint64_t wave_id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
	int64_t cur_round = -1;
	uint64_t in_pos = 0, in_end = 0;

	for (uint64_t unit = wave_id; unit < total_units; unit += n_waves) {
		// round of this unit: last r with unit_prefix[r] <= unit
		uint32_t lo = 0, hi = n_rounds - 1;
		while (lo < hi) {
			const uint32_t mid = (lo + hi + 1) >> 1;
			if (unit_prefix[mid] <= unit) {
				lo = mid;
			} else {
				hi = mid - 1;
			}
		}
		const uint32_t r = uni(lo);
		if ((int64_t)r != cur_round) {
			if (cur_round >= 0) {
				run_until_idle(c, in_pos, in_end, true);
				if (c.lane == 0) {
					for (uint32_t p = 0; p < k; p++) {
						const uint32_t v = c.cnt.get(p);
						if (v) {
							atomicAdd(&counts[(uint64_t)cur_round * k + p], (unsigned long long)v);
						}
					}
				}
				c.cnt.clear();
			}
			cur_round = r;
			const uint32_t pidx = uni(rounds[r].path);
#pragma unroll
			for (int q = 0; q < POLR_KMAX; q++) {
				c.order.v[q] = uni(pipe->paths[pidx].order[q]);
			}
			c.emit = uni(rounds[r].emit) != 0 && !c.overflow;
		}
		const uint64_t rb = uni64(rounds[r].begin);
		const uint64_t rc = uni64(rounds[r].count);
		const uint64_t local = unit - uni64(unit_prefix[r]);
		in_pos = rb + local * unit_size;
		in_end = in_pos + unit_size;
		if (in_end > rb + rc) {
			in_end = rb + rc;
		}
		run_until_idle(c, in_pos, in_end, false);
	}
	if (cur_round >= 0) {
		run_until_idle(c, in_pos, in_end, true);
		if (c.lane == 0) {
			for (uint32_t p = 0; p < k; p++) {
				const uint32_t v = c.cnt.get(p);
				if (v) {
					atomicAdd(&counts[(uint64_t)cur_round * k + p], (unsigned long long)v);
				}
			}
		}
	}
	if (c.cur_chunk != NO_CHUNK && c.lane == 0) {
		out.chunk_count[c.cur_chunk] = c.fill;
	}
}

// ---- launch ------------------------------------------------------------------------------------
extern "C++" size_t polr_path_lds_bytes(uint32_t k, uint32_t W, uint32_t waves_per_block) {
	return (size_t)waves_per_block * ((size_t)(k - 1) * W * QCAP + (size_t)k * 64 * 2 + 64) * sizeof(uint32_t);
}

template <int W>
static hipError_t launch_w(dim3 grid, dim3 block, size_t lds, hipStream_t stream, const DevPipeline *pipe,
                           const DevRound *rounds, const uint64_t *unit_prefix, uint32_t n_rounds,
                           uint32_t unit_size, DevOut out, unsigned long long *counts) {
	// raise the dynamic-LDS limit once per (W, size): a host call we do not want on every launch
	static size_t lds_set = 0;
	if (lds > lds_set) {
		hipError_t e = hipFuncSetAttribute((const void *)polr_path_kernel<W>,
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) {
			return e;
		}
		lds_set = lds;
	}
	hipLaunchKernelGGL(polr_path_kernel<W>, grid, block, lds, stream, pipe, rounds, unit_prefix, n_rounds, unit_size,
	                   out, counts);
	return hipGetLastError();
}

extern "C++" hipError_t polr_launch_path_kernel(uint32_t W, uint32_t k, uint32_t n_blocks, uint32_t waves_per_block,
                                                hipStream_t stream, const DevPipeline *pipe, const DevRound *rounds,
                                                const uint64_t *unit_prefix, uint32_t n_rounds, uint32_t unit_size,
                                                DevOut out, unsigned long long *counts) {
	const size_t lds = polr_path_lds_bytes(k, W, waves_per_block);
	dim3 grid(n_blocks), block(64 * waves_per_block);
#define POLR_CASE(N)                                                                                                   \
	case N:                                                                                                            \
		return launch_w<N>(grid, block, lds, stream, pipe, rounds, unit_prefix, n_rounds, unit_size, out, counts);
	switch (W) {
		POLR_CASE(1)
		POLR_CASE(2)
		POLR_CASE(3)
		POLR_CASE(4)
		POLR_CASE(5)
		POLR_CASE(6)
		POLR_CASE(7)
		POLR_CASE(8)
		POLR_CASE(9)
	default:
		return hipErrorInvalidValue;
	}
#undef POLR_CASE
}

// ---- output materialisation --------------------------------------------------------------------
// One thread per output row of every chunk: dst[row] = src[ids[slot][pos]] (RowOperations::Gather,
// row_gather.cpp:16-86 / DataChunk::Slice).  Rows are numbered chunk-major through `chunk_base`
// (exclusive prefix of chunk_count), so the result is dense.
__global__ void polr_gather_kernel(DevOut out, const uint64_t *__restrict__ chunk_base, uint32_t n_chunks,
                                   uint32_t slot, DevCol src, uint8_t *__restrict__ dst_data,
                                   uint8_t *__restrict__ dst_valid) {
	const uint32_t chunk = blockIdx.x;
	if (chunk >= n_chunks) {
		return;
	}
	const uint32_t n = out.chunk_count[chunk];
	const uint64_t base = chunk_base[chunk];
	const uint32_t *ids = out.ids + (uint64_t)slot * out.slot_stride + (uint64_t)chunk * out.chunk_capacity;
	const uint32_t w = src.width;
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
		const uint32_t row = ids[i];
		const bool valid = src.valid ? src.valid[row] != 0 : true;
		if (dst_valid) {
			dst_valid[base + i] = valid ? 1 : 0;
		}
		const uint8_t *s = src.data + (uint64_t)row * w;
		uint8_t *d = dst_data + (base + i) * w;
		switch (w) {
		case 1:
			*d = valid ? *s : 0;
			break;
		case 2:
			*(uint16_t *)d = valid ? *(const uint16_t *)s : (uint16_t)0;
			break;
		case 4:
			*(uint32_t *)d = valid ? *(const uint32_t *)s : 0u;
			break;
		case 8:
			*(uint64_t *)d = valid ? *(const uint64_t *)s : 0ull;
			break;
		default: {
			uint4 v = valid ? *(const uint4 *)s : make_uint4(0, 0, 0, 0);
			*(uint4 *)d = v;
			break;
		}
		}
	}
}

// compact the row ids of all chunks into a dense [n_rows][W] host-friendly array
__global__ void polr_compact_ids_kernel(DevOut out, const uint64_t *__restrict__ chunk_base, uint32_t n_chunks,
                                        uint32_t *__restrict__ dst) {
	const uint32_t chunk = blockIdx.x;
	if (chunk >= n_chunks) {
		return;
	}
	const uint32_t n = out.chunk_count[chunk];
	const uint64_t base = chunk_base[chunk];
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
		for (uint32_t s = 0; s < out.W_out; s++) {
			dst[(base + i) * out.W_out + s] = out.ids[(uint64_t)s * out.slot_stride + (uint64_t)chunk * out.chunk_capacity + i];
		}
	}
}

extern "C++" void polr_launch_gather(hipStream_t stream, DevOut out, const uint64_t *chunk_base, uint32_t n_chunks,
                                     uint32_t slot, DevCol src, uint8_t *dst_data, uint8_t *dst_valid) {
	if (n_chunks == 0) {
		return;
	}
	hipLaunchKernelGGL(polr_gather_kernel, dim3(n_chunks), dim3(256), 0, stream, out, chunk_base, n_chunks, slot, src,
	                   dst_data, dst_valid);
}

extern "C++" void polr_launch_compact_ids(hipStream_t stream, DevOut out, const uint64_t *chunk_base, uint32_t n_chunks,
                                          uint32_t *dst) {
	if (n_chunks == 0) {
		return;
	}
	hipLaunchKernelGGL(polr_compact_ids_kernel, dim3(n_chunks), dim3(256), 0, stream, out, chunk_base, n_chunks, dst);
}
