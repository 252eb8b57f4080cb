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
//
// Compiled once per stage count K (-DPOLR_K=2|4|8, K >= k) and instantiated per W (tuple slots):
// stage positions are compile-time, so all per-stage state sits in registers with static indices, and
// the per-(join order, position) descriptors are pre-resolved on the host (StageDesc), staged in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "polr_device.h"
#include "polr_mpx_device.h"

#ifndef POLR_K
#error "compile with -DPOLR_K=<compiled stage count>"
#endif

#include "polr_probe_device.h"

template <int W, int K>
__global__ __launch_bounds__(256) void polr_path_kernel(const DevPipeline *__restrict__ pipe,
                                                        const DevRound *__restrict__ rounds,
                                                        const uint64_t *__restrict__ unit_prefix, uint32_t n_rounds,
                                                        const uint32_t *__restrict__ unit_sizes, DevOut out,
                                                        unsigned long long *__restrict__ counts, SelfRoute sr) {
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wave_in_block = threadIdx.x >> 6;
	STAMP(0)
	if (sr.mpx) {
		// self-routing launch: this launch reads descriptor slot (iter & 1), its last workgroup writes the
		// other slot for the next launch (a late-starting idle workgroup must never see the new round)
		const uint32_t slot = sr.iter & 1u;
		rounds += slot;
		unit_prefix += 2 * slot;
		unit_sizes += slot;
	}
	const uint64_t total_units = unit_prefix[n_rounds];
	if (sr.mpx && total_units == 0) {
		// nothing routed (the run is over): keep the other slot empty too, or the launch after this one
		// would find the stale round of two launches ago in it
		if (blockIdx.x == 0 && threadIdx.x == 0) {
			const uint32_t next = (sr.iter + 1u) & 1u;
			sr.prefix_base[2 * next] = 0;
			sr.prefix_base[2 * next + 1] = 0;
		}
		return;
	}
	const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
	const uint64_t wave_id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
	const uint64_t busy_waves = total_units < n_waves ? total_units : n_waves;
	const uint32_t busy_blocks = (uint32_t)((busy_waves + (blockDim.x >> 6) - 1) / (blockDim.x >> 6));
	if (blockIdx.x >= busy_blocks) {
		return; // whole workgroup idle: not part of the arrival count either
	}
	if (!sr.mpx && wave_id >= total_units) {
		return;
	}
	const uint32_t k = uni(pipe->k);
	const uint32_t per_wave = per_wave_dwords<W, K>();
	WaveCtx<W, K> c;
	c.k = k;
	c.lane = threadIdx.x & 63;
	uint32_t *base = lds + wave_in_block * per_wave;
	c.desc = (StageDesc *)base;
	c.q = base + K * STAGE_DESC_DWORDS;
	c.pend_start = c.q + qtotal<W, K>();
	c.pend_pref = c.pend_start + K * 64;
	c.batch0 = c.pend_pref + K * 64;
	c.wpend_start = c.batch0 + 64 * WIDE;
	c.wpend_pref = c.wpend_start + wide_pend_slots<W, K>() * 64 * WIDE;
#pragma unroll
	for (int p = 0; p < K; p++) {
		c.qsize[p] = c.pend_T[p] = c.pend_cur[p] = c.pend_base[p] = c.cnt[p] = 0;
	}
	c.sel = uniptr(pipe->sel);
	c.in_pos = c.in_end = 0;
	c.wide_mask = 0;
	c.pend_wide = 0;
	c.flush_token = 0;
	c.out = out;
	c.emit = false;
	c.cur_chunk = NO_CHUNK;
	c.fill = 0;
	c.overflow = false;

	const StageDesc *stages = uniptr(pipe->stages);
	int64_t cur_round = -1;
	STAMP(1)

	for (uint64_t unit = wave_id; unit < total_units; unit += n_waves) { // (idle waves fall through)
		uint32_t lo = 0, hi = n_rounds - 1; // last r with unit_prefix[r] <= unit
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
				run_until_idle(c, true);
				flush_counts(c, counts, cur_round);
			}
			cur_round = r;
			const uint32_t pidx = uni(rounds[r].path);
			// stage the join order's descriptors: K * 28 dwords, cooperative copy into this wave's LDS
			const uint32_t *src = (const uint32_t *)(stages + (uint64_t)pidx * POLR_KMAX);
			uint32_t *dst = (uint32_t *)c.desc;
			for (uint32_t i = c.lane; i < K * STAGE_DESC_DWORDS; i += 64) {
				dst[i] = src[i];
			}
			c.emit = uni(rounds[r].emit) != 0 && !c.overflow;
			c.wide_mask = stage_wide_mask<W, K>(src, c.k);
		}
		const uint64_t rb = uni64(rounds[r].begin);
		const uint64_t rc = uni64(rounds[r].count);
		const uint32_t us = uni(unit_sizes[r]);
		const uint64_t local = unit - uni64(unit_prefix[r]);
		c.in_pos = rb + local * us;
		c.in_end = c.in_pos + us;
		if (c.in_end > rb + rc) {
			c.in_end = rb + rc;
		}
		run_until_idle(c, false);
	}
	STAMP(2)
	if (cur_round >= 0) {
		run_until_idle(c, true);
		STAMP(3)
		flush_counts(c, counts, cur_round);
	}
	if (c.cur_chunk != NO_CHUNK && c.lane == 0) {
		out.chunk_count[c.cur_chunk] = c.fill;
	}
	STAMP(4)
	if (sr.mpx) {
		// Arrival: every busy workgroup publishes its counters (device-scope atomics above), then takes a
		// ticket; the workgroup that takes the last one has seen all arrivals and routes the next round.
		__shared__ uint32_t is_last;
		__shared__ uint32_t tokens;
		if (threadIdx.x == 0) {
			tokens = 0;
		}
		__syncthreads();
		if (c.lane == 0) {
			atomicOr(&tokens, c.flush_token); // consumes the returned counter values of every wave
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			// all counter adds of this workgroup have been performed (their results were consumed above):
			// take the arrival ticket; data crosses workgroups only through device-scope atomics
			const uint32_t ticket = atomicAdd(sr.ticket, 1u + (tokens & 0u));
			is_last = ticket == busy_blocks - 1 ? 1u : 0u;
			if (is_last) {
				atomicExch(sr.ticket, 0u);
			}
		}
		__syncthreads();
		STAMP(5)
		if (is_last && wave_in_block == 0) {
			const uint32_t next = (sr.iter + 1u) & 1u;
			polr_router_step(sr.mpx, sr.rounds_base + next, sr.prefix_base + 2 * next, sr.unit_base + next, counts,
			                 c.k, sr.resident_waves, c.lane, true, lds);
		}
		STAMP(6)
	}
}

// ---- launch ------------------------------------------------------------------------------------
#define PASTE2(a, b) a##b
#define PASTE(a, b) PASTE2(a, b)

static size_t lds_bytes_k(uint32_t W, uint32_t waves_per_block) {
	const size_t queues = POLR_K <= 1 ? 0 : (size_t)W * QCAP1 + (size_t)(POLR_K - 2) * W * QCAPN;
	const size_t wslots = (POLR_K <= 4 && W <= 4) ? 2 : 1; // (wide_pend_slots<W, K>())
	return (size_t)waves_per_block * ((size_t)POLR_K * STAGE_DESC_DWORDS + queues + (size_t)POLR_K * 64 * 2 + 64 * WIDE + wslots * 64 * WIDE * 2) *
	       sizeof(uint32_t); // (the kernels' few static __shared__ words are accounted by the compiler)
}

extern "C++" size_t PASTE(polr_path_lds_bytes_k, POLR_K)(uint32_t W, uint32_t waves_per_block) {
	return lds_bytes_k(W, waves_per_block);
}

template <int W>
static hipError_t launch_w(dim3 grid, dim3 block, size_t lds, hipStream_t stream, const DevPipeline *pipe,
                           const DevRound *rounds, const uint64_t *unit_prefix, uint32_t n_rounds,
                           const uint32_t *unit_sizes, DevOut out, unsigned long long *counts, SelfRoute sr) {
	// raise the dynamic-LDS limit once per (W, size): a host call we do not want on every launch
	static size_t lds_set = 0;
	if (lds > lds_set) {
		hipError_t e = hipFuncSetAttribute((const void *)polr_path_kernel<W, POLR_K>,
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) {
			return e;
		}
		lds_set = lds;
	}
	// hipLaunchKernel returns the status of THIS launch; nothing is read from (or cleared in) the thread's last-error
	// slot, which any other HIP user of the process shares
	void *args[] = {(void *)&pipe,       (void *)&rounds, (void *)&unit_prefix, (void *)&n_rounds,
	                (void *)&unit_sizes, (void *)&out,    (void *)&counts,      (void *)&sr};
	return hipLaunchKernel((const void *)polr_path_kernel<W, POLR_K>, grid, block, args, lds, stream);
}

template <int W>
static int occupancy_w(size_t lds) {
	int blocks = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *)polr_path_kernel<W, POLR_K>, 256, lds) !=
	    hipSuccess) {
		return 1;
	}
	return blocks < 1 ? 1 : blocks;
}

// W never exceeds K + 1 (probe row + one id per join); larger W values are not instantiated for this K
template <int N>
struct Wc {
	static constexpr int v = (N <= POLR_K + 1) ? N : 1;
};

#define POLR_FOR_EACH_W(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)

extern "C++" int PASTE(polr_path_occupancy_k, POLR_K)(uint32_t W, uint32_t waves_per_block) {
	const size_t lds = lds_bytes_k(W, waves_per_block);
	if (W < 1 || W > POLR_K + 1) {
		return 1;
	}
#define OCC_CASE(N)                                                                                                    \
	if (W == N) {                                                                                                      \
		return occupancy_w<Wc<N>::v>(lds);                                                                             \
	}
	POLR_FOR_EACH_W(OCC_CASE)
#undef OCC_CASE
	return 1;
}

extern "C++" hipError_t PASTE(polr_launch_path_kernel_k, POLR_K)(uint32_t W, uint32_t n_blocks,
                                                                 uint32_t waves_per_block, hipStream_t stream,
                                                                 const DevPipeline *pipe, const DevRound *rounds,
                                                                 const uint64_t *unit_prefix, uint32_t n_rounds,
                                                                 const uint32_t *unit_sizes, DevOut out,
                                                                 unsigned long long *counts, SelfRoute sr) {
	const size_t lds = lds_bytes_k(W, waves_per_block);
	dim3 grid(n_blocks), block(64 * waves_per_block);
	if (W < 1 || W > POLR_K + 1) {
		return hipErrorInvalidValue;
	}
#define LAUNCH_CASE(N)                                                                                                 \
	if (W == N) {                                                                                                      \
		return launch_w<Wc<N>::v>(grid, block, lds, stream, pipe, rounds, unit_prefix, n_rounds, unit_sizes, out,      \
		                          counts, sr);                                                                         \
	}
	POLR_FOR_EACH_W(LAUNCH_CASE)
#undef LAUNCH_CASE
	return hipErrorInvalidValue;
}

