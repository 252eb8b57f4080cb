// duckdb-polr_amd/csrc/polr_poolg.hip -- the whole run in ONE launch (gfx950): routers + a pool of probe waves; the GENERIC
// pipeline's kernel (polr_gen_device.h: any key source, repeated build keys, composite keys, conditions, row-id output).
// The flat pipeline's kernel: polr_pool.hip.  Protocol between routers and probe waves: polr_pool_device.h.
//
//   polr_pool_gen_kernel<W, POLR_EXT>    W = tuple slots carried between joins (1 = the probe row only ... 1 + k: every build id,
//                              materialising runs); the number of joins k is a run-time value (<= 8).  512-thread
//                              workgroups, two per CU when the queues allow it.
// Built twice: POLR_EXT = 0 for the pipelines without packed composite keys or non-equality conditions, POLR_EXT = 1 for
// those that have any (exported names end in 'x').
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "polr_device.h"
#include "polr_mpx_device.h"

#ifndef POLR_EXT
#error "compile with -DPOLR_EXT=0|1"
#endif

#include "polr_pool_common.h"
#include "polr_gen_device.h"

#if POLR_EXT
POOL_DIAG_ENTRY(polr_diag_router_gx, polr_diag_timeline_set_gx)
#else
POOL_DIAG_ENTRY(polr_diag_router_g, polr_diag_timeline_set_g)
#endif

#define POOLG_WAVES 8 // waves per workgroup, at most (fewer when the queues of 8 waves would not leave room for two workgroups per CU)

// a probe wave reports a finished unit: its stage counters (returning atomics), then the arrival
__device__ __forceinline__ void poolg_arrive(const ResidentExec *execs, const PoolUnit &u, uint32_t ring, uint32_t k, uint32_t &v_cnt_lo,
                                             uint32_t &v_cnt_hi, uint32_t lane, unsigned long long tokens) {
	const POLR_GLOBAL ResidentExec *xp = as_global(execs) + u.exec;
	POLR_GLOBAL unsigned long long *bank = as_global((unsigned long long *)uni64((uint64_t)xp->counts)) +
	                                       (size_t)u.slot * POLR_NSHARD * POLR_KMAX +
	                                       (size_t)(ring & (POLR_POOL_SHARDS - 1u)) * POLR_KMAX;
	POLR_GLOBAL ResidentSync *sync = as_global((ResidentSync *)uni64((uint64_t)xp->sync));
	// lane p adds stage p's counter (one instruction for all stages); the returned values are folded into the arrival's
	// operand, so the arrival is issued after the adds have been performed
	unsigned long long seen = 0;
	const unsigned long long mine = ((unsigned long long)v_cnt_hi << 32) | v_cnt_lo;
	if (lane < k && mine) {
		seen = __hip_atomic_fetch_add(&bank[lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	seen >>= 63;
	for (int d = 4; d > 0; d >>= 1) {
		seen |= __shfl_xor(seen, d, 64);
	}
	if (lane == 0) {
		__hip_atomic_fetch_add(&sync->arrived[u.slot][ring & (POLR_POOL_SHARDS - 1u)].v, tokens + seen, __ATOMIC_RELAXED,
		                       __HIP_MEMORY_SCOPE_AGENT);
	}
	v_cnt_lo = v_cnt_hi = 0;
}

// (EXT is part of the kernel's name: the two builds of this file are linked into one library)
template <int W, int EXT>
__global__ __launch_bounds__(64 * POOLG_WAVES, 4) void polr_pool_gen_kernel(const DevPipeline *__restrict__ pipe,
                                                                             const ResidentExec *__restrict__ execs, PoolRun *run,
                                                                             DevOut out, uint32_t lds_per_wave, uint32_t qcap) {
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wave_in_block = threadIdx.x >> 6;
	const uint32_t k = uni(pipe->k);
	PoolRun rh;
	pool_load_run(run, rh);
	if (blockIdx.x < rh.n_router_blocks) {
		pool_router_wave(execs, run, rh, k, 64, lds, lds_per_wave); // (units of big rounds: multiples of 64 tuples)
		return;
	}
	// probe wave g of the pool serves ring g % n_rings (dealt wave by wave: every ring the same number of waves)
	const uint32_t pool_wave = (blockIdx.x - rh.n_router_blocks) * (blockDim.x >> 6) + wave_in_block;
	const uint32_t ring = pool_wave & (rh.n_rings - 1u);
	GenCtx<W> c;
	c.k = k;
	c.lane = threadIdx.x & 63;
	c.qcap = qcap;
	c.full = qcap >> 1;
	POLR_LDS uint32_t *base = as_lds(lds + (size_t)wave_in_block * lds_per_wave);
	c.scratch = base;
	c.q = base + GEN_SCRATCH_DWORDS;
	c.sel = as_global(uniptr(pipe->sel));
	c.v_qsize = c.v_cnt_lo = c.v_cnt_hi = c.v_gstart = c.v_grem = 0;
	c.mult = uni(pipe->mult) != 0;
	c.in_pos = c.in_end = 0;
	c.out = out;
	c.emit = false;
	c.overflow = false;
	c.cur_chunk = GEN_NO_CHUNK;
	c.fill = 0;
	const POLR_CONST StageDesc *stages = (const POLR_CONST StageDesc *)uni64((uint64_t)pipe->stages);
	c.stages = stages;
	PoolUnit u;
	PoolPoller pp;
	polr_pool_poller_init(pp, run, rh.sync, ring, rh.lo_cap, rh.hi_cap, pool_wave / rh.n_rings, rh.hi_lottery, rh.idle_sleep,
	                      rh.timeout_ticks);
	// ---- work sharing ---------------------------------------------------------------------------------------------
	// A unit is a few hundred source tuples, but with repeated build keys on several joins its work is their PRODUCT: on
	// the JOB 25c shape one cast_info row meets 114 x 23 x 134 build rows, and the wave that drew it was still expanding
	// when the rest of the pass had long finished (104 ms of a 375 ms pass, one wave).  A wave that has spent
	// `share_after` steps on one unit therefore gives HALF of what it still has to do to the pool -- shallowest first:
	// the back half of its source range (an ordinary unit), else half of a pending run of build rows or the bottom half
	// of the tuples waiting in front of a stage (a CONT entry: the tuples travel through this wave's record in global
	// memory) -- and again after as many steps; whoever takes the piece does the same.  The arrival tokens are halved
	// with the work (polr_pool_device.h), the counters of every piece are added where the unit's would have been.
	const uint32_t share_stride = rh.share_stride;
	POLR_GLOBAL uint32_t *my_rec = as_global(rh.share_recs) + (size_t)pool_wave * share_stride;
	POLR_GLOBAL uint32_t *share_flags = as_global(rh.share_flags);
	const uint32_t to_ring = (ring + 1u) & (rh.n_rings - 1u);
	const uint32_t share_budget = rh.pool_waves / rh.n_rings > 0u ? rh.pool_waves / rh.n_rings : 1u;
	auto share = [&]() {
		if (u.level >= POLR_POOL_MAX_LEVEL) {
			return;
		}
		// only while the queue the piece goes to is short: pieces nobody has taken yet mean nobody is idle (and the ring
		// capacities leave room for one piece per probe wave of a ring on top of what the routers can have in flight)
		uint32_t backlog = 0;
		if (c.lane == 0) {
			const unsigned long long hh = __hip_atomic_load(&rh.sync->ctl[to_ring].hi_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const unsigned long long ht = __hip_atomic_load(&rh.sync->ctl[to_ring].hi_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			backlog = ht > hh ? (uint32_t)(ht - hh) : 0u;
		}
		if (uni(backlog) >= share_budget) {
			return;
		}
		// (a) the source: everything behind the tuple in progress
		const uint64_t from = c.in_pos + (gen_lane_get(c.v_grem, 0) ? 1u : 0u);
		if (c.in_end > from + 1u) {
			const uint32_t give = (uint32_t)((c.in_end - from) >> 1);
			c.in_end -= give;
			u.level++;
			if (c.lane == 0) {
				polr_pool_publish_shared(rh.sync, to_ring, rh.lo_cap, rh.hi_cap, POLR_POOL_KIND_WORK, u, (uint32_t)c.in_end, give,
				                         u.level);
			}
			return;
		}
		// (b, c) a stage: this wave's record must be free (the last piece it gave away has been taken)
		uint32_t busy = 0;
		if (c.lane == 0) {
			busy = __hip_atomic_load(&share_flags[pool_wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (uni(busy)) {
			return;
		}
		const bool stage_lane = c.lane < c.k;
		const uint64_t run_m = __ballot(stage_lane && c.v_grem >= 256u);
		const uint64_t q_m = __ballot(stage_lane && c.lane >= 1u && c.v_qsize >= 2u);
		if ((run_m | q_m) == 0ull) {
			return;
		}
		const uint32_t p = (uint32_t)__builtin_ctzll(run_m | q_m);
		uint32_t n = 0, gstart = 0, grem = 0, position = 0;
		if ((run_m >> p) & 1ull) {
			// (b) the second half of stage p's pending run, with the tuple it belongs to (the source front / the queue top)
			const uint32_t all = gen_lane_get(c.v_grem, p);
			grem = all >> 1;
			gstart = gen_lane_get(c.v_gstart, p) + (all - grem);
			gen_lane_set(c.v_grem, p, all - grem);
			n = 1;
			if (p == 0) {
				position = (uint32_t)c.in_pos;
			} else {
				const POLR_LDS uint32_t *qq = c.q + (size_t)(p - 1u) * W * c.qcap;
				const uint32_t top = gen_lane_get(c.v_qsize, p) - 1u;
				if (c.lane < (uint32_t)W) {
					__hip_atomic_store(&my_rec[8u + c.lane * 64u], qq[c.lane * c.qcap + top], __ATOMIC_RELAXED,
					                   __HIP_MEMORY_SCOPE_AGENT);
				}
			}
		} else {
			// (c) the bottom half of the tuples waiting in front of stage p (consumed last); the others move down
			POLR_LDS uint32_t *qq = c.q + (size_t)(p - 1u) * W * c.qcap;
			const uint32_t qs = gen_lane_get(c.v_qsize, p);
			n = qs >> 1;
			n = n > 64u ? 64u : n;
			if (c.lane < n) {
#pragma unroll
				for (int i = 0; i < W; i++) {
					__hip_atomic_store(&my_rec[8u + (uint32_t)i * 64u + c.lane], qq[i * c.qcap + c.lane], __ATOMIC_RELAXED,
					                   __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			const uint32_t left = qs - n;
			for (uint32_t b0 = 0; b0 < left; b0 += 64u) {
				const uint32_t idx = b0 + c.lane;
				uint32_t w[W];
#pragma unroll
				for (int i = 0; i < W; i++) {
					w[i] = idx < left ? qq[i * c.qcap + idx + n] : 0u;
				}
#pragma unroll
				for (int i = 0; i < W; i++) {
					if (idx < left) {
						qq[i * c.qcap + idx] = w[i];
					}
				}
			}
			gen_lane_set(c.v_qsize, p, left);
		}
		if (c.lane < 5u) {
			const uint32_t h = c.lane == 0 ? p : (c.lane == 1 ? n : (c.lane == 2 ? gstart : (c.lane == 3 ? grem : position)));
			__hip_atomic_store(&my_rec[c.lane], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		// the record is in memory before the entry that names it can be seen
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		u.level++;
		if (c.lane == 0) {
			__hip_atomic_store(&share_flags[pool_wave], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			polr_pool_publish_shared(rh.sync, to_ring, rh.lo_cap, rh.hi_cap, POLR_POOL_KIND_CONT, u, pool_wave, n, u.level);
		}
	};
	const uint32_t share_after = rh.share_recs ? rh.share_after : 0xFFFFFFFFu;
	TL_BEGIN(rh.n_router_blocks)
	while (polr_pool_next_unit(pp, u, c.lane)) {
		TL_GOT
		c.stages = stages + (size_t)u.path * POLR_KMAX;
		c.emit = u.emit != 0 && !c.overflow;
		if (u.kind == POLR_POOL_KIND_CONT) {
			// a piece of somebody's unit: u.begin names the record it waits in
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			const POLR_GLOBAL uint32_t *rec = as_global(rh.share_recs) + (size_t)u.begin * share_stride;
			uint32_t h = 0;
			if (c.lane < 5u) {
				h = __hip_atomic_load(&rec[c.lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			const uint32_t p = gen_lane_get(h, 0), n = gen_lane_get(h, 1), gstart = gen_lane_get(h, 2), grem = gen_lane_get(h, 3),
			               position = gen_lane_get(h, 4);
			c.in_pos = c.in_end = 0;
			if (grem != 0u && p == 0u) {
				c.in_pos = position;
				c.in_end = (uint64_t)position + 1u;
			} else {
				POLR_LDS uint32_t *qq = c.q + (size_t)(p - 1u) * W * c.qcap;
				if (c.lane < n) {
#pragma unroll
					for (int i = 0; i < W; i++) {
						qq[i * c.qcap + c.lane] =
						    __hip_atomic_load(&rec[8u + (uint32_t)i * 64u + c.lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
				gen_lane_set(c.v_qsize, p, n);
			}
			if (grem != 0u) {
				gen_lane_set(c.v_gstart, p, gstart);
				gen_lane_set(c.v_grem, p, grem);
			}
			// (the tuples are in LDS: the record may be written again)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			if (c.lane == 0) {
				__hip_atomic_store(&share_flags[u.begin], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		} else {
			c.in_pos = u.begin;
			c.in_end = (uint64_t)u.begin + u.count;
		}
		gen_run_unit<W>(c, share_after, share);
		TL_RUN
		poolg_arrive(execs, u, ring, k, c.v_cnt_lo, c.v_cnt_hi, c.lane, POLR_POOL_TOKENS >> u.level);
		TL_DONE(u)
	}
	if (c.cur_chunk != GEN_NO_CHUNK && c.lane == 0) {
		out.chunk_count[c.cur_chunk] = c.fill;
	}
}

// ---- launch ------------------------------------------------------------------------------------
#if POLR_EXT
#define POOLG_FN(stem) stem##x
#else
#define POOLG_FN(stem) stem
#endif

// entries per queue: as many as 8 KB of queues per wave allow, 128 .. 512 in steps of 128
static uint32_t poolg_qcap(uint32_t k, uint32_t W) {
	if (k <= 1) {
		return 128;
	}
	uint32_t q = (8192u / 4u) / ((k - 1u) * W);
	q = (q / 128u) * 128u;
	return q < 128u ? 128u : (q > 512u ? 512u : q);
}

static size_t poolg_wave_dwords(uint32_t k, uint32_t W) { // per probe wave, never less than a router needs
	const size_t probe = GEN_SCRATCH_DWORDS + (size_t)(k > 1 ? k - 1 : 0) * W * poolg_qcap(k, W);
	return probe > POOL_ROUTER_MIN_DWORDS ? probe : POOL_ROUTER_MIN_DWORDS;
}

template <int W>
static hipError_t poolg_prepare(size_t lds) {
	static size_t lds_set = 0;
	if (lds > lds_set) {
		hipError_t e = hipFuncSetAttribute((const void *)polr_pool_gen_kernel<W, POLR_EXT>, hipFuncAttributeMaxDynamicSharedMemorySize,
		                                   (int)lds);
		if (e != hipSuccess) {
			return e;
		}
		lds_set = lds;
	}
	return hipSuccess;
}

template <int W>
static int poolg_occupancy_w(size_t lds, uint32_t waves) {
	int blocks = 0;
	if (poolg_prepare<W>(lds) != hipSuccess ||
	    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *)polr_pool_gen_kernel<W, POLR_EXT>, (int)(64 * waves), lds) !=
	        hipSuccess) {
		return 0;
	}
	return blocks;
}

template <int W>
static hipError_t poolg_launch_w(dim3 grid, dim3 block, size_t lds, hipStream_t stream, const DevPipeline *pipe, const ResidentExec *execs,
                                 PoolRun *run, DevOut out, uint32_t lds_per_wave, uint32_t qcap) {
	hipError_t e = poolg_prepare<W>(lds);
	if (e != hipSuccess) {
		return e;
	}
	void *args[] = {(void *)&pipe, (void *)&execs, (void *)&run, (void *)&out, (void *)&lds_per_wave, (void *)&qcap};
	// (hipLaunchKernel returns the status of THIS launch: nothing is read from the thread's last-error slot)
	return hipLaunchKernel((const void *)polr_pool_gen_kernel<W, POLR_EXT>, grid, block, args, lds, stream);
}

#define POOLG_FOR_EACH_W(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)

// waves per workgroup: as many (8, 4, 2, 1) as leave room for two workgroups per CU; 0: one wave's queues exceed the LDS
static uint32_t poolg_waves(uint32_t k, uint32_t W) {
	const size_t per_wave = poolg_wave_dwords(k, W) * sizeof(uint32_t);
	for (uint32_t w = POOLG_WAVES; w >= 1; w >>= 1) {
		if (per_wave * w <= (w == 1 ? 160u * 1024u : 80u * 1024u)) {
			return w;
		}
	}
	return 0;
}

extern "C++" uint32_t POOLG_FN(polr_poolg_waves_per_block)(uint32_t k, uint32_t W) {
	return poolg_waves(k, W);
}

extern "C++" size_t POOLG_FN(polr_poolg_lds_bytes)(uint32_t k, uint32_t W) {
	return poolg_wave_dwords(k, W) * poolg_waves(k, W) * sizeof(uint32_t);
}

extern "C++" int POOLG_FN(polr_poolg_occupancy)(uint32_t k, uint32_t W) {
	const uint32_t waves = poolg_waves(k, W);
	const size_t lds = poolg_wave_dwords(k, W) * waves * sizeof(uint32_t);
	if (waves == 0) {
		return 0;
	}
#define OCC_CASE(N)                                                                                                    \
	if (W == N) {                                                                                                      \
		return poolg_occupancy_w<N>(lds, waves);                                                                       \
	}
	POOLG_FOR_EACH_W(OCC_CASE)
#undef OCC_CASE
	return 0;
}

extern "C++" hipError_t POOLG_FN(polr_launch_poolg_kernel)(uint32_t W, uint32_t k, uint32_t n_blocks, hipStream_t stream,
                                                            const DevPipeline *pipe, const ResidentExec *execs, PoolRun *run,
                                                            DevOut out) {
	const uint32_t per_wave = (uint32_t)poolg_wave_dwords(k, W);
	const uint32_t waves = poolg_waves(k, W);
	if (waves == 0) {
		return hipErrorInvalidValue;
	}
	const size_t lds = (size_t)per_wave * waves * sizeof(uint32_t);
	const uint32_t qcap = poolg_qcap(k, W);
	dim3 grid(n_blocks), block(64 * waves);
#define LAUNCH_CASE(N)                                                                                                 \
	if (W == N) {                                                                                                      \
		return poolg_launch_w<N>(grid, block, lds, stream, pipe, execs, run, out, per_wave, qcap);                     \
	}
	POOLG_FOR_EACH_W(LAUNCH_CASE)
#undef LAUNCH_CASE
	return hipErrorInvalidValue;
}
