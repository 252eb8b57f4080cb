// duckdb-polr_amd/csrc/polr_pool_common.h -- what the two pool kernels (flat pipeline: polr_pool.hip, generic pipeline:
// polr_poolg.hip) share on the device: loading the run header and an executor's descriptor into scalar registers, the
// router waves of a router workgroup, a probe wave's arrival, and the timeline records of the diagnostic build.
// The including file defines POOL_DIAG_SUFFIX (name suffix of the diagnostic entry points of ITS kernels).
#pragma once

#ifdef POLR_DIAG_TIMELINE
static __device__ unsigned long long polr_diag_router[16];
#endif
#include "polr_pool_device.h"

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

// LDS a router wave needs: state + round scratch | saved state of a rehearsal | window of >= 256 chunk boundaries
#define POOL_ROUTER_STATE ((POLR_RES_ROUTER_DWORDS + 3u) & ~3u)
#define POOL_ROUTER_SAVE ((POLR_RES_HOT_DWORDS + 3u) & ~3u)
#define POOL_ROUTER_MIN_DWORDS (POOL_ROUTER_STATE + POOL_ROUTER_SAVE + 2u * 256u)

__device__ __forceinline__ void pool_load_run(const PoolRun *run, PoolRun &rh) {
	rh.sync = (PoolSync *)uni64((uint64_t)run->sync);
	rh.n_exec = uni(run->n_exec);
	rh.n_router_blocks = uni(run->n_router_blocks);
	rh.pool_waves = uni(run->pool_waves);
	rh.lo_cap = uni(run->lo_cap);
	rh.hi_cap = uni(run->hi_cap);
	rh.hi_tuples = uni(run->hi_tuples);
	rh.n_rings = uni(run->n_rings);
	rh.units_x = uni(run->units_x);
	rh.hi_unit = uni(run->hi_unit);
	rh.hi_lottery = uni(run->hi_lottery);
	rh.idle_sleep = uni(run->idle_sleep);
	rh.routers_done = 0;
	rh.abort = 0;
	rh.host_words = (volatile uint32_t *)uni64((uint64_t)run->host_words);
	rh.timeout_ticks = uni64(run->timeout_ticks);
	rh.share_recs = (uint32_t *)uni64((uint64_t)run->share_recs);
	rh.share_flags = (uint32_t *)uni64((uint64_t)run->share_flags);
	rh.share_stride = uni(run->share_stride);
	rh.share_after = uni(run->share_after);
}

__device__ __forceinline__ void pool_load_exec(const ResidentExec *xp, ResidentExec &x) {
	x.mpx = (DevMpx *)uni64((uint64_t)xp->mpx);
	x.sync = (ResidentSync *)uni64((uint64_t)xp->sync);
	x.counts = (unsigned long long *)uni64((uint64_t)xp->counts);
	x.chunk_begin = uni64(xp->chunk_begin);
	x.chunk_end = uni64(xp->chunk_end);
	x.chunk_offsets = (const uint64_t *)uni64((uint64_t)xp->chunk_offsets);
	x.n_chunks = uni64(xp->n_chunks);
	x.n_tuples = uni64(xp->n_tuples);
	x.flags = uni(xp->flags);
	x.pad = 0;
	x.stats_out = (polr_mpx_stats *)uni64((uint64_t)xp->stats_out);
	x.morsel_cursor = (unsigned long long *)uni64((uint64_t)xp->morsel_cursor);
	x.morsel_end = uni64(xp->morsel_end);
	x.morsel_chunks = uni(xp->morsel_chunks);
	x.path_plus1 = uni(xp->path_plus1);
	x.n_more = uni(xp->n_more);
	x.pad2 = 0;
#pragma unroll
	for (int j = 0; j < POLR_MORE_RANGES; j++) {
		x.more_begin[j] = uni64(xp->more_begin[j]);
		x.more_end[j] = uni64(xp->more_end[j]);
	}
}

// the router waves of a router workgroup; router_dwords: LDS dwords per router wave
__device__ __forceinline__ void pool_router_wave(const ResidentExec *execs, PoolRun *run, const PoolRun &rh, uint32_t k,
                                                 uint32_t gran, uint32_t *lds, uint32_t router_dwords) {
	const uint32_t wave_in_block = threadIdx.x >> 6;
	const uint32_t wpb = blockDim.x >> 6;
	const uint32_t exec = blockIdx.x * wpb + wave_in_block;
	if (exec >= rh.n_exec) {
		return;
	}
	// a router is one wave of mostly scalar-style, dependent code that everybody else waits for: it gets the SIMD's
	// issue slots ahead of the probe waves it shares the SIMD with
	__builtin_amdgcn_s_setprio(3);
	ResidentExec x;
	pool_load_exec(execs + exec, x);
	uint32_t *base = lds + (size_t)wave_in_block * router_dwords;
	const uint32_t cache_dwords = router_dwords - POOL_ROUTER_STATE - POOL_ROUTER_SAVE;
	polr_pool_router(x, run, rh, exec, k, gran, threadIdx.x & 63, base, (uint64_t *)(base + POOL_ROUTER_STATE + POOL_ROUTER_SAVE),
	                 cache_dwords / 2, base + POOL_ROUTER_STATE);
}

// diagnostic build only (-DPOLR_DIAG_TIMELINE, `make diag`; never compiled into the product): every probe wave writes
// {began waiting, got the unit, finished it, exec << 40 | path << 32 | count} per unit, 100 MHz wall clock
#ifdef POLR_DIAG_TIMELINE
static __device__ unsigned long long *polr_diag_tl;
static __device__ uint32_t polr_diag_tl_cap;
#define TL_BEGIN(first_block_)                                                                                         \
	unsigned long long tl_wait = wall_clock64();                                                                       \
	unsigned long long tl_got = 0, tl_run = 0;                                                                         \
	uint32_t tl_n = 0;                                                                                                 \
	const uint32_t tl_wave = (blockIdx.x - (first_block_)) * (blockDim.x >> 6) + (threadIdx.x >> 6);
#define TL_GOT tl_got = wall_clock64();
#define TL_RUN tl_run = wall_clock64();
#define TL_DONE(u_)                                                                                                    \
	if (polr_diag_tl && tl_n < polr_diag_tl_cap && (threadIdx.x & 63) == 0) {                                          \
		unsigned long long *r_ = polr_diag_tl + ((size_t)tl_wave * polr_diag_tl_cap + tl_n) * 4;                       \
		r_[0] = tl_wait;                                                                                               \
		r_[1] = tl_got;                                                                                                \
		r_[2] = wall_clock64();                                                                                        \
		const unsigned long long q_ = (tl_run - tl_got) / 25ull; /* quarter microseconds spent probing, 8 bits */      \
		r_[3] = ((q_ > 255ull ? 255ull : q_) << 56) | ((unsigned long long)((u_).exec & 0xFFFFu) << 40) |              \
		        ((unsigned long long)(u_).path << 32) | (u_).count;                                                    \
	}                                                                                                                  \
	tl_n++;                                                                                                            \
	tl_wait = wall_clock64();
// the host side of the diagnostic build: read (and reset) the routers' phase sums, install the timeline buffer
#define POOL_DIAG_ENTRY(router_name_, timeline_name_)                                                                  \
	extern "C" int router_name_(unsigned long long *dst, int reset) {                                                  \
		unsigned long long z[16] = {};                                                                                 \
		if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(polr_diag_router), sizeof(z)) != hipSuccess) {                         \
			return -1;                                                                                                 \
		}                                                                                                              \
		return reset ? (hipMemcpyToSymbol(HIP_SYMBOL(polr_diag_router), z, sizeof(z)) == hipSuccess ? 0 : -1) : 0;      \
	}                                                                                                                  \
	extern "C" int timeline_name_(unsigned long long *buf, uint32_t cap) {                                             \
		if (hipMemcpyToSymbol(HIP_SYMBOL(polr_diag_tl), &buf, sizeof(buf)) != hipSuccess) {                            \
			return -1;                                                                                                 \
		}                                                                                                              \
		return hipMemcpyToSymbol(HIP_SYMBOL(polr_diag_tl_cap), &cap, sizeof(cap)) == hipSuccess ? 0 : -1;              \
	}
#else
#define POOL_DIAG_ENTRY(router_name_, timeline_name_)
#define TL_BEGIN(first_block_)
#define TL_GOT
#define TL_RUN
#define TL_DONE(u_)
#endif

// a probe wave reports a finished unit: its stage counters (returning atomics), then the arrival
template <int K>
__device__ __forceinline__ void pool_arrive(const ResidentExec *execs, const PoolUnit &u, uint32_t ring, uint32_t k,
                                            uint32_t (&cnt)[K], uint32_t lane) {
	const POLR_GLOBAL ResidentExec *xp = as_global(execs) + u.exec;
	POLR_GLOBAL unsigned long long *bank = as_global((unsigned long long *)uni64((uint64_t)xp->counts)) +
	                                       (size_t)u.slot * POLR_NSHARD * POLR_KMAX +
	                                       (size_t)(ring & (POLR_POOL_SHARDS - 1u)) * POLR_KMAX;
	POLR_GLOBAL ResidentSync *sync = as_global((ResidentSync *)uni64((uint64_t)xp->sync));
	if (lane == 0) {
		unsigned long long seen = 0;
#pragma unroll
		for (int p = 0; p < K; p++) {
			if (p < (int)k && cnt[p]) {
				// returning form: the arrival below consumes `seen`, so it is issued after the adds have been performed
				seen |= __hip_atomic_fetch_add(&bank[p], (unsigned long long)cnt[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		__hip_atomic_fetch_add(&sync->arrived[u.slot][ring & (POLR_POOL_SHARDS - 1u)].v, POLR_POOL_TOKENS + (seen >> 63), __ATOMIC_RELAXED,
		                       __HIP_MEMORY_SCOPE_AGENT);
	}
#pragma unroll
	for (int p = 0; p < K; p++) {
		cnt[p] = 0;
	}
}

