// duckdb-polr_amd/csrc/polr_pool.hip -- the whole run in ONE launch (gfx950): routers + a pool of probe waves.
//
// Grid: the first `n_router_blocks` workgroups host the routers (one WAVE per executor: the multiplexer of
// src/execution/operator/polr/physical_multiplexer.cpp:100-184 with its RoutingStrategy, state in LDS for the whole
// run); every other workgroup is part of the probe pool.  Protocol: polr_pool_device.h.  Two kernels per compiled
// stage count K (-DPOLR_K):
//   polr_pool_kernel<W, K>       the generic per-wave pipeline of polr_probe_device.h (any key source, repeated
//                                keys, row-id output), 256-thread workgroups;
//   polr_pool_flat_kernel<K>     the flat pipeline of polr_flat_device.h (counting runs over banks of single-key,
//                                unique-match joins on probe columns), up to 1024-thread workgroups that share the
//                                LDS-resident bit tables.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "polr_device.h"
#include "polr_mpx_device.h"

#ifndef POLR_K
#error "compile with -DPOLR_K=<compiled stage count>"
#endif
// Two builds per K: POLR_EXT = 0 (everything; generic stages without their uncommon parts) and POLR_EXT = 1 (the generic
// kernel only, with packed composite keys and non-equality conditions -- polr_probe_device.h); exported names end in
// K resp. K'x'
#ifndef POLR_EXT
#error "compile with -DPOLR_EXT=0|1"
#endif

#ifdef POLR_DIAG_TIMELINE
static __device__ unsigned long long polr_diag_router[16];
#endif
#include "polr_probe_device.h"
#include "polr_flat_device.h"
#include "polr_pool_device.h"

#define PASTE_TL2(a, b) a##b
#define PASTE_TL(a, b) PASTE_TL2(a, b)

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
	rh.host_words = nullptr;
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
extern "C" int PASTE_TL(polr_diag_router_k, POLR_K)(unsigned long long *dst, int reset) {
	unsigned long long z[16] = {};
	if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(polr_diag_router), sizeof(z)) != hipSuccess) {
		return -1;
	}
	return reset ? (hipMemcpyToSymbol(HIP_SYMBOL(polr_diag_router), z, sizeof(z)) == hipSuccess ? 0 : -1) : 0;
}
extern "C" int PASTE_TL(polr_diag_timeline_set_k, POLR_K)(unsigned long long *buf, uint32_t cap) {
	if (hipMemcpyToSymbol(HIP_SYMBOL(polr_diag_tl), &buf, sizeof(buf)) != hipSuccess) {
		return -1;
	}
	return hipMemcpyToSymbol(HIP_SYMBOL(polr_diag_tl_cap), &cap, sizeof(cap)) == hipSuccess ? 0 : -1;
}
#else
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
		__hip_atomic_fetch_add(&sync->arrived[u.slot][ring & (POLR_POOL_SHARDS - 1u)].v, 1ull + (seen >> 63), __ATOMIC_RELAXED,
		                       __HIP_MEMORY_SCOPE_AGENT);
	}
#pragma unroll
	for (int p = 0; p < K; p++) {
		cnt[p] = 0;
	}
}

template <int W, int K, int EXT>
__global__ __launch_bounds__(256, 4) void polr_pool_kernel(const DevPipeline *__restrict__ pipe,
                                                           const ResidentExec *__restrict__ execs, PoolRun *run,
                                                           DevOut out, uint32_t lds_per_wave) {
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wave_in_block = threadIdx.x >> 6;
	const uint32_t k = uni(pipe->k);
	PoolRun rh;
	pool_load_run(run, rh);
	if (blockIdx.x < rh.n_router_blocks) {
		pool_router_wave(execs, run, rh, k, 256, lds, lds_per_wave);
		return;
	}
	// probe wave g of the pool serves ring g % n_rings (dealt wave by wave, not workgroup by workgroup: the units of a
	// round go to all rings alike, so every ring needs the same number of waves -- 240 workgroups over 64 rings left a
	// quarter of the rings with 3 workgroups instead of 4, and every round waited for those)
	const uint32_t pool_wave = (blockIdx.x - rh.n_router_blocks) * (blockDim.x >> 6) + wave_in_block;
	const uint32_t ring = pool_wave & (rh.n_rings - 1u);
	WaveCtx<W, K> c;
	c.k = k;
	c.lane = threadIdx.x & 63;
	uint32_t *base = lds + (size_t)wave_in_block * lds_per_wave;
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
	uint32_t cur_path = 0xFFFFFFFFu;
	PoolUnit u;
	PoolPoller pp;
	polr_pool_poller_init(pp, run, rh.sync, ring, rh.lo_cap, rh.hi_cap, pool_wave / rh.n_rings, rh.hi_lottery, rh.idle_sleep);
	TL_BEGIN(rh.n_router_blocks)
	while (polr_pool_next_unit(pp, u, c.lane)) {
		TL_GOT
		if (u.path != cur_path) {
			const uint32_t *src = (const uint32_t *)(stages + (uint64_t)u.path * POLR_KMAX);
			uint32_t *dst = (uint32_t *)c.desc;
			for (uint32_t i = c.lane; i < K * STAGE_DESC_DWORDS; i += 64) {
				dst[i] = src[i];
			}
			c.wide_mask = stage_wide_mask<W, K>(src, c.k);
			cur_path = u.path;
		}
		c.emit = u.emit != 0 && !c.overflow;
		c.in_pos = u.begin;
		c.in_end = (uint64_t)u.begin + u.count;
		run_until_idle(c, false);
		run_until_idle(c, true);
		TL_RUN
		pool_arrive<K>(execs, u, ring, c.k, c.cnt, c.lane);
		TL_DONE(u)
	}
	if (c.cur_chunk != NO_CHUNK && c.lane == 0) {
		out.chunk_count[c.cur_chunk] = c.fill;
	}
}

#if !POLR_EXT
template <int K>
__global__ __launch_bounds__(1024) void polr_pool_flat_kernel(const DevPipeline *__restrict__ pipe,
                                                              const ResidentExec *__restrict__ execs, PoolRun *run,
                                                              uint32_t lds_per_wave, uint32_t table_dwords,
                                                              uint32_t router_dwords) {
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wave_in_block = threadIdx.x >> 6;
	const uint32_t k = uni(pipe->k);
	PoolRun rh;
	pool_load_run(run, rh);
	if (blockIdx.x < rh.n_router_blocks) {
		pool_router_wave(execs, run, rh, k, FLAT_STEP0, lds, router_dwords);
		return;
	}
	// the bit tables that fit stay in LDS for the whole run: one cooperative copy per workgroup
	{
		const uint32_t n_tab = uni(pipe->n_lds_tables);
		for (uint32_t t = 0; t < n_tab; t++) {
			const uint32_t *src = uniptr(pipe->lds_table_src[t]);
			const uint32_t off = uni(pipe->lds_table_off[t]);
			const uint32_t len = uni(pipe->lds_table_len[t]);
			for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
				lds[off + i] = src[i];
			}
		}
		__syncthreads();
	}
	// probe wave g of the pool serves ring g % n_rings (dealt wave by wave, not workgroup by workgroup: the units of a
	// round go to all rings alike, so every ring needs the same number of waves -- 240 workgroups over 64 rings left a
	// quarter of the rings with 3 workgroups instead of 4, and every round waited for those)
	const uint32_t pool_wave = (blockIdx.x - rh.n_router_blocks) * (blockDim.x >> 6) + wave_in_block;
	const uint32_t ring = pool_wave & (rh.n_rings - 1u);
	FlatCtx<K> c;
	c.k = k;
	c.lane = threadIdx.x & 63;
	c.sel = as_global(uniptr(pipe->sel));
	c.lds_tables = as_lds((const uint32_t *)lds);
	c.q = as_lds((uint16_t *)(lds + table_dwords + (size_t)wave_in_block * lds_per_wave));
#pragma unroll
	for (int p = 0; p < K; p++) {
		c.qsize[p] = c.cnt[p] = 0;
		c.st[p].keys = nullptr;
		c.st[p].valid = nullptr;
		c.st[p].table = nullptr;
		c.st[p].kind_lds = c.st[p].a = c.st[p].b = 0;
	}
	c.unit_begin = c.in_pos = c.in_end = 0;
	c.pf_pos = ~0ull;
	c.pf0 = make_uint4(0, 0, 0, 0);
	c.pf1 = make_uint4(0, 0, 0, 0);
	const StageDesc *stages = uniptr(pipe->stages);
	uint32_t cur_path = 0xFFFFFFFFu;
	PoolUnit u, nxt;
	PoolPoller pp;
	polr_pool_poller_init(pp, run, rh.sync, ring, rh.lo_cap, rh.hi_cap, pool_wave / rh.n_rings, rh.hi_lottery, rh.idle_sleep);
	TL_BEGIN(rh.n_router_blocks)
	bool have = polr_pool_next_unit(pp, u, c.lane);
	while (have) {
		TL_GOT
		if (u.path != cur_path) {
			c.pf_pos = ~0ull; // (a prefetch belongs to one join order)
#pragma unroll
			for (int p = 0; p < K; p++) {
				if (p < (int)k) {
					c.st[p] = flat_load_stage(stages + (uint64_t)u.path * POLR_KMAX + p);
				}
			}
			cur_path = u.path;
		}
		c.unit_begin = u.begin;
		c.in_pos = u.begin;
		c.in_end = (uint64_t)u.begin + u.count;
		if (c.pf_pos != c.in_pos) {
			c.pf_pos = ~0ull;
		}
		flat_run_source<K>(c);
		// the source of this unit is used up: ONE look for the next unit before the queues drain, so that its claim and
		// the first keys of its source travel while this unit's last sweeps do (a unit's fixed cost is mostly dependent
		// round trips: measured 12.5 us + 6.3 ns per tuple)
		const uint32_t peek = polr_pool_poll(pp, nxt, c.lane);
		if (peek == POLR_POLL_WORK && nxt.path == cur_path) {
			flat_prefetch_unit<K>(c, nxt.begin, nxt.count);
		}
		flat_run_drain<K>(c);
		TL_RUN
		pool_arrive<K>(execs, u, ring, c.k, c.cnt, c.lane);
		TL_DONE(u)
		if (peek == POLR_POLL_NONE) {
			have = polr_pool_next_unit(pp, nxt, c.lane);
		} else {
			have = peek == POLR_POLL_WORK;
		}
		u = nxt;
	}
}

#endif // !POLR_EXT
// ---- launch ------------------------------------------------------------------------------------
#define PASTE2(a, b) a##b
#define PASTE(a, b) PASTE2(a, b)
#if POLR_EXT
#define POOLFN(stem) PASTE(PASTE(stem, POLR_K), x)
#else
#define POOLFN(stem) PASTE(stem, POLR_K)
#endif

static size_t pool_wave_dwords(uint32_t W) { // per probe wave of the generic kernel, never less than a router needs
	const size_t queues = POLR_K <= 1 ? 0 : (size_t)W * QCAP1 + (size_t)(POLR_K - 2) * W * QCAPN;
	const size_t wslots = (POLR_K <= 4 && W <= 4) ? 2 : 1; // (wide_pend_slots<W, K>())
	const size_t probe = (size_t)POLR_K * STAGE_DESC_DWORDS + queues + (size_t)POLR_K * 64 * 2 + 64 * WIDE + wslots * 64 * WIDE * 2;
	return probe > POOL_ROUTER_MIN_DWORDS ? probe : POOL_ROUTER_MIN_DWORDS;
}

#if !POLR_EXT
static size_t pool_flat_wave_dwords() {
	return (size_t)flat_per_wave_dwords<POLR_K>();
}

// dynamic LDS of the flat kernel in dwords: tables + probe queues, never less than its router waves need (router
// wave r of a router workgroup uses [r * stride, (r + 1) * stride) with stride = total / waves)
static size_t pool_flat_lds_dwords(uint32_t waves_per_block, uint32_t table_dwords) {
	const size_t probe = table_dwords + pool_flat_wave_dwords() * waves_per_block;
	const size_t router = (size_t)POOL_ROUTER_MIN_DWORDS * waves_per_block;
	return probe > router ? probe : router;
}
#endif

template <int W>
static hipError_t pool_prepare(size_t lds) {
	static size_t lds_set = 0;
	if (lds > lds_set) {
		hipError_t e = hipFuncSetAttribute((const void *)polr_pool_kernel<W, POLR_K, POLR_EXT>,
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) {
			return e;
		}
		lds_set = lds;
	}
	return hipSuccess;
}

#if !POLR_EXT
static hipError_t pool_flat_prepare(size_t lds) {
	static size_t lds_set = 0;
	if (lds > lds_set) {
		hipError_t e = hipFuncSetAttribute((const void *)polr_pool_flat_kernel<POLR_K>,
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) {
			return e;
		}
		lds_set = lds;
	}
	return hipSuccess;
}
#endif

template <int W>
static int pool_occupancy_w(size_t lds, uint32_t threads) {
	int blocks = 0;
	if (pool_prepare<W>(lds) != hipSuccess ||
	    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *)polr_pool_kernel<W, POLR_K, POLR_EXT>, (int)threads,
	                                                 lds) != hipSuccess) {
		return 0;
	}
	return blocks;
}

template <int W>
static hipError_t pool_launch_w(dim3 grid, dim3 block, size_t lds, hipStream_t stream, const DevPipeline *pipe,
                                const ResidentExec *execs, PoolRun *run, DevOut out, uint32_t lds_per_wave) {
	hipError_t e = pool_prepare<W>(lds);
	if (e != hipSuccess) {
		return e;
	}
	void *args[] = {(void *)&pipe, (void *)&execs, (void *)&run, (void *)&out, (void *)&lds_per_wave};
	// (hipLaunchKernel returns the status of THIS launch: nothing is read from the thread's last-error slot)
	return hipLaunchKernel((const void *)polr_pool_kernel<W, POLR_K, POLR_EXT>, grid, block, args, lds, stream);
}

template <int N>
struct PoolWc {
	static constexpr int v = (N <= POLR_K + 1) ? N : 1;
};
#define POLR_FOR_EACH_W(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)

extern "C++" size_t POOLFN(polr_pool_lds_bytes_k)(uint32_t W, uint32_t waves_per_block) {
	return pool_wave_dwords(W) * waves_per_block * sizeof(uint32_t);
}

extern "C++" int POOLFN(polr_pool_occupancy_k)(uint32_t W, uint32_t waves_per_block) {
	const size_t lds = pool_wave_dwords(W) * waves_per_block * sizeof(uint32_t);
	if (W < 1 || W > POLR_K + 1) {
		return 0;
	}
#define OCC_CASE(N)                                                                                                    \
	if (W == N) {                                                                                                      \
		return pool_occupancy_w<PoolWc<N>::v>(lds, 64 * waves_per_block);                                              \
	}
	POLR_FOR_EACH_W(OCC_CASE)
#undef OCC_CASE
	return 0;
}

extern "C++" hipError_t POOLFN(polr_launch_pool_kernel_k)(uint32_t W, uint32_t n_blocks, uint32_t waves_per_block,
                                                                 hipStream_t stream, const DevPipeline *pipe,
                                                                 const ResidentExec *execs, PoolRun *run, DevOut out) {
	const uint32_t per_wave = (uint32_t)pool_wave_dwords(W);
	const size_t lds = (size_t)per_wave * waves_per_block * sizeof(uint32_t);
	dim3 grid(n_blocks), block(64 * waves_per_block);
	if (W < 1 || W > POLR_K + 1) {
		return hipErrorInvalidValue;
	}
#define LAUNCH_CASE(N)                                                                                                 \
	if (W == N) {                                                                                                      \
		return pool_launch_w<PoolWc<N>::v>(grid, block, lds, stream, pipe, execs, run, out, per_wave);                 \
	}
	POLR_FOR_EACH_W(LAUNCH_CASE)
#undef LAUNCH_CASE
	return hipErrorInvalidValue;
}

#if !POLR_EXT
extern "C++" size_t PASTE(polr_pool_flat_lds_bytes_k, POLR_K)(uint32_t waves_per_block, uint32_t table_dwords) {
	return pool_flat_lds_dwords(waves_per_block, table_dwords) * sizeof(uint32_t);
}

extern "C++" size_t PASTE(polr_pool_flat_wave_bytes_k, POLR_K)() {
	return pool_flat_wave_dwords() * sizeof(uint32_t);
}

extern "C++" int PASTE(polr_pool_flat_occupancy_k, POLR_K)(uint32_t waves_per_block, uint32_t table_dwords) {
	const size_t lds = pool_flat_lds_dwords(waves_per_block, table_dwords) * sizeof(uint32_t);
	int blocks = 0;
	if (pool_flat_prepare(lds) != hipSuccess ||
	    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *)polr_pool_flat_kernel<POLR_K>,
	                                                 (int)(64 * waves_per_block), lds) != hipSuccess) {
		return 0;
	}
	return blocks;
}

extern "C++" hipError_t PASTE(polr_launch_pool_flat_kernel_k, POLR_K)(uint32_t n_blocks, uint32_t waves_per_block,
                                                                      uint32_t table_dwords, hipStream_t stream,
                                                                      const DevPipeline *pipe, const ResidentExec *execs,
                                                                      PoolRun *run) {
	const size_t dwords = pool_flat_lds_dwords(waves_per_block, table_dwords);
	const size_t lds = dwords * sizeof(uint32_t);
	hipError_t e = pool_flat_prepare(lds);
	if (e != hipSuccess) {
		return e;
	}
	uint32_t per_wave = (uint32_t)pool_flat_wave_dwords();
	uint32_t router_dwords = (uint32_t)(dwords / waves_per_block);
	dim3 grid(n_blocks), block(64 * waves_per_block);
	void *args[] = {(void *)&pipe, (void *)&execs, (void *)&run, (void *)&per_wave, (void *)&table_dwords,
	                (void *)&router_dwords};
	return hipLaunchKernel((const void *)polr_pool_flat_kernel<POLR_K>, grid, block, args, lds, stream);
}
#endif // !POLR_EXT
