// duckdb-polr_amd/csrc/polr_pool.hip -- the whole run in ONE launch (gfx950): routers + a pool of probe waves; the
// FLAT pipeline's kernel (the generic pipeline's: polr_poolg.hip).
//
// Grid: the first `n_router_blocks` workgroups host the routers (one WAVE per executor: the multiplexer of
// src/execution/operator/polr/physical_multiplexer.cpp:100-184 with its RoutingStrategy, state in LDS for the whole
// run); every other workgroup is part of the probe pool.  Protocol: polr_pool_device.h.  One kernel per compiled
// stage count K (-DPOLR_K):
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

#ifndef POLR_FLAT_EMIT
#define POLR_FLAT_EMIT 0 // two builds per K: counting runs (0) and runs whose last join writes row ids (1)
#endif

#include "polr_pool_common.h"
#include "polr_flat_device.h"

#define PASTE_TL2(a, b) a##b
#define PASTE_TL(a, b) PASTE_TL2(a, b)
#if !POLR_FLAT_EMIT
POOL_DIAG_ENTRY(PASTE_TL(polr_diag_router_k, POLR_K), PASTE_TL(polr_diag_timeline_set_k, POLR_K))
#endif

// (EMIT only tells the two builds' kernels apart by name: what differs is compiled in or out by POLR_FLAT_EMIT)
template <int K, int EMIT>
__global__ __launch_bounds__(1024) void polr_pool_flat_kernel(const DevPipeline *__restrict__ pipe,
                                                              const ResidentExec *__restrict__ execs, PoolRun *run,
                                                              DevOut out, uint32_t lds_per_wave, uint32_t table_dwords,
                                                              uint32_t router_dwords, uint32_t fused_off, uint32_t fused_words) {
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wave_in_block = threadIdx.x >> 6;
	const uint32_t k = uni(pipe->k);
	PoolRun rh;
	pool_load_run(run, rh);
	if (blockIdx.x < rh.n_router_blocks) {
		pool_router_wave(execs, run, rh, k, FLAT_STEP0, lds, router_dwords);
		return;
	}
#if POLR_FLAT_EMIT
	// a fused GROUP BY sink whose cells fit: this workgroup's cells start from zero (published by the barrier below)
	POLR_LDS unsigned long long *fused_lds = fused_words ? as_lds((unsigned long long *)(lds + fused_off)) : nullptr;
	for (uint32_t i = threadIdx.x; i < fused_words; i += blockDim.x) {
		fused_lds[i] = 0ull;
	}
#endif
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
		c.st[p].kind_lds = c.st[p].a = c.st[p].b = c.st[p].out_slot = 0;
	}
	c.out = out;
	c.emit = false;
	c.overflow = false;
	c.cur_chunk = FLAT_NO_CHUNK;
	c.fill = 0;
#if POLR_FLAT_EMIT
	c.fused_lds = fused_lds;
#else
	c.fused_lds = nullptr;
#endif
	c.unit_begin = c.in_pos = c.in_end = 0;
	c.pf_pos = ~0ull;
	c.pf0 = make_uint4(0, 0, 0, 0);
	c.pf1 = make_uint4(0, 0, 0, 0);
	const StageDesc *stages = uniptr(pipe->stages);
	uint32_t cur_path = 0xFFFFFFFFu;
	PoolUnit u, nxt;
	PoolPoller pp;
	polr_pool_poller_init(pp, run, rh.sync, ring, rh.lo_cap, rh.hi_cap, pool_wave / rh.n_rings, rh.hi_lottery, rh.idle_sleep,
	                      rh.timeout_ticks);
	TL_BEGIN(rh.n_router_blocks)
	bool have = polr_pool_next_unit(pp, u, c.lane);
	while (have) {
		TL_GOT
		if (u.path != cur_path) {
			c.pf_pos = ~0ull; // (a prefetch belongs to one join order)
#pragma unroll
			for (int p = 0; p < K; p++) {
				if (p < (int)k) {
					c.st[p] = flat_load_stage(stages + (uint64_t)u.path * POLR_KMAX + p, uni(pipe->paths[u.path].order[p]));
				}
			}
			cur_path = u.path;
		}
#if POLR_FLAT_EMIT
		c.emit = u.emit != 0 && out.ids != nullptr && !c.overflow;
#endif
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
#if POLR_FLAT_EMIT
	if (c.cur_chunk != FLAT_NO_CHUNK && c.lane == 0) {
		out.chunk_count[c.cur_chunk] = c.fill;
	}
	if (fused_words) {
		// every wave of the workgroup has left its loop: flush the cells that were touched to the workgroup's table
		__syncthreads();
		const FusedSink *f = out.fused;
		unsigned long long *table = f->cells + (size_t)(blockIdx.x % f->n_tables) * f->words_per_table;
		for (uint32_t i = threadIdx.x; i < fused_words; i += blockDim.x) {
			const unsigned long long v = fused_lds[i];
			if (v) {
				__hip_atomic_fetch_add(&table[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
#endif
}

// ---- launch ------------------------------------------------------------------------------------
#define PASTE2(a, b) a##b
#define PASTE(a, b) PASTE2(a, b)

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

static hipError_t pool_flat_prepare(size_t lds) {
	static size_t lds_set = 0;
	if (lds > lds_set) {
		hipError_t e = hipFuncSetAttribute((const void *)polr_pool_flat_kernel<POLR_K, POLR_FLAT_EMIT>,
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) {
			return e;
		}
		lds_set = lds;
	}
	return hipSuccess;
}

#if POLR_FLAT_EMIT
#define FLAT_EXPORT(name) PASTE(PASTE(name, e_k), POLR_K)
#else
#define FLAT_EXPORT(name) PASTE(PASTE(name, k), POLR_K)
#endif

#if !POLR_FLAT_EMIT
// (LDS layout and waves per workgroup are the same in both builds; both keep 128 VGPRs)
extern "C++" size_t PASTE(polr_pool_flat_lds_bytes_k, POLR_K)(uint32_t waves_per_block, uint32_t table_dwords) {
	return pool_flat_lds_dwords(waves_per_block, table_dwords) * sizeof(uint32_t);
}

extern "C++" size_t PASTE(polr_pool_flat_wave_bytes_k, POLR_K)() {
	return pool_flat_wave_dwords() * sizeof(uint32_t);
}
#endif

extern "C++" int FLAT_EXPORT(polr_pool_flat_occupancy_)(uint32_t waves_per_block, uint32_t table_dwords) {
	const size_t lds = pool_flat_lds_dwords(waves_per_block, table_dwords) * sizeof(uint32_t);
	int blocks = 0;
	if (pool_flat_prepare(lds) != hipSuccess ||
	    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *)polr_pool_flat_kernel<POLR_K, POLR_FLAT_EMIT>,
	                                                 (int)(64 * waves_per_block), lds) != hipSuccess) {
		return 0;
	}
	return blocks;
}

extern "C++" hipError_t FLAT_EXPORT(polr_launch_pool_flat_kernel_)(uint32_t n_blocks, uint32_t waves_per_block,
                                                                      uint32_t table_dwords, hipStream_t stream,
                                                                      const DevPipeline *pipe, const ResidentExec *execs,
                                                                      PoolRun *run, DevOut out, uint32_t fused_words) {
	// (fused_words: 64-bit group cells of a fused GROUP BY sink to keep in LDS behind everything else; 0: none)
	const size_t dwords = pool_flat_lds_dwords(waves_per_block, table_dwords);
	const uint32_t fused_off = (uint32_t)((dwords + 1) & ~(size_t)1);
	const size_t lds = (fused_words ? (size_t)fused_off + 2 * (size_t)fused_words : dwords) * sizeof(uint32_t);
	hipError_t e = pool_flat_prepare(lds);
	if (e != hipSuccess) {
		return e;
	}
	uint32_t per_wave = (uint32_t)pool_flat_wave_dwords();
	uint32_t router_dwords = (uint32_t)(dwords / waves_per_block);
	dim3 grid(n_blocks), block(64 * waves_per_block);
	uint32_t fused_off_arg = fused_words ? fused_off : 0u;
	void *args[] = {(void *)&pipe,         (void *)&execs,         (void *)&run,           (void *)&out,        (void *)&per_wave,
	                (void *)&table_dwords, (void *)&router_dwords, (void *)&fused_off_arg, (void *)&fused_words};
	return hipLaunchKernel((const void *)polr_pool_flat_kernel<POLR_K, POLR_FLAT_EMIT>, grid, block, args, lds, stream);
}
