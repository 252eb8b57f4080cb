// duckdb-polr_amd/csrc/polr_agg.hip -- the sink side of the POLAR pipeline on the device.
//
// Reference: PhysicalUngroupedAggregate (src/execution/operator/aggregate/physical_ungrouped_aggregate.cpp)
// folds every chunk the pipeline produces into one state per aggregate: COUNT(*) counts rows
// (src/function/aggregate/distributive/count.cpp), COUNT(x) the non-NULL x, SUM(x) adds the non-NULL x exactly
// in a HUGEINT (sum.cpp:113-144 SumToHugeintOperation; NULL when nothing was added, sum_helpers.hpp isset),
// MIN / MAX keep the extreme of the non-NULL x (minmax.cpp; NULL when there was none).
//
// Here the pipeline's result is a stream of row-id chunks (polr_out); the aggregate never materialises a
// column: one pass gathers x by row id and reduces it -- thread-local, wave shuffle, one partial per
// workgroup -- and only the per-workgroup partials (a few KB) leave the device; the host adds them up in
// 128-bit arithmetic.  Algorithmic bytes: 4 (row id) + width [+ 1 validity] per output row and aggregate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "polr_internal.h"

struct AggPartial {
	unsigned long long sum_lo; // two's complement 128-bit sum, low limb
	long long sum_hi;
	long long mn, mx;
	unsigned long long count; // rows that took part (non-NULL; all rows for COUNT(*))
};

__device__ __forceinline__ long long agg_cell(const DevCol &c, uint32_t row) {
	const uint8_t *p = c.data + (uint64_t)row * c.width;
	const bool sx = (c.flags & 1u) != 0;
	switch (c.width) {
	case 1:
		return sx ? (long long)*(const int8_t *)p : (long long)*p;
	case 2:
		return sx ? (long long)*(const int16_t *)p : (long long)*(const uint16_t *)p;
	case 4:
		return sx ? (long long)*(const int32_t *)p : (long long)*(const uint32_t *)p;
	default:
		return *(const long long *)p;
	}
}

// grid-stride over the output chunks; partials[blockIdx.x * n + a]
__global__ __launch_bounds__(256) void polr_agg_kernel(DevOut out, uint32_t n_chunks, DevAggSet aggs,
                                                       AggPartial *__restrict__ partials) {
	__shared__ AggPartial wave_part[4][POLR_MAX_AGGS];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (uint32_t a = 0; a < aggs.n; a++) {
		const DevAgg ag = aggs.a[a];
		unsigned long long lo = 0, cnt = 0;
		long long hi = 0, mn = 0x7FFFFFFFFFFFFFFFll, mx = (long long)0x8000000000000000ull;
		for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
			const uint32_t n = out.chunk_count[chunk];
			const uint32_t *ids = out.ids + (uint64_t)ag.slot * out.slot_stride + (uint64_t)chunk * out.chunk_capacity;
			for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
				if (ag.fn == POLR_AGG_COUNT_STAR) {
					cnt++;
					continue;
				}
				const uint32_t row = ids[i];
				if (ag.src.valid && !ag.src.valid[row]) {
					continue; // NULLs take no part
				}
				const long long v = agg_cell(ag.src, row);
				cnt++;
				const unsigned long long nl = lo + (unsigned long long)v;
				hi += (v < 0 ? -1 : 0) + (nl < lo ? 1 : 0); // sign extension of v + carry
				lo = nl;
				mn = v < mn ? v : mn;
				mx = v > mx ? v : mx;
			}
		}
		// wave reduction (128-bit add with carry, min, max, count)
		for (int d = 32; d > 0; d >>= 1) {
			const unsigned long long olo = __shfl_down(lo, d, 64);
			const long long ohi = __shfl_down(hi, d, 64);
			const long long omn = __shfl_down(mn, d, 64), omx = __shfl_down(mx, d, 64);
			const unsigned long long ocnt = __shfl_down(cnt, d, 64);
			const unsigned long long nl = lo + olo;
			hi += ohi + (nl < lo ? 1 : 0);
			lo = nl;
			mn = omn < mn ? omn : mn;
			mx = omx > mx ? omx : mx;
			cnt += ocnt;
		}
		if (lane == 0) {
			wave_part[wave][a].sum_lo = lo;
			wave_part[wave][a].sum_hi = hi;
			wave_part[wave][a].mn = mn;
			wave_part[wave][a].mx = mx;
			wave_part[wave][a].count = cnt;
		}
	}
	__syncthreads();
	if (threadIdx.x < aggs.n) {
		const uint32_t a = threadIdx.x;
		AggPartial r = wave_part[0][a];
		for (uint32_t w = 1; w < (blockDim.x >> 6); w++) {
			const AggPartial o = wave_part[w][a];
			const unsigned long long nl = r.sum_lo + o.sum_lo;
			r.sum_hi += o.sum_hi + (nl < r.sum_lo ? 1 : 0);
			r.sum_lo = nl;
			r.mn = o.mn < r.mn ? o.mn : r.mn;
			r.mx = o.mx > r.mx ? o.mx : r.mx;
			r.count += o.count;
		}
		partials[(uint64_t)blockIdx.x * aggs.n + a] = r;
	}
}

extern "C" {

int polr_out_aggregate(polr_out *o, void *stream, const polr_agg_spec *specs, uint32_t n_aggs,
                       polr_agg_value *results) {
	POLR_ENTRY();
	if (!o || !specs || !results || n_aggs == 0) {
		return POLR_E_INVALID;
	}
	polr_pipeline *p = o->pipe;
	polr_ctx *ctx = p->ctx;
	if (n_aggs > POLR_MAX_AGGS) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "at most %d aggregates per call", POLR_MAX_AGGS);
	}
	DevAggSet set;
	memset(&set, 0, sizeof(set));
	set.n = n_aggs;
	for (uint32_t a = 0; a < n_aggs; a++) {
		const polr_agg_spec &s = specs[a];
		if (s.fn > POLR_AGG_MAX) {
			POLR_FAIL(ctx, POLR_E_INVALID, "aggregate %u: unknown function %u", a, s.fn);
		}
		set.a[a].fn = s.fn;
		if (s.fn == POLR_AGG_COUNT_STAR) {
			continue;
		}
		const OwnedCol *c;
		if (s.src_join < 0) {
			if (s.src_col >= p->n_probe_cols) {
				POLR_FAIL(ctx, POLR_E_INVALID, "aggregate %u: probe column %u out of range", a, s.src_col);
			}
			c = &p->probe_cols[s.src_col];
			set.a[a].slot = 0;
		} else {
			if ((uint32_t)s.src_join >= p->k || s.src_col >= p->hts[s.src_join]->n_payload) {
				POLR_FAIL(ctx, POLR_E_INVALID, "aggregate %u: build column (%d,%u) out of range", a, s.src_join, s.src_col);
			}
			const polr_ht *ht = p->hts[s.src_join];
			c = ht->kind == KIND_PERFECT ? &ht->pcols[s.src_col] : &ht->payload[s.src_col];
			set.a[a].slot = 1 + (uint32_t)s.src_join;
		}
		if (c->width > 8) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "aggregate %u: only integer columns of up to 8 bytes", a);
		}
		if (c->width == 8 && !(c->flags & 1u)) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "aggregate %u: unsigned 64-bit column", a);
		}
		set.a[a].src.data = c->data;
		set.a[a].src.valid = c->valid;
		set.a[a].src.width = c->width;
		set.a[a].src.flags = c->flags;
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	if (!o->stats_valid) {
		int rc = polr_out_stats(o, stream, nullptr, nullptr, nullptr);
		if (rc) {
			return rc;
		}
	}
	const uint32_t n_blocks = std::max<uint32_t>(1, std::min<uint32_t>(o->n_chunks, (uint32_t)ctx->n_cus * 8));
	std::vector<AggPartial> host((size_t)n_blocks * n_aggs);
	if (o->n_chunks) {
		AggPartial *part = nullptr;
		HIPCHK(ctx, hipMalloc((void **)&part, host.size() * sizeof(AggPartial)));
		hipLaunchKernelGGL(polr_agg_kernel, dim3(n_blocks), dim3(256), 0, st, o->dev, o->n_chunks, set, part);
		hipError_t e = hipMemcpyAsync(host.data(), part, host.size() * sizeof(AggPartial), hipMemcpyDeviceToHost, st);
		e = e == hipSuccess ? hipStreamSynchronize(st) : e;
		hipFree(part);
		if (e != hipSuccess) {
			POLR_FAIL(ctx, POLR_E_HIP, "aggregate failed: %s", hipGetErrorString(e));
		}
	}
	for (uint32_t a = 0; a < n_aggs; a++) {
		__int128 sum = 0;
		long long mn = 0x7FFFFFFFFFFFFFFFll, mx = (long long)0x8000000000000000ull;
		unsigned long long cnt = 0;
		for (uint32_t b = 0; b < n_blocks && o->n_chunks; b++) {
			const AggPartial &r = host[(size_t)b * n_aggs + a];
			sum += ((__int128)r.sum_hi << 64) + (__int128)r.sum_lo;
			mn = r.mn < mn ? r.mn : mn;
			mx = r.mx > mx ? r.mx : mx;
			cnt += r.count;
		}
		polr_agg_value &v = results[a];
		memset(&v, 0, sizeof(v));
		v.count = cnt;
		switch (specs[a].fn) {
		case POLR_AGG_COUNT_STAR:
		case POLR_AGG_COUNT:
			v.lo = (int64_t)cnt; // COUNT is never NULL
			break;
		case POLR_AGG_SUM:
			v.is_null = cnt == 0;
			v.lo = (int64_t)(unsigned long long)sum;
			v.hi = (int64_t)(sum >> 64);
			break;
		case POLR_AGG_MIN:
			v.is_null = cnt == 0;
			v.lo = cnt ? mn : 0;
			v.hi = (cnt && mn < 0) ? -1 : 0;
			break;
		default:
			v.is_null = cnt == 0;
			v.lo = cnt ? mx : 0;
			v.hi = (cnt && mx < 0) ? -1 : 0;
			break;
		}
	}
	return POLR_OK;
}

} // extern "C"

// ---- grouped aggregate over small dense group domains (PhysicalPerfectHashAggregate's case:
// src/execution/operator/aggregate/physical_perfecthash_aggregate.cpp -- every group column has a small known
// [min, max] range, the group index is the mixed-radix number of the key offsets).  SSB Q4.x:
// GROUP BY d_year, c_nation = 7 x 25 groups.
// One accumulator cell per (group, aggregate): the sum is kept as two 64-bit limbs -- the sum of the values' low 32
// bits and the sum of their (signed) high parts -- so that plain 64-bit atomic adds stay exact for up to 2^32 rows;
// workgroups accumulate in LDS and flush their non-empty cells to the global table once.
struct GroupCell {
	unsigned long long lo32; // sum of (v & 0xFFFFFFFF)
	long long hi32;          // sum of (v >> 32), arithmetic
	long long mn, mx;
	unsigned long long count;
};

#define POLR_GROUP_LDS_CELLS 1024

__device__ __forceinline__ void cell_add(GroupCell *c, long long v) {
	atomicAdd(&c->lo32, (unsigned long long)((unsigned long long)v & 0xFFFFFFFFull));
	atomicAdd((unsigned long long *)&c->hi32, (unsigned long long)(v >> 32));
	atomicMin(&c->mn, v);
	atomicMax(&c->mx, v);
	atomicAdd(&c->count, 1ull);
}

__global__ __launch_bounds__(256) void polr_group_init_kernel(GroupCell *cells, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		cells[i].lo32 = 0;
		cells[i].hi32 = 0;
		cells[i].mn = 0x7FFFFFFFFFFFFFFFll;
		cells[i].mx = (long long)0x8000000000000000ull;
		cells[i].count = 0;
	}
}

__global__ __launch_bounds__(256) void polr_group_agg_kernel(DevOut out, uint32_t n_chunks, DevGroupSet groups,
                                                             DevAggSet aggs, GroupCell *__restrict__ table,
                                                             unsigned long long *__restrict__ dropped, int use_lds) {
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	GroupCell *local = (GroupCell *)lds_raw;
	const uint32_t n_cells = groups.n_groups * aggs.n;
	if (use_lds) {
		for (uint32_t i = threadIdx.x; i < n_cells; i += blockDim.x) {
			local[i].lo32 = 0;
			local[i].hi32 = 0;
			local[i].mn = 0x7FFFFFFFFFFFFFFFll;
			local[i].mx = (long long)0x8000000000000000ull;
			local[i].count = 0;
		}
		__syncthreads();
	}
	GroupCell *dst = use_lds ? local : table;
	unsigned long long my_dropped = 0;
	for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
		const uint32_t n = out.chunk_count[chunk];
		const uint64_t chunk_base = (uint64_t)chunk * out.chunk_capacity;
		for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
			// group index: mixed-radix number of the key offsets
			uint64_t g = 0;
			bool ok = true;
			for (uint32_t q = 0; q < groups.n; q++) {
				const DevGroupKey &gk = groups.k[q];
				const uint32_t row = out.ids[(uint64_t)gk.slot * out.slot_stride + chunk_base + i];
				if (gk.src.valid && !gk.src.valid[row]) {
					ok = false;
					break;
				}
				const long long v = agg_cell(gk.src, row);
				const unsigned long long off = (unsigned long long)(v - gk.min_value);
				if (off >= gk.n_values) {
					ok = false;
					break;
				}
				g = g * gk.n_values + off;
			}
			if (!ok) {
				my_dropped++;
				continue;
			}
			for (uint32_t a = 0; a < aggs.n; a++) {
				const DevAgg &ag = aggs.a[a];
				GroupCell *c = &dst[g * aggs.n + a];
				if (ag.fn == POLR_AGG_COUNT_STAR) {
					atomicAdd(&c->count, 1ull);
					continue;
				}
				const uint32_t row = out.ids[(uint64_t)ag.slot * out.slot_stride + chunk_base + i];
				if (ag.src.valid && !ag.src.valid[row]) {
					continue;
				}
				cell_add(c, agg_cell(ag.src, row));
			}
		}
	}
	if (my_dropped) {
		atomicAdd(dropped, my_dropped);
	}
	if (use_lds) {
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < n_cells; i += blockDim.x) {
			const GroupCell c = local[i];
			if (c.count) {
				atomicAdd(&table[i].lo32, c.lo32);
				atomicAdd((unsigned long long *)&table[i].hi32, (unsigned long long)c.hi32);
				atomicMin(&table[i].mn, c.mn);
				atomicMax(&table[i].mx, c.mx);
				atomicAdd(&table[i].count, c.count);
			}
		}
	}
}

static int resolve_agg_col(polr_pipeline *p, int32_t src_join, uint32_t src_col, const OwnedCol **c, uint32_t *slot,
                           const char *what, uint32_t idx) {
	polr_ctx *ctx = p->ctx;
	if (src_join < 0) {
		if (src_col >= p->n_probe_cols) {
			POLR_FAIL(ctx, POLR_E_INVALID, "%s %u: probe column %u out of range", what, idx, src_col);
		}
		*c = &p->probe_cols[src_col];
		*slot = 0;
	} else {
		if ((uint32_t)src_join >= p->k || src_col >= p->hts[src_join]->n_payload) {
			POLR_FAIL(ctx, POLR_E_INVALID, "%s %u: build column (%d,%u) out of range", what, idx, src_join, src_col);
		}
		const polr_ht *ht = p->hts[src_join];
		*c = ht->kind == KIND_PERFECT ? &ht->pcols[src_col] : &ht->payload[src_col];
		*slot = 1 + (uint32_t)src_join;
	}
	if ((*c)->width > 8 || ((*c)->width == 8 && !((*c)->flags & 1u))) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "%s %u: only integer columns of up to 8 bytes (signed if 8)", what, idx);
	}
	return POLR_OK;
}

extern "C" int polr_out_aggregate_grouped(polr_out *o, void *stream, const polr_group_key *keys, uint32_t n_keys,
                                          const polr_agg_spec *specs, uint32_t n_aggs, polr_agg_value *results,
                                          uint64_t n_groups, uint64_t *n_dropped) {
	if (!o || !keys || !specs || !results || n_keys == 0 || n_aggs == 0) {
		return POLR_E_INVALID;
	}
	polr_pipeline *p = o->pipe;
	polr_ctx *ctx = p->ctx;
	if (n_keys > POLR_MAX_GROUP_KEYS || n_aggs > POLR_MAX_AGGS) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "at most %d group columns and %d aggregates", POLR_MAX_GROUP_KEYS, POLR_MAX_AGGS);
	}
	DevGroupSet gs;
	memset(&gs, 0, sizeof(gs));
	gs.n = n_keys;
	uint64_t groups = 1;
	for (uint32_t q = 0; q < n_keys; q++) {
		const OwnedCol *c = nullptr;
		uint32_t slot = 0;
		int rc = resolve_agg_col(p, keys[q].src_join, keys[q].src_col, &c, &slot, "group column", q);
		if (rc) {
			return rc;
		}
		if (keys[q].n_values == 0) {
			POLR_FAIL(ctx, POLR_E_INVALID, "group column %u: empty domain", q);
		}
		groups *= keys[q].n_values;
		if (groups > (1u << 20)) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "more than 2^20 groups: not a perfect-hash aggregate");
		}
		gs.k[q].src.data = c->data;
		gs.k[q].src.valid = c->valid;
		gs.k[q].src.width = c->width;
		gs.k[q].src.flags = c->flags;
		gs.k[q].slot = slot;
		gs.k[q].n_values = keys[q].n_values;
		gs.k[q].min_value = keys[q].min_value;
	}
	if (groups != n_groups) {
		POLR_FAIL(ctx, POLR_E_INVALID, "results hold %llu groups, the domains span %llu", (unsigned long long)n_groups,
		          (unsigned long long)groups);
	}
	gs.n_groups = (uint32_t)groups;
	DevAggSet as;
	memset(&as, 0, sizeof(as));
	as.n = n_aggs;
	for (uint32_t a = 0; a < n_aggs; a++) {
		if (specs[a].fn > POLR_AGG_MAX) {
			POLR_FAIL(ctx, POLR_E_INVALID, "aggregate %u: unknown function %u", a, specs[a].fn);
		}
		as.a[a].fn = specs[a].fn;
		if (specs[a].fn == POLR_AGG_COUNT_STAR) {
			continue;
		}
		const OwnedCol *c = nullptr;
		uint32_t slot = 0;
		int rc = resolve_agg_col(p, specs[a].src_join, specs[a].src_col, &c, &slot, "aggregate", a);
		if (rc) {
			return rc;
		}
		as.a[a].src.data = c->data;
		as.a[a].src.valid = c->valid;
		as.a[a].src.width = c->width;
		as.a[a].src.flags = c->flags;
		as.a[a].slot = slot;
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	if (!o->stats_valid) {
		int rc = polr_out_stats(o, stream, nullptr, nullptr, nullptr);
		if (rc) {
			return rc;
		}
	}
	const uint32_t n_cells = (uint32_t)groups * n_aggs;
	std::vector<GroupCell> host(n_cells);
	unsigned long long h_dropped = 0;
	GroupCell *table = nullptr;
	unsigned long long *dropped = nullptr;
	hipError_t e = hipMalloc((void **)&table, (size_t)n_cells * sizeof(GroupCell));
	e = e == hipSuccess ? hipMalloc((void **)&dropped, 8) : e;
	e = e == hipSuccess ? hipMemsetAsync(dropped, 0, 8, st) : e;
	if (e == hipSuccess) {
		hipLaunchKernelGGL(polr_group_init_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, st, table, n_cells);
		if (o->n_chunks) {
			const int use_lds = n_cells <= POLR_GROUP_LDS_CELLS;
			const uint32_t n_blocks = std::max<uint32_t>(1, std::min<uint32_t>(o->n_chunks, (uint32_t)ctx->n_cus * 4));
			hipLaunchKernelGGL(polr_group_agg_kernel, dim3(n_blocks), dim3(256),
			                   use_lds ? (size_t)n_cells * sizeof(GroupCell) : 0, st, o->dev, o->n_chunks, gs, as, table,
			                   dropped, use_lds);
		}
		e = hipMemcpyAsync(host.data(), table, (size_t)n_cells * sizeof(GroupCell), hipMemcpyDeviceToHost, st);
		e = e == hipSuccess ? hipMemcpyAsync(&h_dropped, dropped, 8, hipMemcpyDeviceToHost, st) : e;
		e = e == hipSuccess ? hipStreamSynchronize(st) : e;
	}
	if (table) {
		hipFree(table);
	}
	if (dropped) {
		hipFree(dropped);
	}
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "grouped aggregate failed: %s", hipGetErrorString(e));
	}
	for (uint32_t i = 0; i < n_cells; i++) {
		const GroupCell &c = host[i];
		polr_agg_value &v = results[i];
		memset(&v, 0, sizeof(v));
		v.count = c.count;
		const uint32_t fn = specs[i % n_aggs].fn;
		const __int128 sum = ((__int128)c.hi32 << 32) + (__int128)c.lo32;
		switch (fn) {
		case POLR_AGG_COUNT_STAR:
		case POLR_AGG_COUNT:
			v.lo = (int64_t)c.count;
			break;
		case POLR_AGG_SUM:
			v.is_null = c.count == 0;
			v.lo = (int64_t)(unsigned long long)sum;
			v.hi = (int64_t)(sum >> 64);
			break;
		case POLR_AGG_MIN:
			v.is_null = c.count == 0;
			v.lo = c.count ? c.mn : 0;
			v.hi = (c.count && c.mn < 0) ? -1 : 0;
			break;
		default:
			v.is_null = c.count == 0;
			v.lo = c.count ? c.mx : 0;
			v.hi = (c.count && c.mx < 0) ? -1 : 0;
			break;
		}
	}
	if (n_dropped) {
		*n_dropped = h_dropped;
	}
	return POLR_OK;
}


// ---- the general GROUP BY sink: group columns of any integer domain (PhysicalHashAggregate,
// src/execution/operator/aggregate/physical_hash_aggregate.cpp -- the plan when the group columns' statistics do not allow
// a perfect-hash aggregate) -----------------------------------------------------------------------------------------------
// An open-addressing table of groups in global memory: state[s] = 0 empty / 1 being written / 2 ready, the group's key
// values (NULL is a group value of its own, as GROUP BY has it: a bit per column) and its cells.  A row claims an empty
// slot with a compare-and-swap and publishes its key; every other row of the group finds the key and adds to the cells.
// A lane that meets a slot "being written" does not wait inside the iteration -- the writer may be a lane of its own wave,
// which runs in lockstep -- it comes back to the slot in the next iteration of the loop all lanes share.
struct HashAggTable {
	uint32_t *state;
	long long *keys;          // [capacity][n_cols]
	uint32_t *nulls;          // [capacity]: bit c = group column c is NULL
	GroupCell *cells;         // [capacity][n_aggs]
	unsigned long long *n_groups, *overflow;
	uint64_t mask;            // capacity - 1
	uint64_t max_groups;
};

__global__ __launch_bounds__(256) void polr_hash_agg_init_kernel(HashAggTable t, uint32_t n_aggs) {
	const uint64_t n = (t.mask + 1) * n_aggs;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		t.cells[i].lo32 = 0;
		t.cells[i].hi32 = 0;
		t.cells[i].mn = 0x7FFFFFFFFFFFFFFFll;
		t.cells[i].mx = (long long)0x8000000000000000ull;
		t.cells[i].count = 0;
	}
}

__global__ __launch_bounds__(256) void polr_hash_agg_kernel(DevOut out, uint32_t n_chunks, DevGroupSet groups, DevAggSet aggs,
                                                            HashAggTable t) {
	for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
		const uint32_t n = out.chunk_count[chunk];
		const uint64_t chunk_base = (uint64_t)chunk * out.chunk_capacity;
		for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
			const uint32_t i = i0 + threadIdx.x;
			const bool active = i < n;
			long long key[POLR_MAX_GROUP_KEYS] = {0, 0, 0};
			uint32_t null_mask = 0;
			uint64_t h = 0x9E3779B97F4A7C15ull;
			if (active) {
				for (uint32_t q = 0; q < groups.n; q++) {
					const DevGroupKey &gk = groups.k[q];
					const uint32_t row = out.ids[(uint64_t)gk.slot * out.slot_stride + chunk_base + i];
					if (gk.src.valid && !gk.src.valid[row]) {
						null_mask |= 1u << q;
					} else {
						key[q] = agg_cell(gk.src, row);
					}
					h = polr_murmurhash64(h ^ (uint64_t)key[q]) + q;
				}
				h = polr_murmurhash64(h ^ null_mask);
			}
			uint64_t s = h & t.mask;
			bool done = !active;
			uint64_t probes = 0;
			while (__syncthreads_or(!done)) { // (every lane of the workgroup takes part in every iteration)
				if (!done) {
					uint32_t st = __hip_atomic_load(&t.state[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
					if (st == 0u) {
						if (atomicCAS(&t.state[s], 0u, 1u) == 0u) {
							if (atomicAdd(t.n_groups, 1ull) >= t.max_groups) {
								atomicExch(t.overflow, 1ull); // (more groups than the caller made room for)
							}
							for (uint32_t q = 0; q < groups.n; q++) {
								t.keys[s * groups.n + q] = key[q];
							}
							t.nulls[s] = null_mask;
							__hip_atomic_store(&t.state[s], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
							st = 2u;
						}
					}
					if (st == 2u) {
						bool same = t.nulls[s] == null_mask;
						for (uint32_t q = 0; q < groups.n; q++) {
							same = same && t.keys[s * groups.n + q] == key[q];
						}
						if (same) {
							done = true;
						} else {
							s = (s + 1) & t.mask;
							if (++probes > t.mask) { // (a full table: cannot happen with capacity >= 2 x max_groups before overflow)
								atomicExch(t.overflow, 1ull);
								done = true;
								s = ~0ull;
							}
						}
					}
					// (st == 1: somebody is writing this slot's key: look again in the next iteration)
				}
			}
			if (active && s != ~0ull) {
				for (uint32_t a = 0; a < aggs.n; a++) {
					const DevAgg &ag = aggs.a[a];
					GroupCell *c = &t.cells[s * aggs.n + a];
					if (ag.fn == POLR_AGG_COUNT_STAR) {
						atomicAdd(&c->count, 1ull);
						continue;
					}
					const uint32_t row = out.ids[(uint64_t)ag.slot * out.slot_stride + chunk_base + i];
					if (ag.src.valid && !ag.src.valid[row]) {
						continue;
					}
					cell_add(c, agg_cell(ag.src, row));
				}
			}
		}
	}
}

// the groups that exist, compacted: [idx] <- slot
__global__ __launch_bounds__(256) void polr_hash_agg_compact_kernel(HashAggTable t, uint32_t n_cols, uint32_t n_aggs, long long *keys_out,
                                                                    uint32_t *nulls_out, GroupCell *cells_out, unsigned long long *cursor,
                                                                    uint64_t max_groups) {
	for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= t.mask; s += (uint64_t)gridDim.x * blockDim.x) {
		if (t.state[s] != 2u) {
			continue;
		}
		const unsigned long long idx = atomicAdd(cursor, 1ull);
		if (idx >= max_groups) {
			continue;
		}
		for (uint32_t q = 0; q < n_cols; q++) {
			keys_out[idx * n_cols + q] = t.keys[s * n_cols + q];
		}
		nulls_out[idx] = t.nulls[s];
		for (uint32_t a = 0; a < n_aggs; a++) {
			cells_out[idx * n_aggs + a] = t.cells[s * n_aggs + a];
		}
	}
}

static void cell_to_value(const GroupCell &c, uint32_t fn, polr_agg_value &v) {
	memset(&v, 0, sizeof(v));
	v.count = c.count;
	const __int128 sum = ((__int128)c.hi32 << 32) + (__int128)c.lo32;
	switch (fn) {
	case POLR_AGG_COUNT_STAR:
	case POLR_AGG_COUNT:
		v.lo = (int64_t)c.count;
		break;
	case POLR_AGG_SUM:
		v.is_null = c.count == 0;
		v.lo = (int64_t)(unsigned long long)sum;
		v.hi = (int64_t)(sum >> 64);
		break;
	case POLR_AGG_MIN:
		v.is_null = c.count == 0;
		v.lo = c.count ? c.mn : 0;
		v.hi = (c.count && c.mn < 0) ? -1 : 0;
		break;
	default:
		v.is_null = c.count == 0;
		v.lo = c.count ? c.mx : 0;
		v.hi = (c.count && c.mx < 0) ? -1 : 0;
		break;
	}
}

extern "C" int polr_out_aggregate_hashed(polr_out *o, void *stream, const polr_group_key *cols, uint32_t n_cols,
                                         const polr_agg_spec *specs, uint32_t n_aggs, uint64_t max_groups, int64_t *group_keys,
                                         uint32_t *group_nulls, polr_agg_value *results, uint64_t *n_groups) {
	if (!o || !cols || !specs || !group_keys || !group_nulls || !results || !n_groups || n_cols == 0 || n_aggs == 0 ||
	    max_groups == 0) {
		return POLR_E_INVALID;
	}
	polr_pipeline *p = o->pipe;
	polr_ctx *ctx = p->ctx;
	if (n_cols > POLR_MAX_GROUP_KEYS || n_aggs > POLR_MAX_AGGS || max_groups > (1ull << 24)) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "at most %d group columns, %d aggregates and 2^24 groups", POLR_MAX_GROUP_KEYS,
		          POLR_MAX_AGGS);
	}
	DevGroupSet gs;
	memset(&gs, 0, sizeof(gs));
	gs.n = n_cols;
	for (uint32_t q = 0; q < n_cols; q++) {
		const OwnedCol *c = nullptr;
		uint32_t slot = 0;
		int rc = resolve_agg_col(p, cols[q].src_join, cols[q].src_col, &c, &slot, "group column", q);
		if (rc) {
			return rc;
		}
		gs.k[q].src.data = c->data;
		gs.k[q].src.valid = c->valid;
		gs.k[q].src.width = c->width;
		gs.k[q].src.flags = c->flags;
		gs.k[q].slot = slot;
	}
	DevAggSet as;
	memset(&as, 0, sizeof(as));
	as.n = n_aggs;
	for (uint32_t a = 0; a < n_aggs; a++) {
		if (specs[a].fn > POLR_AGG_MAX) {
			POLR_FAIL(ctx, POLR_E_INVALID, "aggregate %u: unknown function %u", a, specs[a].fn);
		}
		as.a[a].fn = specs[a].fn;
		if (specs[a].fn == POLR_AGG_COUNT_STAR) {
			continue;
		}
		const OwnedCol *c = nullptr;
		uint32_t slot = 0;
		int rc = resolve_agg_col(p, specs[a].src_join, specs[a].src_col, &c, &slot, "aggregate", a);
		if (rc) {
			return rc;
		}
		as.a[a].src.data = c->data;
		as.a[a].src.valid = c->valid;
		as.a[a].src.width = c->width;
		as.a[a].src.flags = c->flags;
		as.a[a].slot = slot;
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	if (!o->stats_valid) {
		int rc = polr_out_stats(o, stream, nullptr, nullptr, nullptr);
		if (rc) {
			return rc;
		}
	}
	uint64_t capacity = 1024;
	while (capacity < 2 * max_groups) {
		capacity <<= 1;
	}
	HashAggTable t;
	memset(&t, 0, sizeof(t));
	t.mask = capacity - 1;
	t.max_groups = max_groups;
	// one allocation: state, nulls, counters, keys, cells, and the compacted outputs behind them
	const size_t b_state = capacity * 4, b_nulls = capacity * 4, b_cnt = 64, b_keys = capacity * n_cols * 8,
	             b_cells = capacity * n_aggs * sizeof(GroupCell), b_okeys = max_groups * n_cols * 8, b_onulls = max_groups * 4,
	             b_ocells = max_groups * n_aggs * sizeof(GroupCell);
	uint8_t *base = nullptr;
	HIPCHK(ctx, hipMalloc((void **)&base, b_state + b_nulls + b_cnt + b_keys + b_cells + b_okeys + b_onulls + b_ocells));
	uint8_t *at = base;
	t.state = (uint32_t *)at;
	at += b_state;
	t.nulls = (uint32_t *)at;
	at += b_nulls;
	unsigned long long *cnt = (unsigned long long *)at; // [0] groups, [1] overflow, [2] compaction cursor
	at += b_cnt;
	t.n_groups = cnt;
	t.overflow = cnt + 1;
	t.keys = (long long *)at;
	at += b_keys;
	t.cells = (GroupCell *)at;
	at += b_cells;
	long long *okeys = (long long *)at;
	at += b_okeys;
	uint32_t *onulls = (uint32_t *)at;
	at += b_onulls;
	GroupCell *ocells = (GroupCell *)at;
	hipError_t e = hipMemsetAsync(base, 0, b_state + b_nulls + b_cnt, st);
	unsigned long long h_cnt[3] = {0, 0, 0};
	if (e == hipSuccess) {
		hipLaunchKernelGGL(polr_hash_agg_init_kernel, dim3(256), dim3(256), 0, st, t, n_aggs);
		if (o->n_chunks) {
			hipLaunchKernelGGL(polr_hash_agg_kernel, dim3(std::min<uint32_t>(o->n_chunks, 2048u)), dim3(256), 0, st, o->dev, o->n_chunks,
			                   gs, as, t);
		}
		hipLaunchKernelGGL(polr_hash_agg_compact_kernel, dim3(256), dim3(256), 0, st, t, n_cols, n_aggs, okeys, onulls, ocells, cnt + 2,
		                   max_groups);
		e = hipMemcpyAsync(h_cnt, cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st);
		e = e == hipSuccess ? hipStreamSynchronize(st) : e;
	}
	std::vector<GroupCell> hcells;
	if (e == hipSuccess && !h_cnt[1] && h_cnt[0] <= max_groups && h_cnt[0]) {
		const uint64_t g = h_cnt[0];
		hcells.resize(g * n_aggs);
		e = hipMemcpy(group_keys, okeys, g * n_cols * 8, hipMemcpyDeviceToHost);
		e = e == hipSuccess ? hipMemcpy(group_nulls, onulls, g * 4, hipMemcpyDeviceToHost) : e;
		e = e == hipSuccess ? hipMemcpy(hcells.data(), ocells, g * n_aggs * sizeof(GroupCell), hipMemcpyDeviceToHost) : e;
	}
	hipFree(base);
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "hash aggregate failed: %s", hipGetErrorString(e));
	}
	*n_groups = h_cnt[0];
	if (h_cnt[1] || h_cnt[0] > max_groups) {
		POLR_FAIL(ctx, POLR_E_OVERFLOW, "the result has %llu groups or more, the caller made room for %llu",
		          (unsigned long long)h_cnt[0], (unsigned long long)max_groups);
	}
	for (uint64_t g = 0; g < h_cnt[0]; g++) {
		for (uint32_t a = 0; a < n_aggs; a++) {
			cell_to_value(hcells[g * n_aggs + a], specs[a].fn, results[g * n_aggs + a]);
		}
	}
	return POLR_OK;
}

// ---- the GROUP BY sink fused into an emitting flat pipeline (FusedSink, polr_device.h; polr_flat_device.h) -------------
static_assert(POLR_AGG_COUNT_STAR == POLR_DEV_AGG_COUNT_STAR && POLR_AGG_COUNT == POLR_DEV_AGG_COUNT &&
                  POLR_AGG_SUM == POLR_DEV_AGG_SUM,
              "aggregate function codes");
#define POLR_FUSED_TABLES 256u // one table of group cells per workgroup (modulo): the adds of a workgroup stay among themselves

// sums the workgroup tables: out[w] += sum over the tables t = blockIdx.y, blockIdx.y + gridDim.y, ... of cells[t][w]
// (out zeroed by the caller; the table loop is split over gridDim.y for loads in flight)
__global__ __launch_bounds__(256) void polr_fused_reduce_kernel(const unsigned long long *__restrict__ cells, uint32_t n_tables,
                                                                uint32_t words, unsigned long long *__restrict__ out) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= words) {
		return;
	}
	unsigned long long s = 0;
	for (uint32_t t = blockIdx.y; t < n_tables; t += gridDim.y) {
		s += cells[(size_t)t * words + w];
	}
	if (s) {
		atomicAdd(&out[w], s);
	}
}

extern "C" int polr_out_fuse_grouped(polr_out *o, const polr_group_key *keys, uint32_t n_keys, const polr_agg_spec *specs,
                                     uint32_t n_aggs) {
	if (!o) {
		return POLR_E_INVALID;
	}
	polr_pipeline *p = o->pipe;
	polr_ctx *ctx = p->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	if (!keys || n_keys == 0) { // un-fuse: the output object collects row ids again
		if (o->fused_dev) {
			HIPCHK(ctx, hipDeviceSynchronize());
			hipFree(o->fused_dev);
			hipFree(o->fused_cells);
			hipFree(o->fused_dropped);
			o->fused_dev = nullptr;
			o->fused_cells = o->fused_dropped = nullptr;
			o->dev.fused = nullptr;
		}
		return POLR_OK;
	}
	if (!specs || n_aggs == 0 || n_keys > POLR_MAX_GROUP_KEYS || n_aggs > POLR_MAX_AGGS) {
		POLR_FAIL(ctx, POLR_E_INVALID, "1..%d group columns and 1..%d aggregates", POLR_MAX_GROUP_KEYS, POLR_MAX_AGGS);
	}
	if (!p->host_count.flat || !p->flat_emit) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED,
		          "a GROUP BY sink is fused into FLAT pipelines whose joins are all perfect tables (polr_pipeline_launch_info); "
		          "others: polr_out_aggregate_grouped over the emitted row ids");
	}
	if (o->fused_dev) {
		POLR_FAIL(ctx, POLR_E_INVALID, "the output object already has a fused sink");
	}
	FusedSink fs;
	memset(&fs, 0, sizeof(fs));
	fs.groups.n = n_keys;
	uint64_t groups = 1;
	for (uint32_t q = 0; q < n_keys; q++) {
		const OwnedCol *c = nullptr;
		uint32_t slot = 0;
		int rc = resolve_agg_col(p, keys[q].src_join, keys[q].src_col, &c, &slot, "group column", q);
		if (rc) {
			return rc;
		}
		groups *= keys[q].n_values;
		if (keys[q].n_values == 0 || groups > 4096) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "group column %u: a fused sink holds at most 4096 groups", q);
		}
		fs.groups.k[q].src.data = c->data;
		fs.groups.k[q].src.valid = c->valid;
		fs.groups.k[q].src.width = c->width;
		fs.groups.k[q].src.flags = c->flags;
		fs.groups.k[q].slot = slot;
		fs.groups.k[q].n_values = keys[q].n_values;
		fs.groups.k[q].min_value = keys[q].min_value;
	}
	fs.groups.n_groups = (uint32_t)groups;
	fs.aggs.n = n_aggs;
	for (uint32_t a = 0; a < n_aggs; a++) {
		if (specs[a].fn > POLR_AGG_SUM) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "aggregate %u: a fused sink computes COUNT(*), COUNT and SUM", a);
		}
		fs.aggs.a[a].fn = specs[a].fn;
		o->fused_fn[a] = specs[a].fn;
		o->fused_has_valid[a] = false;
		if (specs[a].fn == POLR_AGG_COUNT_STAR) {
			continue;
		}
		const OwnedCol *c = nullptr;
		uint32_t slot = 0;
		int rc = resolve_agg_col(p, specs[a].src_join, specs[a].src_col, &c, &slot, "aggregate", a);
		if (rc) {
			return rc;
		}
		if (specs[a].fn == POLR_AGG_SUM && c->width > 4) {
			// (the cells hold plain 64-bit sums: exact for 2^32 rows of values below 2^31)
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "aggregate %u: a fused SUM takes columns of at most 4 bytes", a);
		}
		fs.aggs.a[a].src.data = c->data;
		fs.aggs.a[a].src.valid = c->valid;
		fs.aggs.a[a].src.width = c->width;
		fs.aggs.a[a].src.flags = c->flags;
		fs.aggs.a[a].slot = slot;
		o->fused_has_valid[a] = c->valid != nullptr;
	}
	const uint32_t words = (uint32_t)groups * (1u + 2u * n_aggs);
	fs.n_tables = POLR_FUSED_TABLES;
	fs.words_per_table = words;
	const size_t bytes = (size_t)(POLR_FUSED_TABLES + 1u) * words * 8u; // (+ 1: where the read-out sums the tables)
	hipError_t e = hipMalloc((void **)&o->fused_cells, bytes);
	e = e == hipSuccess ? hipMalloc((void **)&o->fused_dropped, 8) : e;
	e = e == hipSuccess ? hipMalloc((void **)&o->fused_dev, sizeof(FusedSink)) : e;
	fs.cells = o->fused_cells;
	fs.dropped = o->fused_dropped;
	e = e == hipSuccess ? hipMemsetAsync(o->fused_cells, 0, bytes, ctx->stream) : e;
	e = e == hipSuccess ? hipMemsetAsync(o->fused_dropped, 0, 8, ctx->stream) : e;
	e = e == hipSuccess ? hipMemcpyAsync(o->fused_dev, &fs, sizeof(fs), hipMemcpyHostToDevice, ctx->stream) : e;
	e = e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
	if (e != hipSuccess) { // (nothing half-made stays behind)
		hipFree(o->fused_cells);
		hipFree(o->fused_dropped);
		hipFree(o->fused_dev);
		o->fused_cells = o->fused_dropped = nullptr;
		o->fused_dev = nullptr;
		POLR_FAIL(ctx, POLR_E_HIP, "fused sink: %s", hipGetErrorString(e));
	}
	o->fused_tables = POLR_FUSED_TABLES;
	o->fused_groups = (uint32_t)groups;
	o->fused_aggs = n_aggs;
	o->dev.fused = o->fused_dev;
	return POLR_OK;
}

extern "C" int polr_out_fused_result(polr_out *o, void *stream, polr_agg_value *results, uint64_t n_groups, uint64_t *n_dropped) {
	if (!o || !results) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = o->pipe->ctx;
	if (!o->fused_dev) {
		POLR_FAIL(ctx, POLR_E_INVALID, "the output object has no fused sink (polr_out_fuse_grouped)");
	}
	if (n_groups != o->fused_groups) {
		POLR_FAIL(ctx, POLR_E_INVALID, "results hold %llu groups, the sink %u", (unsigned long long)n_groups, o->fused_groups);
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	const uint32_t n_aggs = o->fused_aggs, words = o->fused_groups * (1u + 2u * n_aggs);
	unsigned long long *sum = o->fused_cells + (size_t)o->fused_tables * words; // (one more table behind the workgroups')
	HIPCHK(ctx, hipMemsetAsync(sum, 0, (size_t)words * 8u, st));
	hipLaunchKernelGGL(polr_fused_reduce_kernel, dim3((words + 255) / 256, 16), dim3(256), 0, st, o->fused_cells, o->fused_tables,
	                   words, sum);
	std::vector<unsigned long long> host(words);
	unsigned long long dropped = 0;
	hipError_t e = hipMemcpyAsync(host.data(), sum, (size_t)words * 8u, hipMemcpyDeviceToHost, st);
	e = e == hipSuccess ? hipMemcpyAsync(&dropped, o->fused_dropped, 8, hipMemcpyDeviceToHost, st) : e;
	e = e == hipSuccess ? hipStreamSynchronize(st) : e;
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "fused aggregate read-out failed: %s", hipGetErrorString(e));
	}
	for (uint32_t g = 0; g < o->fused_groups; g++) {
		const unsigned long long *c = &host[(size_t)g * (1u + 2u * n_aggs)];
		for (uint32_t a = 0; a < n_aggs; a++) {
			polr_agg_value &v = results[(size_t)g * n_aggs + a];
			memset(&v, 0, sizeof(v));
			const uint64_t count = o->fused_fn[a] == POLR_AGG_COUNT_STAR || !o->fused_has_valid[a] ? c[0] : c[2u + 2u * a];
			v.count = count;
			if (o->fused_fn[a] == POLR_AGG_SUM) {
				v.is_null = count == 0;
				v.lo = (int64_t)c[1u + 2u * a];
				v.hi = v.lo < 0 ? -1 : 0;
			} else {
				v.lo = (int64_t)count;
			}
		}
	}
	if (n_dropped) {
		*n_dropped = dropped;
	}
	return POLR_OK;
}

// ---- VARCHAR: string heaps on the device and the MIN / MAX sink over string_t cells -------------------------------------
// (string_type.hpp:23-28: {u32 length, char inlined[12]} or {u32 length, char prefix[4], char *ptr})
__global__ __launch_bounds__(256) void polr_rebase_strings_kernel(uint4 *cells, uint64_t n, uint64_t host_base, uint64_t host_end,
                                                                  uint64_t dev_base, unsigned long long *outside) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) {
		return;
	}
	uint4 c = cells[i];
	if (c.x > 12u) {
		const uint64_t ptr = ((uint64_t)c.w << 32) | c.z;
		if (ptr < host_base || ptr + c.x > host_end) {
			atomicAdd(outside, 1ull); // (a cell pointing outside the heap it was said to live in)
			return;
		}
		const uint64_t moved = ptr - host_base + dev_base;
		c.z = (uint32_t)moved;
		c.w = (uint32_t)(moved >> 32);
		cells[i] = c;
	}
}

__device__ __forceinline__ uint32_t str_byte(const uint4 &c, uint32_t i) {
	if (c.x <= 12u) {
		const uint32_t w = i < 4 ? c.y : (i < 8 ? c.z : c.w);
		return (w >> (8u * (i & 3u))) & 0xFFu;
	}
	if (i < 4) {
		return (c.y >> (8u * i)) & 0xFFu;
	}
	const uint8_t *p = (const uint8_t *)(((uint64_t)c.w << 32) | c.z);
	return p[i];
}

// a < b in the order of the reference's string comparison: unsigned bytes, a proper prefix first
__device__ __forceinline__ bool str_less(const uint4 &a, const uint4 &b) {
	const uint32_t n = a.x < b.x ? a.x : b.x;
	// the first four characters sit in the same place in both forms
	const uint32_t pa = __builtin_bswap32(a.y), pb = __builtin_bswap32(b.y);
	if (n >= 4 && pa != pb) {
		return pa < pb;
	}
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t x = str_byte(a, i), y = str_byte(b, i);
		if (x != y) {
			return x < y;
		}
	}
	return a.x < b.x;
}

struct StrPartial {
	uint4 cell;
	uint32_t have;
	uint32_t pad[3];
};

// phase 1: per workgroup the extreme of its share of the output rows; phase 2 (one workgroup, n_in partials): the extreme
__global__ __launch_bounds__(256) void polr_agg_string_kernel(DevOut out, uint32_t n_chunks, DevCol src, uint32_t slot, int want_max,
                                                              const StrPartial *__restrict__ in, uint32_t n_in,
                                                              StrPartial *__restrict__ partials) {
	__shared__ StrPartial wave_part[4];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint4 best = make_uint4(0, 0, 0, 0);
	bool have = false;
	auto offer = [&](const uint4 &c) {
		if (!have || (want_max ? str_less(best, c) : str_less(c, best))) {
			best = c;
			have = true;
		}
	};
	if (in) {
		for (uint32_t i = threadIdx.x; i < n_in; i += blockDim.x) {
			if (in[i].have) {
				offer(in[i].cell);
			}
		}
	} else {
		for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
			const uint32_t n = out.chunk_count[chunk];
			const uint32_t *ids = out.ids + (uint64_t)slot * out.slot_stride + (uint64_t)chunk * out.chunk_capacity;
			for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
				const uint32_t row = ids[i];
				if (src.valid && !src.valid[row]) {
					continue; // NULLs take no part
				}
				offer(((const uint4 *)src.data)[row]);
			}
		}
	}
	for (int d = 32; d > 0; d >>= 1) {
		uint4 o;
		o.x = __shfl_down(best.x, d, 64);
		o.y = __shfl_down(best.y, d, 64);
		o.z = __shfl_down(best.z, d, 64);
		o.w = __shfl_down(best.w, d, 64);
		const bool ohave = __shfl_down(have ? 1 : 0, d, 64) != 0;
		if (ohave && (int)lane + d < 64) {
			offer(o);
		}
	}
	if (lane == 0) {
		wave_part[wave].cell = best;
		wave_part[wave].have = have ? 1u : 0u;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		have = false;
		for (uint32_t w = 0; w < (blockDim.x >> 6); w++) {
			if (wave_part[w].have) {
				offer(wave_part[w].cell);
			}
		}
		StrPartial r;
		r.cell = best;
		r.have = have ? 1u : 0u;
		r.pad[0] = r.pad[1] = r.pad[2] = 0;
		partials[blockIdx.x] = r;
	}
}

static int set_string_heap(polr_ctx *ctx, std::vector<OwnedCol *> cols, uint64_t n_rows, const void *heap_base, uint64_t heap_bytes,
                           std::vector<void *> &owner) {
	if (!heap_base || heap_bytes == 0) {
		return POLR_E_INVALID;
	}
	for (OwnedCol *c : cols) {
		if (c->width != 16) {
			POLR_FAIL(ctx, POLR_E_INVALID, "a string heap belongs to a column of 16-byte string cells (this one: %u bytes)", c->width);
		}
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	void *heap = nullptr;
	unsigned long long *outside = nullptr;
	HIPCHK(ctx, hipMalloc(&heap, heap_bytes));
	hipError_t e = hipMalloc((void **)&outside, 8);
	e = e == hipSuccess ? hipMemcpyAsync(heap, heap_base, heap_bytes, hipMemcpyHostToDevice, ctx->stream) : e;
	e = e == hipSuccess ? hipMemsetAsync(outside, 0, 8, ctx->stream) : e;
	if (e == hipSuccess && n_rows) {
		for (OwnedCol *c : cols) {
			hipLaunchKernelGGL(polr_rebase_strings_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, ctx->stream,
			                   (uint4 *)c->data, n_rows, (uint64_t)heap_base, (uint64_t)heap_base + heap_bytes, (uint64_t)heap, outside);
		}
	}
	unsigned long long bad = 0;
	e = e == hipSuccess ? hipMemcpyAsync(&bad, outside, 8, hipMemcpyDeviceToHost, ctx->stream) : e;
	e = e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
	if (outside) {
		hipFree(outside);
	}
	if (e != hipSuccess || bad) {
		hipFree(heap);
		if (e != hipSuccess) {
			POLR_FAIL(ctx, POLR_E_HIP, "string heap upload failed: %s", hipGetErrorString(e));
		}
		POLR_FAIL(ctx, POLR_E_INVALID, "%llu string cells point outside the heap given for their column", bad);
	}
	owner.push_back(heap);
	return POLR_OK;
}

extern "C" {

int polr_ht_set_payload_heap(polr_ht *ht, uint32_t payload_col, const void *heap_base, uint64_t heap_bytes) {
	POLR_ENTRY();
	if (!ht || payload_col >= ht->n_payload) {
		return POLR_E_INVALID;
	}
	if (ht->kind != KIND_NONE) {
		// (a finalized perfect table keeps a re-ordered copy of every payload column: rebase before finalizing)
		POLR_FAIL(ht->ctx, POLR_E_INVALID, "set the string heap of a payload column before the table is finalized");
	}
	return set_string_heap(ht->ctx, {&ht->payload[payload_col]}, ht->n_rows_in, heap_base, heap_bytes, ht->heaps);
}

int polr_pipeline_set_probe_heap(polr_pipeline *p, uint32_t probe_col, const void *heap_base, uint64_t heap_bytes) {
	POLR_ENTRY();
	if (!p || probe_col >= p->n_probe_cols) {
		return POLR_E_INVALID;
	}
	if (!p->probe_cols[probe_col].owned) {
		POLR_FAIL(p->ctx, POLR_E_INVALID, "probe column %u lives in the caller's device memory: its cells must point into HBM already",
		          probe_col);
	}
	return set_string_heap(p->ctx, {&p->probe_cols[probe_col]}, p->n_probe_rows, heap_base, heap_bytes, p->heaps);
}

int polr_out_aggregate_string(polr_out *o, void *stream, uint32_t fn, int32_t src_join, uint32_t src_col, char *dst,
                              uint32_t dst_cap, uint32_t *len, uint32_t *is_null) {
	POLR_ENTRY();
	if (!o || !len || !is_null || (!dst && dst_cap)) {
		return POLR_E_INVALID;
	}
	polr_pipeline *p = o->pipe;
	polr_ctx *ctx = p->ctx;
	if (fn != POLR_AGG_MIN && fn != POLR_AGG_MAX) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "string aggregate %u (MIN or MAX)", fn);
	}
	const OwnedCol *c;
	uint32_t slot;
	if (src_join < 0) {
		if (src_col >= p->n_probe_cols) {
			POLR_FAIL(ctx, POLR_E_INVALID, "probe column %u out of range", src_col);
		}
		c = &p->probe_cols[src_col];
		slot = 0;
	} else {
		if ((uint32_t)src_join >= p->k || src_col >= p->hts[src_join]->n_payload) {
			POLR_FAIL(ctx, POLR_E_INVALID, "build column (%d,%u) out of range", src_join, src_col);
		}
		const polr_ht *ht = p->hts[src_join];
		c = ht->kind == KIND_PERFECT ? &ht->pcols[src_col] : &ht->payload[src_col];
		slot = 1 + (uint32_t)src_join;
	}
	if (c->width != 16) {
		POLR_FAIL(ctx, POLR_E_INVALID, "not a VARCHAR column (%u-byte cells)", c->width);
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	if (!o->stats_valid) {
		int rc = polr_out_stats(o, stream, nullptr, nullptr, nullptr);
		if (rc) {
			return rc;
		}
	}
	*len = 0;
	*is_null = 1;
	if (o->n_chunks == 0) {
		return POLR_OK;
	}
	const uint32_t n_blocks = std::max<uint32_t>(1, std::min<uint32_t>(o->n_chunks, (uint32_t)ctx->n_cus * 4));
	StrPartial *part = nullptr;
	HIPCHK(ctx, hipMalloc((void **)&part, ((size_t)n_blocks + 1) * sizeof(StrPartial)));
	DevCol src;
	src.data = c->data;
	src.valid = c->valid;
	src.width = c->width;
	src.flags = c->flags;
	hipLaunchKernelGGL(polr_agg_string_kernel, dim3(n_blocks), dim3(256), 0, st, o->dev, o->n_chunks, src, slot,
	                   fn == POLR_AGG_MAX ? 1 : 0, (const StrPartial *)nullptr, 0u, part);
	hipLaunchKernelGGL(polr_agg_string_kernel, dim3(1), dim3(256), 0, st, o->dev, 0u, src, slot, fn == POLR_AGG_MAX ? 1 : 0,
	                   (const StrPartial *)part, n_blocks, part + n_blocks);
	StrPartial win;
	hipError_t e = hipMemcpyAsync(&win, part + n_blocks, sizeof(win), hipMemcpyDeviceToHost, st);
	e = e == hipSuccess ? hipStreamSynchronize(st) : e;
	if (e == hipSuccess && win.have) {
		*is_null = 0;
		*len = win.cell.x;
		const uint32_t take = std::min<uint32_t>(win.cell.x, dst_cap);
		if (win.cell.x <= 12) {
			const uint32_t words[3] = {win.cell.y, win.cell.z, win.cell.w};
			memcpy(dst, words, take);
		} else if (take) {
			e = hipMemcpy(dst, (const void *)(((uint64_t)win.cell.w << 32) | win.cell.z), take, hipMemcpyDeviceToHost);
		}
	}
	hipFree(part);
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "string aggregate failed: %s", hipGetErrorString(e));
	}
	return POLR_OK;
}

} // extern "C"
