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

struct DevAgg {
	DevCol src;
	uint32_t slot; // 0: probe row ids, 1 + j: build row ids of join j
	uint32_t fn;
};

#define POLR_MAX_AGGS 8
struct DevAggSet {
	DevAgg a[POLR_MAX_AGGS];
	uint32_t n;
	uint32_t pad;
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
