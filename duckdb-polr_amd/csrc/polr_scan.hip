// duckdb-polr_amd/csrc/polr_scan.hip -- the source side of the POLAR pipeline on the device.
//
// Reference: PhysicalTableScan hands the pipeline one DataChunk per STANDARD_VECTOR_SIZE-row vector of the
// table; pushed-down table filters (ConstantFilter / IS [NOT] NULL, AND-ed per column:
// src/storage/table/row_group.cpp:316-452 RowGroup::TemplatedScan, src/storage/table/column_segment.cpp:194-300
// TemplatedFilterSelection / FilterSelectionSwitch, :304-475 ColumnSegment::FilterSelection) thin every vector
// to the rows that pass, in row order; a vector with no survivor is skipped (row_group.cpp:399-416), a NULL
// never passes a comparison (column_segment.cpp:200).  The multiplexer therefore sees chunks of 1..V tuples.
//
// Here: three HBM-streaming kernels over the filter columns (already resident in HBM)
//   1. count   : one wave per vector, 64 rows per step, survivors counted with a ballot
//   2. scan    : exclusive prefix over the vectors of (tuples, non-empty vectors) packed in one u64
//                (per-1024-vector block sums -> one block scans the sums -> apply)
//   3. write   : one wave per vector again: ascending row ids to sel[], chunk boundary of every non-empty vector
// and the result -- selection + chunk boundaries -- stays on the device, installed in the pipeline; a
// multiplexer takes the boundaries with polr_mpx_use_scan_chunks.  Algorithmic bytes: 2 x (sum of filter column
// widths [+ validity bytes]) per table row + 4 per surviving row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "polr_internal.h"

struct DevFilter {
	const uint8_t *data;
	const uint8_t *valid;
	uint32_t width;
	uint32_t is_signed;
	uint32_t op;
	uint32_t pad;
	int64_t constant;
};

#define POLR_MAX_FILTERS 8
struct DevFilterSet {
	DevFilter f[POLR_MAX_FILTERS];
	uint32_t n;
	uint32_t pad;
};

// LIP (lookahead information passing, the reference's `PRAGMA enable_lip`): the filters of the joins further up the
// pipeline are applied to the source chunks before they enter it (PipelineExecutor::FetchFromSource,
// src/parallel/pipeline_executor.cpp:396-465 -> PhysicalHashJoin::ProbeBloomFilter, physical_hash_join.cpp:579-635).
// The reference probes a bloom filter per eligible join (single condition, key traced back to a source column, filtered
// build side: physical_join.cpp:57-107) in an order it re-sorts by miss rate every 64 chunks; the order changes the
// work, never the survivors.  Here the "filter" of a join is its own index -- bit table, unique-key or run hash table:
// a bloom filter without false positives -- evaluated for the rows that pass the table filters, cheapest first.
struct DevLip {
	const uint8_t *key_data;
	const uint8_t *key_valid;
	const void *table;
	uint64_t mask;
	int64_t min_value;
	uint64_t range;
	uint32_t key_width, key_signed, kind, pad;
};
struct DevLipSet {
	DevLip f[POLR_KMAX];
	uint32_t n;
	uint32_t pad;
};

__device__ __forceinline__ bool lip_contains(const DevLip &f, uint64_t row) {
	if (f.key_valid && !f.key_valid[row]) {
		return false; // NULL never joins
	}
	const uint8_t *p = f.key_data + row * f.key_width;
	uint64_t key;
	switch (f.key_width) {
	case 1:
		key = f.key_signed && f.kind == KIND_PERFECT ? (uint64_t)(int64_t)*(const int8_t *)p : (uint64_t)*p;
		break;
	case 2:
		key = f.key_signed && f.kind == KIND_PERFECT ? (uint64_t)(int64_t)*(const int16_t *)p : (uint64_t)*(const uint16_t *)p;
		break;
	case 4:
		key = f.key_signed && f.kind == KIND_PERFECT ? (uint64_t)(int64_t)*(const int32_t *)p : (uint64_t)*(const uint32_t *)p;
		break;
	default:
		key = *(const uint64_t *)p;
		break;
	}
	if (f.kind == KIND_PERFECT) {
		uint64_t idx;
		bool in_range;
		if (f.key_signed) {
			const int64_t v = (int64_t)key;
			in_range = v >= f.min_value && (uint64_t)(v - f.min_value) <= f.range;
			idx = (uint64_t)(v - f.min_value);
		} else {
			in_range = key >= (uint64_t)f.min_value && key - (uint64_t)f.min_value <= f.range;
			idx = key - (uint64_t)f.min_value;
		}
		return in_range && ((((const uint32_t *)f.table)[idx >> 5] >> (idx & 31)) & 1u);
	}
	if (f.kind == KIND_S8) {
		const uint2 *tab = (const uint2 *)f.table; // {key32, row}
		const uint32_t k32 = (uint32_t)key;
		uint64_t h = polr_murmurhash64((uint64_t)k32) & f.mask;
		while (true) {
			const uint2 e = tab[h];
			if (e.y == S8_EMPTY_ROW) {
				return false;
			}
			if (e.x == k32) {
				return true;
			}
			h = (h + 1) & f.mask;
		}
	}
	// KIND_S16: {key64, start, count}
	if (key == S16_EMPTY_KEY) {
		return true; // (the sentinel key has a side entry: let the join decide)
	}
	const uint4 *tab = (const uint4 *)f.table;
	uint64_t h = polr_murmurhash64(key) & f.mask;
	while (true) {
		const uint4 e = tab[h];
		const uint64_t k = ((uint64_t)e.y << 32) | e.x;
		if (k == S16_EMPTY_KEY) {
			return false;
		}
		if (k == key) {
			return e.w != 0;
		}
		h = (h + 1) & f.mask;
	}
}

#define PACK_SHIFT 40 // low 40 bits: tuples, high 24: non-empty vectors
#define PACK_MASK ((1ull << PACK_SHIFT) - 1)

__device__ __forceinline__ bool row_passes_filters(const DevFilterSet &fs, uint64_t row);

__device__ __forceinline__ bool row_passes(const DevFilterSet &fs, const DevLipSet &lip, uint64_t row) {
	bool ok = row_passes_filters(fs, row);
	for (uint32_t i = 0; i < lip.n; i++) {
		ok = ok && lip_contains(lip.f[i], row);
	}
	return ok;
}

__device__ __forceinline__ bool row_passes_filters(const DevFilterSet &fs, uint64_t row) {
	bool ok = true;
	for (uint32_t i = 0; i < fs.n; i++) {
		const DevFilter &f = fs.f[i];
		const bool valid = !(f.valid && !f.valid[row]);
		if (f.op == POLR_CMP_IS_NULL) {
			ok = ok && !valid;
			continue;
		}
		if (f.op == POLR_CMP_IS_NOT_NULL) {
			ok = ok && valid;
			continue;
		}
		const uint8_t *p = f.data + row * f.width;
		bool r;
		if (f.is_signed) {
			int64_t v;
			switch (f.width) {
			case 1:
				v = *(const int8_t *)p;
				break;
			case 2:
				v = *(const int16_t *)p;
				break;
			case 4:
				v = *(const int32_t *)p;
				break;
			default:
				v = *(const int64_t *)p;
				break;
			}
			const int64_t c = f.constant;
			r = f.op == POLR_CMP_EQ   ? v == c
			    : f.op == POLR_CMP_NE ? v != c
			    : f.op == POLR_CMP_LT ? v < c
			    : f.op == POLR_CMP_GT ? v > c
			    : f.op == POLR_CMP_LE ? v <= c
			                          : v >= c;
		} else {
			uint64_t v;
			switch (f.width) {
			case 1:
				v = *p;
				break;
			case 2:
				v = *(const uint16_t *)p;
				break;
			case 4:
				v = *(const uint32_t *)p;
				break;
			default:
				v = *(const uint64_t *)p;
				break;
			}
			const uint64_t c = (uint64_t)f.constant; // (host: constant >= 0 for unsigned columns)
			r = f.op == POLR_CMP_EQ   ? v == c
			    : f.op == POLR_CMP_NE ? v != c
			    : f.op == POLR_CMP_LT ? v < c
			    : f.op == POLR_CMP_GT ? v > c
			    : f.op == POLR_CMP_LE ? v <= c
			                          : v >= c;
		}
		ok = ok && valid && r;
	}
	return ok;
}

// one wave per vector (grid-stride); counts[v] = survivors of vector v, packed with its non-empty flag
__global__ __launch_bounds__(256) void polr_tscan_count_kernel(DevFilterSet fs, DevLipSet lip, uint64_t n_rows, uint32_t V,
                                                              uint64_t n_vec, unsigned long long *__restrict__ packed) {
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	for (uint64_t v = wave; v < n_vec; v += n_waves) {
		const uint64_t begin = v * V;
		const uint64_t end = begin + V < n_rows ? begin + V : n_rows;
		uint32_t cnt = 0;
		for (uint64_t r0 = begin; r0 < end; r0 += 64) {
			const uint64_t row = r0 + lane;
			const bool pass = row < end && row_passes(fs, lip, row);
			cnt += (uint32_t)__popcll(__ballot(pass));
		}
		if (lane == 0) {
			packed[v] = (unsigned long long)cnt | (cnt ? (1ull << PACK_SHIFT) : 0ull);
		}
	}
}

// block sums of 1024 packed entries each
__global__ __launch_bounds__(1024) void polr_tscan_block_sums_kernel(const unsigned long long *__restrict__ packed,
                                                                    uint64_t n_vec,
                                                                    unsigned long long *__restrict__ sums) {
	__shared__ unsigned long long warp_sums[16];
	const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
	unsigned long long v = i < n_vec ? packed[i] : 0ull;
	for (int d = 32; d > 0; d >>= 1) {
		v += __shfl_down(v, d, 64);
	}
	if ((threadIdx.x & 63) == 0) {
		warp_sums[threadIdx.x >> 6] = v;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long s = 0;
		for (int w = 0; w < 16; w++) {
			s += warp_sums[w];
		}
		sums[blockIdx.x] = s;
	}
}

// one block: exclusive scan of the block sums in place; total -> totals[0] (tuples), totals[1] (chunks)
__global__ __launch_bounds__(1024) void polr_tscan_sums_kernel(unsigned long long *__restrict__ sums, uint64_t n_blocks,
                                                              unsigned long long *__restrict__ totals) {
	__shared__ unsigned long long warp_tot[16];
	__shared__ unsigned long long carry_s;
	if (threadIdx.x == 0) {
		carry_s = 0;
	}
	__syncthreads();
	for (uint64_t base = 0; base < n_blocks; base += 1024) {
		const uint64_t i = base + threadIdx.x;
		const unsigned long long mine = i < n_blocks ? sums[i] : 0ull;
		unsigned long long incl = mine;
		for (int d = 1; d < 64; d <<= 1) {
			const unsigned long long o = __shfl_up(incl, d, 64);
			if ((int)(threadIdx.x & 63) >= d) {
				incl += o;
			}
		}
		if ((threadIdx.x & 63) == 63) {
			warp_tot[threadIdx.x >> 6] = incl;
		}
		__syncthreads();
		unsigned long long before = carry_s;
		for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) {
			before += warp_tot[w];
		}
		if (i < n_blocks) {
			sums[i] = before + incl - mine;
		}
		__syncthreads();
		if (threadIdx.x == 1023) {
			carry_s = before + incl;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		totals[0] = carry_s & PACK_MASK;
		totals[1] = carry_s >> PACK_SHIFT;
	}
}

// exclusive prefix of every vector = block base + scan inside the block of 1024; in place
__global__ __launch_bounds__(1024) void polr_tscan_apply_kernel(unsigned long long *__restrict__ packed, uint64_t n_vec,
                                                               const unsigned long long *__restrict__ sums) {
	__shared__ unsigned long long warp_tot[16];
	const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
	const unsigned long long mine = i < n_vec ? packed[i] : 0ull;
	unsigned long long incl = mine;
	for (int d = 1; d < 64; d <<= 1) {
		const unsigned long long o = __shfl_up(incl, d, 64);
		if ((int)(threadIdx.x & 63) >= d) {
			incl += o;
		}
	}
	if ((threadIdx.x & 63) == 63) {
		warp_tot[threadIdx.x >> 6] = incl;
	}
	__syncthreads();
	unsigned long long before = sums[blockIdx.x];
	for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) {
		before += warp_tot[w];
	}
	if (i < n_vec) {
		// keep this vector's own non-empty flag in bit 63 (the write kernel needs it)
		packed[i] = (before + incl - mine) | ((mine >> PACK_SHIFT) ? (1ull << 63) : 0ull);
	}
}

// one wave per vector: ascending row ids of the survivors, chunk boundary of every non-empty vector
__global__ __launch_bounds__(256) void polr_tscan_write_kernel(DevFilterSet fs, DevLipSet lip, uint64_t n_rows, uint32_t V,
                                                              uint64_t n_vec,
                                                              const unsigned long long *__restrict__ prefix,
                                                              uint32_t *__restrict__ sel,
                                                              uint64_t *__restrict__ chunk_offsets,
                                                              const unsigned long long *__restrict__ totals) {
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	if (wave == 0 && lane == 0) {
		chunk_offsets[totals[1]] = totals[0]; // the end of the last chunk
	}
	for (uint64_t v = wave; v < n_vec; v += n_waves) {
		const unsigned long long pv = prefix[v];
		if (!(pv >> 63)) {
			continue; // no survivor: the scan skips the vector
		}
		uint64_t out = pv & PACK_MASK;
		const uint64_t chunk = (pv & ~(1ull << 63)) >> PACK_SHIFT;
		if (lane == 0) {
			chunk_offsets[chunk] = out;
		}
		const uint64_t begin = v * V;
		const uint64_t end = begin + V < n_rows ? begin + V : n_rows;
		for (uint64_t r0 = begin; r0 < end; r0 += 64) {
			const uint64_t row = r0 + lane;
			const bool pass = row < end && row_passes(fs, lip, row);
			const uint64_t m = __ballot(pass);
			if (pass) {
				const uint32_t rank =
				    __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
				sel[out + rank] = (uint32_t)row;
			}
			out += (uint64_t)__popcll(m);
		}
	}
}

extern "C" {

int polr_pipeline_scan_filter(polr_pipeline *p, void *stream, const polr_scan_filter *filters, uint32_t n_filters,
                              uint32_t vector_size, uint64_t *n_selected, uint64_t *n_chunks) {
	return polr_pipeline_scan_filter_lip(p, stream, filters, n_filters, 0, vector_size, n_selected, n_chunks);
}

int polr_pipeline_scan_filter_lip(polr_pipeline *p, void *stream, const polr_scan_filter *filters, uint32_t n_filters,
                                  uint32_t lip_joins, uint32_t vector_size, uint64_t *n_selected, uint64_t *n_chunks) {
	POLR_ENTRY();
	if (!p || (!filters && n_filters)) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	if (n_filters > POLR_MAX_FILTERS) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "at most %d pushed-down filters", POLR_MAX_FILTERS);
	}
	if (vector_size < 2 || vector_size > 65536) {
		POLR_FAIL(ctx, POLR_E_INVALID, "vector size %u out of range", vector_size);
	}
	if (p->n_probe_rows >= 0xFFFFFFF0ull) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "source partition too large for 32-bit row ids");
	}
	DevFilterSet fs;
	memset(&fs, 0, sizeof(fs));
	fs.n = n_filters;
	for (uint32_t i = 0; i < n_filters; i++) {
		const polr_scan_filter &f = filters[i];
		if (f.col >= p->n_probe_cols) {
			POLR_FAIL(ctx, POLR_E_INVALID, "filter %u: column %u out of range", i, f.col);
		}
		if (f.op > POLR_CMP_IS_NOT_NULL) {
			POLR_FAIL(ctx, POLR_E_INVALID, "filter %u: unknown comparison %u", i, f.op);
		}
		const OwnedCol &c = p->probe_cols[f.col];
		const bool is_signed = (c.flags & 1u) != 0;
		if (!is_signed && f.constant < 0 && f.op <= POLR_CMP_GE) {
			POLR_FAIL(ctx, POLR_E_INVALID, "filter %u: negative constant against an unsigned column", i);
		}
		fs.f[i].data = c.data;
		fs.f[i].valid = c.valid;
		fs.f[i].width = c.width;
		fs.f[i].is_signed = is_signed ? 1u : 0u;
		fs.f[i].op = f.op;
		fs.f[i].constant = f.constant;
	}
	// LIP: the joins whose filters are applied at the source, smallest index structure first (cheapest test first)
	DevLipSet lip;
	memset(&lip, 0, sizeof(lip));
	{
		std::vector<uint32_t> js;
		for (uint32_t j = 0; j < p->k; j++) {
			if (!((lip_joins >> j) & 1u)) {
				continue;
			}
			const DevJoin &dj = p->host_count.joins[j];
			if (dj.n_keys != 1 || dj.key_src_join[0] >= 0) {
				POLR_FAIL(ctx, POLR_E_INVALID, "LIP: join %u is not keyed by one column of the source (physical_join.cpp:57-107)", j);
			}
			if (p->hts[j]->pack.packed) {
				POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "LIP: join %u compares its key by value / NULL = NULL (packed form)", j);
			}
			js.push_back(j);
		}
		if (lip_joins >> p->k) {
			POLR_FAIL(ctx, POLR_E_INVALID, "LIP: join mask names a join beyond the %u of the pipeline", p->k);
		}
		std::sort(js.begin(), js.end(), [&](uint32_t a, uint32_t b) { return p->hts[a]->device_bytes < p->hts[b]->device_bytes; });
		for (uint32_t j : js) {
			const DevJoin &dj = p->host_count.joins[j];
			const OwnedCol &c = p->probe_cols[dj.key_src_col[0]];
			DevLip &f = lip.f[lip.n++];
			f.key_data = c.data;
			f.key_valid = c.valid;
			f.key_width = c.width;
			f.key_signed = dj.key_signed;
			f.kind = dj.kind;
			f.table = dj.table;
			f.mask = dj.mask;
			f.min_value = dj.min_value;
			f.range = dj.range;
		}
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	if (p->scan_valid) {
		// A re-scan rewrites the selection, the chunk boundaries and the device copies of the pipeline in place, and runs
		// of the previous scan may still be in flight on their multiplexers' streams (passes are enqueued without a
		// host synchronisation): settle the device first.  (The first scan of a pipeline has nothing to wait for.)
		HIPCHK(ctx, hipDeviceSynchronize());
	}
	const uint64_t n_rows = p->n_probe_rows;
	const uint64_t n_vec = (n_rows + vector_size - 1) / vector_size;
	const uint64_t n_blocks = (n_vec + 1023) / 1024;
	if (n_vec >= (1ull << 23)) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "more than 2^23 scan vectors per partition");
	}
	// Scratch and result buffers belong to the pipeline and are sized for the worst case (every row survives, every
	// vector is a chunk), so the whole scan is enqueued without a host round trip in the middle; one
	// synchronisation at the end reads the two totals.
	if (p->scan_cap_rows < n_rows || p->scan_cap_vec < n_vec) {
		if (p->scan_packed) {
			hipFree(p->scan_packed);
			hipFree(p->scan_sums);
			hipFree(p->scan_totals);
			p->scan_packed = p->scan_sums = p->scan_totals = nullptr;
		}
		if (p->scan_sel) {
			if (p->sel_dev == p->scan_sel) {
				p->sel_dev = nullptr;
			}
			hipFree(p->scan_sel);
			p->scan_sel = nullptr;
		}
		if (p->scan_offsets_dev) {
			hipFree(p->scan_offsets_dev);
			p->scan_offsets_dev = nullptr;
		}
		hipError_t ea = hipMalloc((void **)&p->scan_packed, std::max<uint64_t>(n_vec, 1) * 8);
		ea = ea == hipSuccess ? hipMalloc((void **)&p->scan_sums, std::max<uint64_t>(n_blocks, 1) * 8) : ea;
		ea = ea == hipSuccess ? hipMalloc((void **)&p->scan_totals, 16) : ea;
		ea = ea == hipSuccess ? hipMalloc((void **)&p->scan_sel, std::max<uint64_t>(n_rows, 1) * 4) : ea;
		ea = ea == hipSuccess ? hipMalloc((void **)&p->scan_offsets_dev, (n_vec + 1) * 8) : ea;
		if (ea != hipSuccess) {
			p->scan_cap_rows = p->scan_cap_vec = 0;
			POLR_FAIL(ctx, POLR_E_HIP, "scan filter buffers: %s", hipGetErrorString(ea));
		}
		p->scan_cap_rows = n_rows;
		p->scan_cap_vec = n_vec;
	}
	unsigned long long *packed = p->scan_packed, *sums = p->scan_sums, *totals = p->scan_totals;
	uint32_t *sel = p->scan_sel;
	uint64_t *offs = p->scan_offsets_dev;
	uint64_t h_tot[2] = {0, 0};
	hipError_t e = hipSuccess;
	if (n_vec) {
		const uint32_t waves_per_block = 4;
		const uint32_t grid = (uint32_t)std::min<uint64_t>((n_vec + waves_per_block - 1) / waves_per_block,
		                                                   (uint64_t)ctx->n_cus * 8);
		hipLaunchKernelGGL(polr_tscan_count_kernel, dim3(grid), dim3(256), 0, st, fs, lip, n_rows, vector_size, n_vec, packed);
		hipLaunchKernelGGL(polr_tscan_block_sums_kernel, dim3((uint32_t)n_blocks), dim3(1024), 0, st, packed, n_vec, sums);
		hipLaunchKernelGGL(polr_tscan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, n_blocks, totals);
		hipLaunchKernelGGL(polr_tscan_apply_kernel, dim3((uint32_t)n_blocks), dim3(1024), 0, st, packed, n_vec, sums);
		hipLaunchKernelGGL(polr_tscan_write_kernel, dim3(grid), dim3(256), 0, st, fs, lip, n_rows, vector_size, n_vec, packed,
		                   sel, offs, totals);
		e = hipMemcpyAsync(h_tot, totals, 16, hipMemcpyDeviceToHost, st);
		e = e == hipSuccess ? hipStreamSynchronize(st) : e;
	} else {
		e = hipMemsetAsync(offs, 0, 8, st);
		e = e == hipSuccess ? hipStreamSynchronize(st) : e;
	}
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "scan filter failed: %s", hipGetErrorString(e));
	}
	// install: the selection is the pipeline's source now (the buffers stay the pipeline's scan buffers)
	if (p->sel_dev && p->sel_owned && p->sel_dev != p->scan_sel) {
		hipFree(p->sel_dev);
	}
	p->sel_dev = sel;
	p->sel_owned = false; // (freed as scan_sel)
	p->n_tuples = h_tot[0];
	p->scan_valid = true;
	p->scan_generation++;
	p->scan_n_chunks = h_tot[1];
	p->scan_vector_size = vector_size;
	p->host_mat.sel = p->sel_dev;
	p->host_mat.n_tuples = p->n_tuples;
	p->host_count.sel = p->sel_dev;
	p->host_count.n_tuples = p->n_tuples;
	HIPCHK(ctx, hipMemcpy(p->dev_mat, &p->host_mat, sizeof(DevPipeline), hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(p->dev_count, &p->host_count, sizeof(DevPipeline), hipMemcpyHostToDevice));
	if (n_selected) {
		*n_selected = h_tot[0];
	}
	if (n_chunks) {
		*n_chunks = h_tot[1];
	}
	return POLR_OK;
}

int polr_pipeline_fetch_scan(polr_pipeline *p, uint32_t *sel, uint64_t *chunk_offsets) {
	POLR_ENTRY();
	if (!p) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	if (!p->scan_valid) {
		POLR_FAIL(ctx, POLR_E_INVALID, "no scan result: call polr_pipeline_scan_filter first");
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	if (sel && p->n_tuples) {
		HIPCHK(ctx, hipMemcpy(sel, p->sel_dev, p->n_tuples * 4, hipMemcpyDeviceToHost));
	}
	if (chunk_offsets) {
		HIPCHK(ctx, hipMemcpy(chunk_offsets, p->scan_offsets_dev, (p->scan_n_chunks + 1) * 8, hipMemcpyDeviceToHost));
	}
	return POLR_OK;
}

} // extern "C"
