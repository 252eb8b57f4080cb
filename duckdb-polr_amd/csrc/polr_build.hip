// duckdb-polr_amd/csrc/polr_build.hip -- device-side (re)build of the join tables.
//
// Takes what the reference's build side leaves behind (row-format blob of JoinHashTable, or the
// sunk columns) and produces the device-native layouts of polr_device.h.  Replaces, on the device:
//   JoinHashTable::Finalize / InsertHashes          src/execution/join_hashtable.cpp:305-377
//   PerfectHashJoinExecutor::BuildPerfectHashTable   src/execution/operator/join/perfect_hash_join_executor.cpp:20-122
// Not on the probe hot path; kept simple (global atomics, three-pass scan).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "polr_device.h"

// ---- row format -> SoA -------------------------------------------------------------------------
// rows: n x row_width bytes, validity bits first (RowLayout, row_layout.cpp:23-53).  One thread per
// (row, column) cell group: thread r copies column `col` of row r.
__global__ void polr_deserialize_col_kernel(const uint8_t *__restrict__ rows, uint64_t n_rows, uint32_t row_width,
                                            uint32_t col, uint32_t offset, uint32_t width,
                                            uint8_t *__restrict__ dst, uint8_t *__restrict__ dst_valid) {
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rows) {
		return;
	}
	const uint8_t *row = rows + r * row_width;
	const bool valid = (row[col >> 3] >> (col & 7)) & 1;
	if (dst_valid) {
		dst_valid[r] = valid ? 1 : 0;
	}
	for (uint32_t b = 0; b < width; b++) {
		dst[r * width + b] = valid ? row[offset + b] : 0;
	}
}

// ---- key normalisation ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t build_cell(const DevCol &col, uint64_t r, bool sx) {
	const uint8_t *p = col.data + r * col.width;
	switch (col.width) {
	case 1:
		return sx ? (uint64_t)(int64_t)(int8_t)*p : (uint64_t)*p;
	case 2:
		return sx ? (uint64_t)(int64_t) * (const int16_t *)p : (uint64_t) * (const uint16_t *)p;
	case 4:
		return sx ? (uint64_t)(int64_t) * (const int32_t *)p : (uint64_t) * (const uint32_t *)p;
	default:
		return *(const uint64_t *)p;
	}
}

// same convention as fetch_key() in polr_probe_device.h: plain form = zero-extended bit pattern, two 32-bit keys
// packed; packed form = sum of (value - min) << shift (KeyPack, polr_device.h)
__device__ __forceinline__ bool build_key(const DevCol *keys, uint32_t n_keys, uint64_t r, const KeyPack &pack,
                                          uint64_t &key) {
	key = 0;
	bool valid = true;
	for (uint32_t c = 0; c < n_keys; c++) {
		const DevCol col = keys[c];
		if (col.valid && !col.valid[r]) {
			if (pack.packed && ((pack.null_eq >> c) & 1u)) {
				key |= (pack.range[c] + 1u) << pack.shift[c]; // IS NOT DISTINCT FROM: the row stays, NULL is a key value
				continue;
			}
			valid = false;
		}
		if (pack.packed) {
			key |= (build_cell(col, r, pack.sx[c] != 0) - (uint64_t)pack.min[c]) << pack.shift[c];
		} else {
			const uint64_t v = build_cell(col, r, false);
			key = c == 0 ? v : (key | (v << 32));
		}
	}
	return valid;
}

// per key column: min and max of the (sign- or zero-extended) values of the rows that enter the table -- a NULL key
// drops the row unless its column is in null_eq (IS NOT DISTINCT FROM), where the cell itself is skipped -- compared
// as int64; out[2c] = min, out[2c+1] = max (initialised to INT64_MAX / INT64_MIN by the launcher)
__global__ void polr_key_minmax_kernel(const DevCol *__restrict__ keys, uint32_t n_keys, uint64_t n_rows, uint32_t null_eq,
                                       long long *out) {
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	bool valid = r < n_rows;
	for (uint32_t c = 0; c < n_keys && valid; c++) {
		if (keys[c].valid && !keys[c].valid[r] && !((null_eq >> c) & 1u)) {
			valid = false;
		}
	}
	for (uint32_t c = 0; c < n_keys; c++) {
		long long lo = 0x7FFFFFFFFFFFFFFFll, hi = -0x7FFFFFFFFFFFFFFFll - 1;
		if (valid && !(keys[c].valid && !keys[c].valid[r])) {
			lo = hi = (long long)build_cell(keys[c], r, (keys[c].flags & 1u) != 0);
		}
		for (int d = 32; d > 0; d >>= 1) {
			const long long l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64);
			lo = l2 < lo ? l2 : lo;
			hi = h2 > hi ? h2 : hi;
		}
		if ((threadIdx.x & 63) == 0 && lo <= hi) {
			atomicMin(&out[2 * c], lo);
			atomicMax(&out[2 * c + 1], hi);
		}
	}
}

__global__ void polr_s16_init_kernel(uint4 *slots, uint64_t capacity) {
	const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < capacity) {
		slots[s] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
	}
}

// pass 1: claim a slot per distinct key (64-bit CAS on the key half) and count the run length
__global__ void polr_s16_insert_kernel(const DevCol *__restrict__ keys, uint32_t n_keys, uint64_t n_rows, KeyPack pack,
                                       uint4 *slots, uint64_t mask, uint32_t *__restrict__ slot_of_row,
                                       uint32_t *sentinel_count, unsigned long long *n_valid) {
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rows) {
		return;
	}
	uint64_t key;
	if (!build_key(keys, n_keys, r, pack, key)) {
		slot_of_row[r] = 0xFFFFFFFFu; // NULL key: dropped (join_hashtable.cpp:170-192)
		return;
	}
	atomicAdd(n_valid, 1ull);
	if (key == S16_EMPTY_KEY) {
		atomicAdd(sentinel_count, 1u);
		slot_of_row[r] = 0xFFFFFFFEu;
		return;
	}
	uint64_t s = polr_murmurhash64(key) & mask;
	while (true) {
		unsigned long long *kp = (unsigned long long *)&slots[s];
		const unsigned long long old = atomicCAS(kp, (unsigned long long)S16_EMPTY_KEY, (unsigned long long)key);
		if (old == S16_EMPTY_KEY || old == key) {
			atomicAdd(&slots[s].w, 1u);
			slot_of_row[r] = (uint32_t)s;
			return;
		}
		s = (s + 1) & mask;
	}
}

// ---- exclusive scan of slots[].w into slots[].z (three passes, 1024 slots per block) --------------
#define SCAN_BLOCK 256
#define SCAN_ITEMS 4
__global__ void polr_scan_reduce_kernel(const uint4 *__restrict__ slots, uint64_t capacity,
                                        uint32_t *__restrict__ block_sums, uint32_t *max_run) {
	__shared__ uint32_t sh[SCAN_BLOCK];
	__shared__ uint32_t shm[SCAN_BLOCK];
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS;
	uint32_t sum = 0, mx = 0;
	for (int i = 0; i < SCAN_ITEMS; i++) {
		const uint64_t s = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
		if (s < capacity) {
			const uint32_t c = slots[s].w;
			sum += c;
			mx = c > mx ? c : mx;
		}
	}
	sh[threadIdx.x] = sum;
	shm[threadIdx.x] = mx;
	__syncthreads();
	for (int d = SCAN_BLOCK / 2; d > 0; d >>= 1) {
		if ((int)threadIdx.x < d) {
			sh[threadIdx.x] += sh[threadIdx.x + d];
			shm[threadIdx.x] = shm[threadIdx.x] > shm[threadIdx.x + d] ? shm[threadIdx.x] : shm[threadIdx.x + d];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		block_sums[blockIdx.x] = sh[0];
		atomicMax(max_run, shm[0]);
	}
}

// single block: exclusive scan of block_sums in place (n_blocks up to a few 100k)
__global__ void polr_scan_blocksums_kernel(uint32_t *block_sums, uint32_t n_blocks, uint32_t *total) {
	__shared__ uint32_t sh[1024];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0) {
		carry = 0;
	}
	__syncthreads();
	for (uint32_t base = 0; base < n_blocks; base += 1024) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < n_blocks ? block_sums[i] : 0;
		sh[threadIdx.x] = v;
		__syncthreads();
		for (int d = 1; d < 1024; d <<= 1) {
			uint32_t t = 0;
			if ((int)threadIdx.x >= d) {
				t = sh[threadIdx.x - d];
			}
			__syncthreads();
			sh[threadIdx.x] += t;
			__syncthreads();
		}
		const uint32_t incl = sh[threadIdx.x];
		if (i < n_blocks) {
			block_sums[i] = carry + incl - v;
		}
		__syncthreads();
		if (threadIdx.x == 1023) {
			carry += incl;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		*total = carry;
	}
}

__global__ void polr_scan_apply_kernel(uint4 *slots, uint64_t capacity, const uint32_t *__restrict__ block_sums) {
	__shared__ uint32_t sh[SCAN_BLOCK];
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS;
	uint32_t c[SCAN_ITEMS];
	uint32_t sum = 0;
	for (int i = 0; i < SCAN_ITEMS; i++) {
		const uint64_t s = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
		c[i] = s < capacity ? slots[s].w : 0;
		sum += c[i];
	}
	sh[threadIdx.x] = sum;
	__syncthreads();
	for (int d = 1; d < SCAN_BLOCK; d <<= 1) {
		uint32_t t = 0;
		if ((int)threadIdx.x >= d) {
			t = sh[threadIdx.x - d];
		}
		__syncthreads();
		sh[threadIdx.x] += t;
		__syncthreads();
	}
	uint32_t run = block_sums[blockIdx.x] + sh[threadIdx.x] - sum;
	for (int i = 0; i < SCAN_ITEMS; i++) {
		const uint64_t s = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
		if (s < capacity) {
			slots[s].z = run;
			run += c[i];
		}
	}
}

// pass 3: place every row into its key's run
__global__ void polr_s16_scatter_kernel(uint64_t n_rows, const uint4 *__restrict__ slots,
                                        const uint32_t *__restrict__ slot_of_row, uint32_t *__restrict__ cursor,
                                        uint32_t *__restrict__ rowids, uint32_t sentinel_start,
                                        uint32_t *sentinel_cursor) {
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rows) {
		return;
	}
	const uint32_t s = slot_of_row[r];
	if (s == 0xFFFFFFFFu) {
		return;
	}
	if (s == 0xFFFFFFFEu) {
		rowids[sentinel_start + atomicAdd(sentinel_cursor, 1u)] = (uint32_t)r;
		return;
	}
	rowids[slots[s].z + atomicAdd(&cursor[s], 1u)] = (uint32_t)r;
}

// unique 32-bit key: shrink to 8-byte {key,row} slots (same slot index, same probe sequence)
__global__ void polr_s16_to_s8_kernel(const uint4 *__restrict__ slots, uint64_t capacity,
                                      const uint32_t *__restrict__ rowids, uint2 *__restrict__ s8) {
	const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= capacity) {
		return;
	}
	const uint4 e = slots[s];
	if (e.x == 0xFFFFFFFFu && e.y == 0xFFFFFFFFu) {
		s8[s] = make_uint2(0u, S8_EMPTY_ROW);
	} else {
		s8[s] = make_uint2(e.x, rowids[e.z]);
	}
}

// ---- perfect hash table ----------------------------------------------------------------------------
// TemplatedFillSelectionVectorBuild (perfect_hash_join_executor.cpp:98-122): keys inside [min,max]
// set their bit; a second row on a set bit is a duplicate -> abort.
__global__ void polr_pht_mark_kernel(const DevCol *__restrict__ keys, uint64_t n_rows, int64_t min_value,
                                     uint64_t range, uint32_t is_signed, uint32_t *bits,
                                     uint32_t *__restrict__ idx_row, uint32_t *flags /* [0]=dup [1]=has_null */,
                                     unsigned long long *unique_keys) {
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rows) {
		return;
	}
	const DevCol col = keys[0];
	if (col.valid && !col.valid[r]) {
		atomicExch(&flags[1], 1u);
		return;
	}
	const uint8_t *p = col.data + r * col.width;
	uint64_t idx;
	bool in_range;
	if (is_signed) {
		int64_t v;
		switch (col.width) {
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
		in_range = v >= min_value && (uint64_t)(v - min_value) <= range;
		idx = (uint64_t)(v - min_value);
	} else {
		uint64_t v;
		switch (col.width) {
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
		in_range = v >= (uint64_t)min_value && v - (uint64_t)min_value <= range;
		idx = v - (uint64_t)min_value;
	}
	if (!in_range) {
		return;
	}
	const uint32_t bit = 1u << (idx & 31);
	const uint32_t old = atomicOr(&bits[idx >> 5], bit);
	if (old & bit) {
		atomicExch(&flags[0], 1u);
		return;
	}
	idx_row[idx] = (uint32_t)r;
	atomicAdd(unique_keys, 1ull);
}

// perfect column c: cell idx = payload cell of the row that owns idx (RowOperations::Gather into
// perfect_hash_table[i], perfect_hash_join_executor.cpp:66-73)
__global__ void polr_pht_gather_kernel(const uint32_t *__restrict__ bits, const uint32_t *__restrict__ idx_row,
                                       uint64_t size, DevCol src, uint8_t *__restrict__ dst,
                                       uint8_t *__restrict__ dst_valid) {
	const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= size) {
		return;
	}
	const bool set = (bits[idx >> 5] >> (idx & 31)) & 1u;
	const uint32_t w = src.width;
	bool valid = false;
	uint32_t row = 0;
	if (set) {
		row = idx_row[idx];
		valid = src.valid ? src.valid[row] != 0 : true;
	}
	if (dst_valid) {
		dst_valid[idx] = valid ? 1 : 0;
	}
	for (uint32_t b = 0; b < w; b++) {
		dst[idx * w + b] = valid ? src.data[(uint64_t)row * w + b] : 0;
	}
}

// bool bitmap (one byte per key value, as the reference keeps it) -> bit words
__global__ void polr_pack_bitmap_kernel(const uint8_t *__restrict__ bytes, uint64_t size, uint32_t *bits) {
	const uint64_t wd = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (wd * 32 >= size) {
		return;
	}
	uint32_t v = 0;
	for (uint32_t b = 0; b < 32; b++) {
		const uint64_t i = wd * 32 + b;
		if (i < size && bytes[i]) {
			v |= 1u << b;
		}
	}
	bits[wd] = v;
}

// exclusive prefix of chunk_count -> chunk_base (single block; output chunk lists are short)
__global__ void polr_chunk_prefix_kernel(const uint32_t *__restrict__ chunk_count, uint32_t n_chunks,
                                         uint64_t *__restrict__ chunk_base, uint64_t *total) {
	__shared__ uint64_t sh[1024];
	__shared__ uint64_t carry;
	if (threadIdx.x == 0) {
		carry = 0;
	}
	__syncthreads();
	for (uint32_t base = 0; base < n_chunks; base += 1024) {
		const uint32_t i = base + threadIdx.x;
		const uint64_t v = i < n_chunks ? chunk_count[i] : 0;
		sh[threadIdx.x] = v;
		__syncthreads();
		for (int d = 1; d < 1024; d <<= 1) {
			uint64_t t = 0;
			if ((int)threadIdx.x >= d) {
				t = sh[threadIdx.x - d];
			}
			__syncthreads();
			sh[threadIdx.x] += t;
			__syncthreads();
		}
		const uint64_t incl = sh[threadIdx.x];
		if (i < n_chunks) {
			chunk_base[i] = carry + incl - v;
		}
		__syncthreads();
		if (threadIdx.x == 1023) {
			carry += incl;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		*total = carry;
	}
}

// ---- launchers (called from polr_capi.hip) ----------------------------------------------------------
static inline dim3 grid1d(uint64_t n, uint32_t block) {
	return dim3((unsigned)((n + block - 1) / block));
}

extern "C++" void polr_launch_deserialize_col(hipStream_t st, const uint8_t *rows, uint64_t n_rows, uint32_t row_width,
                                              uint32_t col, uint32_t offset, uint32_t width, uint8_t *dst,
                                              uint8_t *dst_valid) {
	if (n_rows == 0) {
		return;
	}
	hipLaunchKernelGGL(polr_deserialize_col_kernel, grid1d(n_rows, 256), dim3(256), 0, st, rows, n_rows, row_width, col,
	                   offset, width, dst, dst_valid);
}

extern "C++" void polr_launch_key_minmax(hipStream_t st, const DevCol *keys_dev, uint32_t n_keys, uint64_t n_rows,
                                         uint32_t null_eq, long long *out /* [2 * n_keys], initialised by the caller */) {
	if (n_rows) {
		hipLaunchKernelGGL(polr_key_minmax_kernel, grid1d(n_rows, 256), dim3(256), 0, st, keys_dev, n_keys, n_rows, null_eq,
		                   out);
	}
}

extern "C++" void polr_launch_s16_build(hipStream_t st, const DevCol *keys_dev, uint32_t n_keys, const KeyPack &pack,
                                        uint64_t n_rows, uint4 *slots, uint64_t capacity, uint32_t *slot_of_row, uint32_t *cursor,
                                        uint32_t *rowids, uint32_t *block_sums, uint32_t *scalars
                                        /* [0]=sentinel_count [1]=max_run [2]=total [3]=sentinel_cursor */,
                                        unsigned long long *n_valid) {
	const uint64_t mask = capacity - 1;
	hipLaunchKernelGGL(polr_s16_init_kernel, grid1d(capacity, 256), dim3(256), 0, st, slots, capacity);
	if (n_rows) {
		hipLaunchKernelGGL(polr_s16_insert_kernel, grid1d(n_rows, 256), dim3(256), 0, st, keys_dev, n_keys, n_rows,
		                   pack, slots, mask, slot_of_row, &scalars[0], n_valid);
	}
	const uint32_t n_blocks = (uint32_t)((capacity + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS));
	hipLaunchKernelGGL(polr_scan_reduce_kernel, dim3(n_blocks), dim3(SCAN_BLOCK), 0, st, slots, capacity, block_sums,
	                   &scalars[1]);
	hipLaunchKernelGGL(polr_scan_blocksums_kernel, dim3(1), dim3(1024), 0, st, block_sums, n_blocks, &scalars[2]);
	hipLaunchKernelGGL(polr_scan_apply_kernel, dim3(n_blocks), dim3(SCAN_BLOCK), 0, st, slots, capacity, block_sums);
	(void)cursor;
	(void)rowids;
}

extern "C++" void polr_launch_s16_scatter(hipStream_t st, uint64_t n_rows, const uint4 *slots,
                                          const uint32_t *slot_of_row, uint32_t *cursor, uint32_t *rowids,
                                          uint32_t sentinel_start, uint32_t *sentinel_cursor) {
	if (n_rows == 0) {
		return;
	}
	hipLaunchKernelGGL(polr_s16_scatter_kernel, grid1d(n_rows, 256), dim3(256), 0, st, n_rows, slots, slot_of_row,
	                   cursor, rowids, sentinel_start, sentinel_cursor);
}

extern "C++" void polr_launch_s16_to_s8(hipStream_t st, const uint4 *slots, uint64_t capacity, const uint32_t *rowids,
                                        uint2 *s8) {
	hipLaunchKernelGGL(polr_s16_to_s8_kernel, grid1d(capacity, 256), dim3(256), 0, st, slots, capacity, rowids, s8);
}

extern "C++" void polr_launch_pht_mark(hipStream_t st, const DevCol *keys_dev, uint64_t n_rows, int64_t min_value,
                                       uint64_t range, uint32_t is_signed, uint32_t *bits, uint32_t *idx_row,
                                       uint32_t *flags, unsigned long long *unique_keys) {
	if (n_rows == 0) {
		return;
	}
	hipLaunchKernelGGL(polr_pht_mark_kernel, grid1d(n_rows, 256), dim3(256), 0, st, keys_dev, n_rows, min_value, range,
	                   is_signed, bits, idx_row, flags, unique_keys);
}

extern "C++" void polr_launch_pht_gather(hipStream_t st, const uint32_t *bits, const uint32_t *idx_row, uint64_t size,
                                         DevCol src, uint8_t *dst, uint8_t *dst_valid) {
	hipLaunchKernelGGL(polr_pht_gather_kernel, grid1d(size, 256), dim3(256), 0, st, bits, idx_row, size, src, dst,
	                   dst_valid);
}

extern "C++" void polr_launch_pack_bitmap(hipStream_t st, const uint8_t *bytes, uint64_t size, uint32_t *bits) {
	hipLaunchKernelGGL(polr_pack_bitmap_kernel, grid1d((size + 31) / 32, 256), dim3(256), 0, st, bytes, size, bits);
}

extern "C++" void polr_launch_chunk_prefix(hipStream_t st, const uint32_t *chunk_count, uint32_t n_chunks,
                                           uint64_t *chunk_base, uint64_t *total) {
	hipLaunchKernelGGL(polr_chunk_prefix_kernel, dim3(1), dim3(1024), 0, st, chunk_count, n_chunks, chunk_base, total);
}
