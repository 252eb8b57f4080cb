// duckdb-polr_amd/csrc/polr_gather.hip -- late materialisation of the output row ids + the dispatcher
// that picks the path-kernel instantiation (compiled per stage count K in polr_probe.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "polr_device.h"
#include "polr_mpx_device.h"

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

// sum the per-workgroup shards of the round counters: dst[r*k + j] = sum_s src[(r*NSHARD + s)*k + j]
__global__ void polr_reduce_counts_kernel(const unsigned long long *__restrict__ src, uint64_t n, uint32_t k,
                                          unsigned long long *__restrict__ dst) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // i = r*k + j
	if (i >= n) {
		return;
	}
	const uint64_t r = i / k, j = i % k;
	unsigned long long s = 0;
	for (uint32_t sh = 0; sh < POLR_NSHARD; sh++) {
		s += src[(r * POLR_NSHARD + sh) * k + j];
	}
	dst[i] = s;
}

extern "C++" void polr_launch_reduce_counts(hipStream_t stream, const unsigned long long *src, uint64_t n_rounds,
                                            uint32_t k, unsigned long long *dst) {
	const uint64_t n = n_rounds * k;
	if (n == 0) {
		return;
	}
	hipLaunchKernelGGL(polr_reduce_counts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, n, k,
	                   dst);
}

// ---- path-kernel dispatch over the compiled stage counts ------------------------------------------
#define DECL_K(KK)                                                                                                     \
	size_t polr_path_lds_bytes_k##KK(uint32_t W, uint32_t waves_per_block);                                           \
	int polr_path_occupancy_k##KK(uint32_t W, uint32_t waves_per_block);                                              \
	hipError_t polr_launch_path_kernel_k##KK(uint32_t W, uint32_t n_blocks, uint32_t waves_per_block,                 \
	                                         hipStream_t stream, const DevPipeline *pipe, const DevRound *rounds,    \
	                                         const uint64_t *unit_prefix, uint32_t n_rounds,                         \
	                                         const uint32_t *unit_sizes, DevOut out, unsigned long long *counts,    \
	                                         SelfRoute sr);
DECL_K(2)
DECL_K(4)
DECL_K(6)
DECL_K(8)
struct PoolRun;
// the generic pipeline's pool kernel (polr_poolg.hip; the build with packed composite keys / conditions: ...x)
uint32_t polr_poolg_waves_per_block(uint32_t k, uint32_t W);
size_t polr_poolg_lds_bytes(uint32_t k, uint32_t W);
int polr_poolg_occupancy(uint32_t k, uint32_t W);
int polr_poolg_occupancyx(uint32_t k, uint32_t W);
hipError_t polr_launch_poolg_kernel(uint32_t W, uint32_t k, uint32_t n_blocks, hipStream_t stream, const DevPipeline *pipe,
                                    const ResidentExec *execs, PoolRun *run, DevOut out);
hipError_t polr_launch_poolg_kernelx(uint32_t W, uint32_t k, uint32_t n_blocks, hipStream_t stream, const DevPipeline *pipe,
                                     const ResidentExec *execs, PoolRun *run, DevOut out);
#define DECL_POOL_K(KK)                                                                                                \
	size_t polr_pool_flat_lds_bytes_k##KK(uint32_t waves_per_block, uint32_t table_dwords);                           \
	size_t polr_pool_flat_wave_bytes_k##KK();                                                                         \
	int polr_pool_flat_occupancy_k##KK(uint32_t waves_per_block, uint32_t table_dwords);                              \
	int polr_pool_flat_occupancy_e_k##KK(uint32_t waves_per_block, uint32_t table_dwords);                            \
	hipError_t polr_launch_pool_flat_kernel_k##KK(uint32_t n_blocks, uint32_t waves_per_block, uint32_t table_dwords, \
	                                              hipStream_t stream, const DevPipeline *pipe,                        \
	                                              const ResidentExec *execs, PoolRun *run, DevOut out,                \
	                                              uint32_t fused_words);                                              \
	hipError_t polr_launch_pool_flat_kernel_e_k##KK(uint32_t n_blocks, uint32_t waves_per_block,                      \
	                                                uint32_t table_dwords, hipStream_t stream, const DevPipeline *pipe, \
	                                                const ResidentExec *execs, PoolRun *run, DevOut out,              \
	                                                uint32_t fused_words);
DECL_POOL_K(2)
DECL_POOL_K(4)
DECL_POOL_K(6)
DECL_POOL_K(8)

static uint32_t compiled_k(uint32_t k) {
	return k <= 2 ? 2 : (k <= 4 ? 4 : (k <= 6 ? 6 : 8));
}

extern "C++" size_t polr_path_lds_bytes(uint32_t k, uint32_t W, uint32_t waves_per_block) {
	switch (compiled_k(k)) {
	case 2:
		return polr_path_lds_bytes_k2(W, waves_per_block);
	case 4:
		return polr_path_lds_bytes_k4(W, waves_per_block);
	case 6:
		return polr_path_lds_bytes_k6(W, waves_per_block);
	default:
		return polr_path_lds_bytes_k8(W, waves_per_block);
	}
}

extern "C++" int polr_path_occupancy(uint32_t k, uint32_t W, uint32_t waves_per_block) {
	switch (compiled_k(k)) {
	case 2:
		return polr_path_occupancy_k2(W, waves_per_block);
	case 4:
		return polr_path_occupancy_k4(W, waves_per_block);
	case 6:
		return polr_path_occupancy_k6(W, waves_per_block);
	default:
		return polr_path_occupancy_k8(W, waves_per_block);
	}
}

extern "C++" hipError_t polr_launch_path_kernel(uint32_t W, uint32_t k, uint32_t n_blocks, uint32_t waves_per_block,
                                                hipStream_t stream, const DevPipeline *pipe, const DevRound *rounds,
                                                const uint64_t *unit_prefix, uint32_t n_rounds,
                                                const uint32_t *unit_sizes, DevOut out, unsigned long long *counts,
                                                SelfRoute sr) {
	switch (compiled_k(k)) {
	case 2:
		return polr_launch_path_kernel_k2(W, n_blocks, waves_per_block, stream, pipe, rounds, unit_prefix, n_rounds,
		                                  unit_sizes, out, counts, sr);
	case 4:
		return polr_launch_path_kernel_k4(W, n_blocks, waves_per_block, stream, pipe, rounds, unit_prefix, n_rounds,
		                                  unit_sizes, out, counts, sr);
	case 6:
		return polr_launch_path_kernel_k6(W, n_blocks, waves_per_block, stream, pipe, rounds, unit_prefix, n_rounds,
		                                  unit_sizes, out, counts, sr);
	default:
		return polr_launch_path_kernel_k8(W, n_blocks, waves_per_block, stream, pipe, rounds, unit_prefix, n_rounds,
		                                  unit_sizes, out, counts, sr);
	}
}

#define POOL_SWITCH(k_, call2, call4, call6, call8)                                                                  \
	switch (compiled_k(k_)) {                                                                                          \
	case 2:                                                                                                            \
		return call2;                                                                                                  \
	case 4:                                                                                                            \
		return call4;                                                                                                  \
	case 6:                                                                                                            \
		return call6;                                                                                                  \
	default:                                                                                                           \
		return call8;                                                                                                  \
	}

// the generic pipeline's pool kernel: one kernel per carried-slot count W, the join count is a run-time value
extern "C++" uint32_t polr_pool_waves_per_block(uint32_t k, uint32_t W) {
	return polr_poolg_waves_per_block(k, W);
}

extern "C++" size_t polr_pool_lds_bytes(uint32_t k, uint32_t W) {
	return polr_poolg_lds_bytes(k, W);
}

// ext: the pipeline has stages with an extension record (DevPipeline::ext) -> the POLR_EXT build of the kernel
extern "C++" int polr_pool_occupancy(uint32_t k, uint32_t W, bool ext) {
	return ext ? polr_poolg_occupancyx(k, W) : polr_poolg_occupancy(k, W);
}

extern "C++" hipError_t polr_launch_pool_kernel(uint32_t W, uint32_t k, uint32_t n_blocks, hipStream_t stream,
                                                const DevPipeline *pipe, const ResidentExec *execs, PoolRun *run,
                                                DevOut out, bool ext) {
	return ext ? polr_launch_poolg_kernelx(W, k, n_blocks, stream, pipe, execs, run, out)
	           : polr_launch_poolg_kernel(W, k, n_blocks, stream, pipe, execs, run, out);
}

extern "C++" size_t polr_pool_flat_lds_bytes(uint32_t k, uint32_t wpb, uint32_t table_dwords) {
	POOL_SWITCH(k, polr_pool_flat_lds_bytes_k2(wpb, table_dwords), polr_pool_flat_lds_bytes_k4(wpb, table_dwords),
	            polr_pool_flat_lds_bytes_k6(wpb, table_dwords), polr_pool_flat_lds_bytes_k8(wpb, table_dwords))
}

extern "C++" size_t polr_pool_flat_wave_bytes(uint32_t k) {
	POOL_SWITCH(k, polr_pool_flat_wave_bytes_k2(), polr_pool_flat_wave_bytes_k4(), polr_pool_flat_wave_bytes_k6(),
	            polr_pool_flat_wave_bytes_k8())
}

extern "C++" int polr_pool_flat_occupancy(uint32_t k, uint32_t wpb, uint32_t table_dwords, bool emit) {
	if (emit) {
		POOL_SWITCH(k, polr_pool_flat_occupancy_e_k2(wpb, table_dwords), polr_pool_flat_occupancy_e_k4(wpb, table_dwords),
		            polr_pool_flat_occupancy_e_k6(wpb, table_dwords), polr_pool_flat_occupancy_e_k8(wpb, table_dwords))
	}
	POOL_SWITCH(k, polr_pool_flat_occupancy_k2(wpb, table_dwords), polr_pool_flat_occupancy_k4(wpb, table_dwords),
	            polr_pool_flat_occupancy_k6(wpb, table_dwords), polr_pool_flat_occupancy_k8(wpb, table_dwords))
}

// emit: the run may write row ids (the build of the flat kernel that carries that code)
extern "C++" hipError_t polr_launch_pool_flat_kernel(uint32_t k, uint32_t n_blocks, uint32_t wpb, uint32_t table_dwords,
                                                     hipStream_t stream, const DevPipeline *pipe, const ResidentExec *execs,
                                                     PoolRun *run, DevOut out, bool emit, uint32_t fused_words) {
	if (emit) {
		POOL_SWITCH(k, polr_launch_pool_flat_kernel_e_k2(n_blocks, wpb, table_dwords, stream, pipe, execs, run, out, fused_words),
		            polr_launch_pool_flat_kernel_e_k4(n_blocks, wpb, table_dwords, stream, pipe, execs, run, out, fused_words),
		            polr_launch_pool_flat_kernel_e_k6(n_blocks, wpb, table_dwords, stream, pipe, execs, run, out, fused_words),
		            polr_launch_pool_flat_kernel_e_k8(n_blocks, wpb, table_dwords, stream, pipe, execs, run, out, fused_words))
	}
	POOL_SWITCH(k, polr_launch_pool_flat_kernel_k2(n_blocks, wpb, table_dwords, stream, pipe, execs, run, out, 0u),
	            polr_launch_pool_flat_kernel_k4(n_blocks, wpb, table_dwords, stream, pipe, execs, run, out, 0u),
	            polr_launch_pool_flat_kernel_k6(n_blocks, wpb, table_dwords, stream, pipe, execs, run, out, 0u),
	            polr_launch_pool_flat_kernel_k8(n_blocks, wpb, table_dwords, stream, pipe, execs, run, out, 0u))
}
