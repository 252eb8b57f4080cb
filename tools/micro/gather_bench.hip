// tools/micro/gather_bench.hip -- design microbenchmark (not product code): what does one bitmap lookup cost on gfx950
// by where the bitmap lives (LDS / L2 / MALL+HBM) and how fast do k key columns stream?
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// stream NCOL u32 columns, uint4 per lane per column, test a bit in global bitmap(s)
template <int NCOL, int MODE> // MODE 0: keys only (sum), 1: global bitmap lookups, 2: LDS bitmap lookups, 3: global sc1 lookups
__global__ __launch_bounds__(256) void stream_kernel(const uint32_t *const *cols, uint64_t n, const uint32_t *bits, uint32_t nbits_mask,
                                                     unsigned long long *out, uint32_t lds_words) {
	extern __shared__ uint32_t lbits[];
	if (MODE == 2) {
		for (uint32_t i = threadIdx.x; i < lds_words; i += blockDim.x) lbits[i] = bits[i];
		__syncthreads();
	}
	const uint32_t *c[NCOL];
#pragma unroll
	for (int j = 0; j < NCOL; j++) c[j] = cols[j];
	uint64_t cnt = 0;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
	for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 3 < n; i += stride) {
		uint4 k[NCOL];
#pragma unroll
		for (int j = 0; j < NCOL; j++) k[j] = *(const uint4 *)(c[j] + i);
		uint32_t alive = 0xF;
#pragma unroll
		for (int j = 0; j < NCOL; j++) {
			const uint32_t kk[4] = {k[j].x, k[j].y, k[j].z, k[j].w};
#pragma unroll
			for (int q = 0; q < 4; q++) {
				uint32_t idx = kk[q] & nbits_mask;
				uint32_t w;
				if (MODE == 0) w = kk[q];
				else if (MODE == 1) w = bits[idx >> 5] >> (idx & 31);
				else if (MODE == 2) w = lbits[idx >> 5] >> (idx & 31);
				else w = __builtin_nontemporal_load(&bits[idx >> 5]) >> (idx & 31);
				if (!(w & 1)) alive &= ~(1u << q);
			}
		}
		cnt += __popc(alive);
	}
	// wave reduce
	for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
	if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, (unsigned long long)cnt);
}

// lazy variant: stage j only for alive lanes (predicated), 4 tuples per lane
template <int NCOL>
__global__ __launch_bounds__(256) void lazy_kernel(const uint32_t *const *cols, uint64_t n, const uint32_t *const *bitsv, const uint32_t *masks,
                                                   unsigned long long *out) {
	const uint32_t *c[NCOL]; const uint32_t *b[NCOL]; uint32_t m[NCOL];
#pragma unroll
	for (int j = 0; j < NCOL; j++) { c[j] = cols[j]; b[j] = bitsv[j]; m[j] = masks[j]; }
	uint64_t cnt = 0;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
	for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 3 < n; i += stride) {
		uint32_t alive = 0xF;
#pragma unroll
		for (int j = 0; j < NCOL; j++) {
			if (__ballot(alive != 0) == 0) break;
			uint32_t kk[4];
#pragma unroll
			for (int q = 0; q < 4; q++) kk[q] = (alive >> q) & 1 ? c[j][i + q] : 0;
#pragma unroll
			for (int q = 0; q < 4; q++) {
				if ((alive >> q) & 1) {
					uint32_t idx = kk[q] & m[j];
					if (!((b[j][idx >> 5] >> (idx & 31)) & 1)) alive &= ~(1u << q);
				}
			}
		}
		cnt += __popc(alive);
	}
	for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
	if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, (unsigned long long)cnt);
}

template <class F>
static float timeit(F f, int reps) {
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	f(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	for (int i = 0; i < reps; i++) f();
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

int main(int argc, char **argv) {
	const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 400000000ull;
	const int NC = 4;
	std::vector<uint32_t *> dcols(NC);
	std::vector<uint32_t> h(1 << 24);
	for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7) * 40503u;
	for (int j = 0; j < NC; j++) {
		CK(hipMalloc(&dcols[j], n * 4));
		// fill by repeating a 64 MB pseudo-random block with an offset
		for (uint64_t off = 0; off < n; off += h.size()) {
			uint64_t cnt = n - off < h.size() ? n - off : h.size();
			CK(hipMemcpy(dcols[j] + off, h.data() + 0, cnt * 4, hipMemcpyHostToDevice));
		}
	}
	// make the columns differ: xor with j via a tiny kernel is overkill; keys are masked per test anyway
	const uint32_t **dcolv; CK(hipMalloc(&dcolv, NC * sizeof(void *)));
	CK(hipMemcpy(dcolv, dcols.data(), NC * sizeof(void *), hipMemcpyHostToDevice));
	unsigned long long *dout; CK(hipMalloc(&dout, 8)); CK(hipMemset(dout, 0, 8));
	// bitmaps of various sizes (bits): 64K (8KB), 256K (32KB), 2M (256KB), 4M (512KB), 32M (4MB), 256M (32MB), 2G bits (256MB)
	const uint64_t sizes_bits[] = {1ull << 16, 1ull << 18, 1ull << 21, 1ull << 22, 1ull << 25, 1ull << 28, 1ull << 31};
	uint32_t *dbits; CK(hipMalloc(&dbits, (1ull << 31) / 8));
	CK(hipMemset(dbits, 0xFF, (1ull << 31) / 8)); // all ones: every tuple stays alive (worst case: all lookups needed)
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int grid = prop.multiProcessorCount * 8;
	printf("n=%llu tuples, grid=%d x 256\n", (unsigned long long)n, grid);
	{
		float ms1 = timeit([&] { hipLaunchKernelGGL((stream_kernel<1, 0>), dim3(grid), dim3(256), 0, 0, dcolv, n, dbits, 0u, dout, 0u); }, 5);
		float ms4 = timeit([&] { hipLaunchKernelGGL((stream_kernel<4, 0>), dim3(grid), dim3(256), 0, 0, dcolv, n, dbits, 0u, dout, 0u); }, 5);
		printf("stream 1 col: %.3f ms  %.1f GB/s | 4 cols: %.3f ms %.1f GB/s  %.1f Gtuples/s\n", ms1, n * 4 / ms1 / 1e6, ms4, n * 16 / ms4 / 1e6, n / ms4 / 1e6);
	}
	for (uint64_t nb : sizes_bits) {
		const uint32_t mask = (uint32_t)(nb - 1);
		float g1 = timeit([&] { hipLaunchKernelGGL((stream_kernel<1, 1>), dim3(grid), dim3(256), 0, 0, dcolv, n, dbits, mask, dout, 0u); }, 3);
		float g4 = timeit([&] { hipLaunchKernelGGL((stream_kernel<4, 1>), dim3(grid), dim3(256), 0, 0, dcolv, n, dbits, mask, dout, 0u); }, 3);
		float s1 = timeit([&] { hipLaunchKernelGGL((stream_kernel<1, 3>), dim3(grid), dim3(256), 0, 0, dcolv, n, dbits, mask, dout, 0u); }, 3);
		printf("bitmap %8llu KB: global 1 col %.3f ms (%.1f Glookups/s) | 4 cols eager %.3f ms (%.1f Glookups/s) | nt 1 col %.3f ms (%.1f)\n",
		       (unsigned long long)(nb / 8 / 1024), g1, n / g1 / 1e6, g4, 4.0 * n / g4 / 1e6, s1, n / s1 / 1e6);
		if (nb / 8 <= 64 * 1024) {
			const uint32_t words = (uint32_t)(nb / 32);
			float l1 = timeit([&] { hipLaunchKernelGGL((stream_kernel<1, 2>), dim3(grid), dim3(256), words * 4, 0, dcolv, n, dbits, mask, dout, words); }, 3);
			float l4 = timeit([&] { hipLaunchKernelGGL((stream_kernel<4, 2>), dim3(grid), dim3(256), words * 4, 0, dcolv, n, dbits, mask, dout, words); }, 3);
			printf("                    LDS    1 col %.3f ms (%.1f Glookups/s) | 4 cols eager %.3f ms (%.1f Glookups/s)\n", l1, n / l1 / 1e6, l4, 4.0 * n / l4 / 1e6);
		}
	}
	// lazy with selective first stage: bitmap 0 has ~1/16 ones
	{
		std::vector<uint32_t> sparse((1u << 22) / 32);
		for (size_t i = 0; i < sparse.size(); i++) sparse[i] = 0x00010001u << (i % 16); // 2 of 32 bits
		uint32_t *dsp; CK(hipMalloc(&dsp, sparse.size() * 4)); CK(hipMemcpy(dsp, sparse.data(), sparse.size() * 4, hipMemcpyHostToDevice));
		const uint32_t *hb[4] = {dsp, dbits, dbits, dbits};
		uint32_t hm[4] = {(1u << 22) - 1, (1u << 22) - 1, (1u << 22) - 1, (1u << 16) - 1};
		const uint32_t **dbv; uint32_t *dm; CK(hipMalloc(&dbv, sizeof(hb))); CK(hipMalloc(&dm, sizeof(hm)));
		CK(hipMemcpy(dbv, hb, sizeof(hb), hipMemcpyHostToDevice)); CK(hipMemcpy(dm, hm, sizeof(hm), hipMemcpyHostToDevice));
		float lz = timeit([&] { hipLaunchKernelGGL((lazy_kernel<4>), dim3(grid), dim3(256), 0, 0, dcolv, n, dbv, dm, dout); }, 3);
		printf("lazy 4 stages, first keeps 1/16 (512KB bitmaps): %.3f ms  %.1f Gtuples/s\n", lz, n / lz / 1e6);
		const uint32_t *hb2[4] = {dbits, dbits, dbits, dbits};
		CK(hipMemcpy(dbv, hb2, sizeof(hb2), hipMemcpyHostToDevice));
		float lz2 = timeit([&] { hipLaunchKernelGGL((lazy_kernel<4>), dim3(grid), dim3(256), 0, 0, dcolv, n, dbv, dm, dout); }, 3);
		printf("lazy 4 stages, all alive (512KB bitmaps):        %.3f ms  %.1f Gtuples/s\n", lz2, n / lz2 / 1e6);
	}
	return 0;
}
