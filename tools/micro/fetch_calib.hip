// tools/micro/fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the two access patterns of the flat
// probe pipeline (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern"):
//   stream   every lane loads 16 bytes, consecutive lanes consecutive addresses (stage 0's key stream)
//   gather   lane j loads the 4-byte element at  j * stride + hash(j) % stride  (ascending positions, one survivor per
//            `stride` tuples on average: a deeper stage's key gather); the lines it touches are counted exactly
// Prints the known byte counts; run it under `rocprofv3 --pmc FETCH_SIZE` and compare (tools/calib_fetch.sh).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/fetch_calib tools/micro/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                    \
			exit(1);                                                                                                   \
		}                                                                                                              \
	} while (0)

__host__ __device__ inline uint32_t mix(uint64_t x) {
	x ^= x >> 33;
	x *= 0xff51afd7ed558ccdull;
	x ^= x >> 33;
	x *= 0xc4ceb9fe1a85ec53ull;
	x ^= x >> 33;
	return (uint32_t)x;
}

__global__ void calib_stream(const uint4 *__restrict__ src, uint64_t n16, unsigned long long *out) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for (; i < n16; i += step) {
		const uint4 v = src[i];
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) {
		atomicAdd(out, 1ull);
	}
}

__global__ void calib_gather(const uint32_t *__restrict__ src, uint64_t m, uint32_t stride, unsigned long long *out) {
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for (; j < m; j += step) {
		acc += src[j * stride + mix(j) % stride];
	}
	if (acc == 0x12345678u) {
		atomicAdd(out, 1ull);
	}
}

// marks the 64-byte lines the gather touches (one bit per line) -- the exact count is the known traffic
__global__ void calib_mark(uint32_t *bits, uint64_t m, uint32_t stride) {
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	for (; j < m; j += step) {
		const uint64_t line = (j * stride + mix(j) % stride) >> 4;
		atomicOr(&bits[line >> 5], 1u << (line & 31));
	}
}
__global__ void calib_popc(const uint32_t *bits, uint64_t words, unsigned long long *out) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	unsigned long long c = 0;
	for (; i < words; i += step) {
		c += __popc(bits[i]);
	}
	atomicAdd(out, c);
}

int main(int argc, char **argv) {
	const char *mode = argc > 1 ? argv[1] : "stream";
	const uint32_t stride = argc > 2 ? (uint32_t)atoi(argv[2]) : 5;
	const uint64_t n = 1ull << 29; // 2^29 elements = 2 GiB, far beyond the 256 MiB infinity cache
	uint32_t *src = nullptr;
	unsigned long long *out = nullptr;
	CHECK(hipMalloc((void **)&src, n * 4));
	CHECK(hipMalloc((void **)&out, 8));
	CHECK(hipMemset(src, 1, n * 4));
	CHECK(hipMemset(out, 0, 8));
	CHECK(hipDeviceSynchronize());
	const dim3 grid(256 * 8), block(1024);
	if (!strcmp(mode, "stream")) {
		for (int rep = 0; rep < 3; rep++) {
			hipLaunchKernelGGL(calib_stream, grid, block, 0, 0, (const uint4 *)src, n / 4, out);
		}
		CHECK(hipDeviceSynchronize());
		printf("{\"mode\": \"stream\", \"kernel\": \"calib_stream\", \"launches\": 3, \"known_bytes_per_launch\": %llu}\n",
		       (unsigned long long)(n * 4));
	} else {
		const uint64_t m = n / stride;
		uint32_t *bits = nullptr;
		const uint64_t words = (n / 16 + 31) / 32;
		CHECK(hipMalloc((void **)&bits, words * 4));
		CHECK(hipMemset(bits, 0, words * 4));
		hipLaunchKernelGGL(calib_mark, grid, block, 0, 0, bits, m, stride);
		hipLaunchKernelGGL(calib_popc, grid, block, 0, 0, (const uint32_t *)bits, words, out);
		unsigned long long lines = 0;
		CHECK(hipMemcpy(&lines, out, 8, hipMemcpyDeviceToHost));
		CHECK(hipMemset(out, 0, 8));
		for (int rep = 0; rep < 3; rep++) {
			hipLaunchKernelGGL(calib_gather, grid, block, 0, 0, (const uint32_t *)src, m, stride, out);
		}
		CHECK(hipDeviceSynchronize());
		printf("{\"mode\": \"gather\", \"stride\": %u, \"kernel\": \"calib_gather\", \"launches\": 3, \"elements\": %llu, "
		       "\"lines_touched\": %llu, \"known_bytes_per_launch\": %llu, \"requested_bytes_per_launch\": %llu}\n",
		       stride, (unsigned long long)m, lines, lines * 64ull, (unsigned long long)(m * 4));
	}
	return 0;
}
