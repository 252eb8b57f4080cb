// tools/micro/stream_bw.hip -- what the memory system of one MI355X gives the flat probe pipeline's REQUEST MIX when
// nothing else is in the way (no routing, no queues, no LDS tables): the ceiling the pool kernel is held against in
// DESIGN.md section 7.
//   stream   every lane loads 16 bytes, consecutive lanes consecutive addresses: one column read front to back
//   mix      a four-column pass shaped like SSB Q4.1 on the skewed table: column A is streamed (16 bytes per lane); a
//            tuple survives stage s with probability p[s] (a hash of its position) and only survivors read their key of
//            the next column, at the SAME row position (ascending, sparse: the pipeline's key gathers).  Lanes are not
//            compacted: the request stream (which lines are asked for, in which order) is the pipeline's, the VALU work
//            is not
// Prints GB/s of the bytes asked for (keys x 4) and of the lines touched (64-byte lines, counted exactly on the host
// from the same hash).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/stream_bw tools/micro/stream_bw.hip && tools/micro/stream_bw [rows_millions]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                    \
			exit(1);                                                                                                   \
		}                                                                                                              \
	} while (0)

__host__ __device__ inline uint32_t mix32(uint64_t x) {
	x ^= x >> 33;
	x *= 0xff51afd7ed558ccdull;
	x ^= x >> 33;
	x *= 0xc4ceb9fe1a85ec53ull;
	x ^= x >> 33;
	return (uint32_t)x;
}

__global__ __launch_bounds__(256) void bw_stream(const uint4 *__restrict__ src, uint64_t n16, unsigned long long *out) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for (; i + 3 * step < n16; i += 4 * step) {
		const uint4 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
		acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
	}
	for (; i < n16; i += step) {
		const uint4 a = src[i];
		acc += a.x ^ a.y ^ a.z ^ a.w;
	}
	if (acc == 0x12345678u) {
		atomicAdd(out, 1ull);
	}
}

// thresholds: a tuple at row r survives stage s iff mix32(r * 4 + s) < thr[s]
struct Thr {
	uint32_t t[3];
};

__global__ __launch_bounds__(256) void bw_mix(const uint4 *__restrict__ a, const uint32_t *__restrict__ b,
                                              const uint32_t *__restrict__ c, const uint32_t *__restrict__ d, uint64_t n16,
                                              Thr thr, unsigned long long *out) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for (; i < n16; i += step) {
		const uint4 ka = a[i];
		acc += ka.x ^ ka.y ^ ka.z ^ ka.w;
		uint32_t kb[4], kc[4], kd[4];
		bool s0[4], s1[4], s2[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const uint64_t r = i * 4 + j;
			s0[j] = mix32(r * 4 + 0) < thr.t[0];
			s1[j] = s0[j] && mix32(r * 4 + 1) < thr.t[1];
			s2[j] = s1[j] && mix32(r * 4 + 2) < thr.t[2];
			kb[j] = s0[j] ? b[r] : 0u;
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			kc[j] = s1[j] ? c[i * 4 + j] : 0u;
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			kd[j] = s2[j] ? d[i * 4 + j] : 0u;
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			acc += kb[j] ^ kc[j] ^ kd[j];
		}
	}
	if (acc == 0x12345678u) {
		atomicAdd(out, 1ull);
	}
}

static double time_ms(hipEvent_t e0, hipEvent_t e1) {
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	return ms;
}

int main(int argc, char **argv) {
	const uint64_t rows = (uint64_t)(argc > 1 ? atof(argv[1]) : 600.0) * 1000000ull / 4 * 4;
	const uint64_t n16 = rows / 4;
	uint32_t *col[4];
	for (int k = 0; k < 4; k++) {
		CHECK(hipMalloc(&col[k], rows * 4));
		CHECK(hipMemset(col[k], k + 1, rows * 4));
	}
	unsigned long long *out;
	CHECK(hipMalloc(&out, 8));
	CHECK(hipMemset(out, 0, 8));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("{\"rows\": %llu, \"cus\": %d, \"results\": [\n", (unsigned long long)rows, cus);
	bool first = true;
	const int reps = 5;
	for (int wg_per_cu : {2, 4, 8}) {
		const int grid = cus * wg_per_cu;
		bw_stream<<<grid, 256>>>((const uint4 *)col[0], n16, out);
		CHECK(hipDeviceSynchronize());
		CHECK(hipEventRecord(e0));
		for (int r = 0; r < reps; r++) {
			bw_stream<<<grid, 256>>>((const uint4 *)col[r % 4], n16, out);
		}
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		const double ms = time_ms(e0, e1) / reps;
		printf("%s {\"kernel\": \"stream\", \"waves_per_cu\": %d, \"ms\": %.4f, \"GBps\": %.1f}", first ? "" : ",\n", wg_per_cu * 4,
		       ms, rows * 4.0 / ms / 1e6);
		first = false;
	}
	// survivor fractions per stage: Q4.1-like (1/5, 1/5, 2/5) and two sparser mixes
	const double mixes[3][3] = {{0.2, 0.2, 0.4}, {0.08, 0.2, 0.4}, {0.5, 0.5, 0.5}};
	for (int mi = 0; mi < 3; mi++) {
		Thr thr;
		for (int s = 0; s < 3; s++) {
			thr.t[s] = (uint32_t)(mixes[mi][s] * 4294967295.0);
		}
		// exact line counts (64-byte lines = 16 keys) and key counts, on a 1/64 sample of the rows
		uint64_t keys[4] = {0, 0, 0, 0}, lines[4] = {0, 0, 0, 0};
		const uint64_t sample = rows / 64 / 16 * 16;
		for (uint64_t l = 0; l < sample / 16; l++) {
			bool touch[3] = {false, false, false};
			for (int j = 0; j < 16; j++) {
				const uint64_t r = l * 16 + j;
				const bool s0 = mix32(r * 4 + 0) < thr.t[0];
				const bool s1 = s0 && mix32(r * 4 + 1) < thr.t[1];
				const bool s2 = s1 && mix32(r * 4 + 2) < thr.t[2];
				keys[1] += s0;
				keys[2] += s1;
				keys[3] += s2;
				touch[0] |= s0;
				touch[1] |= s1;
				touch[2] |= s2;
			}
			for (int s = 0; s < 3; s++) {
				lines[s + 1] += touch[s];
			}
		}
		const double scale = (double)rows / (double)sample;
		const double key_bytes = rows * 4.0 + (keys[1] + keys[2] + keys[3]) * scale * 4.0;
		const double line_bytes = rows * 4.0 + (lines[1] + lines[2] + lines[3]) * scale * 64.0;
		for (int wg_per_cu : {4, 8}) {
			const int grid = cus * wg_per_cu;
			bw_mix<<<grid, 256>>>((const uint4 *)col[0], col[1], col[2], col[3], n16, thr, out);
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0));
			for (int r = 0; r < reps; r++) {
				bw_mix<<<grid, 256>>>((const uint4 *)col[0], col[1], col[2], col[3], n16, thr, out);
			}
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			const double ms = time_ms(e0, e1) / reps;
			printf(",\n {\"kernel\": \"mix\", \"survive\": [%.2f, %.2f, %.2f], \"waves_per_cu\": %d, \"ms\": %.4f, \"key_bytes\": %.0f, "
			       "\"line_bytes\": %.0f, \"key_GBps\": %.1f, \"line_GBps\": %.1f}",
			       mixes[mi][0], mixes[mi][1], mixes[mi][2], wg_per_cu * 4, ms, key_bytes, line_bytes, key_bytes / ms / 1e6,
			       line_bytes / ms / 1e6);
		}
	}
	printf("\n]}\n");
	return 0;
}
