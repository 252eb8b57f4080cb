// tools/micro/probe_shapes.hip -- design microbenchmark (not product code): what bounds "selection -> key gather -> unique-key
// hash probe -> count" on gfx950, by the shape of the loop?  The data are those of the JOB 18a shape's first join
// (3.6 M selected rows of a 36 M-row key column, 10 % density, a 20 k-key table of 64 Ki 8-byte slots).
//   hipcc --offload-arch=gfx950 -O3 -o probe_shapes probe_shapes.hip && ./probe_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#define GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline uint64_t mh64(uint64_t x) {
	x ^= x >> 32; x *= 0xd6e8feb86659fd93ULL; x ^= x >> 32; x *= 0xd6e8feb86659fd93ULL; x ^= x >> 32; return x;
}
__host__ __device__ inline uint32_t mh32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

struct Args {
	const uint32_t *sel; const uint32_t *keys; const uint32_t *tab; uint32_t mask; uint64_t n; unsigned long long *out;
	uint32_t unit; // tuples per unit (a wave takes units round-robin)
};

// MODE bits: 1 = through sel (else dense rows), 2 = probe with ONE 16-byte load (2 slots) per round trip (else two = 4 slots),
//            4 = 32-bit hash, 8 = no probe at all (count keys & 1), 16 = 8-byte single-slot probe
template <int F, int MODE>
__global__ __launch_bounds__(512, 4) void probe_kernel(Args a) {
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	const GLOBAL uint32_t *sel = (const GLOBAL uint32_t *)a.sel;
	const GLOBAL uint32_t *keys = (const GLOBAL uint32_t *)a.keys;
	const GLOBAL uint32_t *tab = (const GLOBAL uint32_t *)a.tab;
	uint32_t cnt = 0;
	for (uint64_t u0 = wave * a.unit; u0 < a.n; u0 += n_waves * a.unit) {
		const uint64_t u1 = u0 + a.unit < a.n ? u0 + a.unit : a.n;
		for (uint64_t p = u0; p < u1; p += 64 * F) {
			uint32_t row[F], key[F];
			bool act[F];
#pragma unroll
			for (int i = 0; i < F; i++) {
				const uint64_t tp = p + 64 * i + lane;
				act[i] = tp < u1;
				row[i] = act[i] ? ((MODE & 1) ? sel[tp] : (uint32_t)tp) : 0u;
			}
#pragma unroll
			for (int i = 0; i < F; i++) key[i] = act[i] ? keys[row[i]] : 0u;
			if (MODE & 8) {
#pragma unroll
				for (int i = 0; i < F; i++) cnt += (act[i] && (key[i] & 1)) ? 1 : 0;
				continue;
			}
			uint32_t slot[F];
			bool searching[F], any = false;
#pragma unroll
			for (int i = 0; i < F; i++) {
				slot[i] = ((MODE & 4) ? mh32(key[i]) : (uint32_t)mh64(key[i])) & a.mask;
				searching[i] = act[i];
				any = any || act[i];
			}
			while (__ballot(any)) {
				any = false;
				if (MODE & 16) {
					u32x2 e[F];
#pragma unroll
					for (int i = 0; i < F; i++) if (searching[i]) e[i] = *(const GLOBAL u32x2 *)(tab + (uint64_t)slot[i] * 2);
#pragma unroll
					for (int i = 0; i < F; i++) if (searching[i]) {
						if (e[i].y == 0xFFFFFFFFu) searching[i] = false;
						else if (e[i].x == key[i]) { cnt++; searching[i] = false; }
						slot[i] = (slot[i] + 1) & a.mask;
						any = any || searching[i];
					}
				} else if (MODE & 2) {
					u32x4 e[F];
#pragma unroll
					for (int i = 0; i < F; i++) if (searching[i]) e[i] = *(const GLOBAL u32x4 *)(tab + (uint64_t)(slot[i] & ~1u) * 2);
#pragma unroll
					for (int i = 0; i < F; i++) if (searching[i]) {
						const uint32_t kk[2] = {e[i].x, e[i].z}, rr[2] = {e[i].y, e[i].w};
#pragma unroll
						for (int j = 0; j < 2; j++) if (searching[i] && (uint32_t)j >= (slot[i] & 1u)) {
							if (rr[j] == 0xFFFFFFFFu) searching[i] = false;
							else if (kk[j] == key[i]) { cnt++; searching[i] = false; }
						}
						slot[i] = ((slot[i] & ~1u) + 2) & a.mask;
						any = any || searching[i];
					}
				} else {
					u32x4 e0[F], e1[F];
#pragma unroll
					for (int i = 0; i < F; i++) if (searching[i]) {
						e0[i] = *(const GLOBAL u32x4 *)(tab + (uint64_t)(slot[i] & ~3u) * 2);
						e1[i] = *(const GLOBAL u32x4 *)(tab + (uint64_t)(slot[i] & ~3u) * 2 + 4);
					}
#pragma unroll
					for (int i = 0; i < F; i++) if (searching[i]) {
						const uint32_t kk[4] = {e0[i].x, e0[i].z, e1[i].x, e1[i].z}, rr[4] = {e0[i].y, e0[i].w, e1[i].y, e1[i].w};
#pragma unroll
						for (int j = 0; j < 4; j++) if (searching[i] && (uint32_t)j >= (slot[i] & 3u)) {
							if (rr[j] == 0xFFFFFFFFu) searching[i] = false;
							else if (kk[j] == key[i]) { cnt++; searching[i] = false; }
						}
						slot[i] = ((slot[i] & ~3u) + 4) & a.mask;
						any = any || searching[i];
					}
				}
			}
		}
	}
	unsigned long long c = cnt;
	for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
	if (lane == 0 && c) atomicAdd(a.out, c);
}

template <int F, int MODE>
static void run(const char *name, Args a, int blocks, uint32_t unit) {
	a.unit = unit;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e9;
	unsigned long long got = 0;
	for (int r = 0; r < 6; r++) {
		CK(hipMemset(a.out, 0, 8));
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((probe_kernel<F, MODE>), dim3(blocks), dim3(512), 0, 0, a);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		if (r) best = std::min(best, ms);
		CK(hipMemcpy(&got, a.out, 8, hipMemcpyDeviceToHost));
	}
	printf("%-58s F=%d blocks=%4d unit=%6u: %8.1f us  %7.2f G tuples/s  (count %llu)\n", name, F, blocks, unit, best * 1e3, a.n / best / 1e6, got);
}

int main() {
	const uint64_t n_rows = 36244344, n_name = 4167491;
	std::vector<uint32_t> keys(n_rows), sel;
	uint64_t s = 88172645463325252ull;
	auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
	for (uint64_t i = 0; i < n_rows; i++) {
		keys[i] = 1 + (uint32_t)(rnd() % n_name);
		if (rnd() % 10 == 0) sel.push_back((uint32_t)i);
	}
	const uint32_t cap = 65536;
	std::vector<uint32_t> tab(cap * 2, 0xFFFFFFFFu);
	uint32_t inserted = 0;
	for (uint32_t k = 1; k <= n_name && inserted < 20552; k++) {
		if (rnd() % 200 != 0) continue;
		uint32_t h = (uint32_t)mh64(k) & (cap - 1);
		while (tab[h * 2 + 1] != 0xFFFFFFFFu) h = (h + 1) & (cap - 1);
		tab[h * 2] = k; tab[h * 2 + 1] = inserted++;
	}
	std::vector<uint32_t> tab32(cap * 2, 0xFFFFFFFFu);
	for (uint32_t i = 0; i < cap; i++) if (tab[i * 2 + 1] != 0xFFFFFFFFu) {
		uint32_t k = tab[i * 2];
		uint32_t h = mh32(k) & (cap - 1);
		while (tab32[h * 2 + 1] != 0xFFFFFFFFu) h = (h + 1) & (cap - 1);
		tab32[h * 2] = k; tab32[h * 2 + 1] = tab[i * 2 + 1];
	}
	printf("rows %llu, selected %zu, table %u keys in %u slots\n", (unsigned long long)n_rows, sel.size(), inserted, cap);
	uint32_t *d_keys, *d_sel, *d_tab, *d_tab32; unsigned long long *d_out;
	CK(hipMalloc(&d_keys, n_rows * 4)); CK(hipMalloc(&d_sel, sel.size() * 4)); CK(hipMalloc(&d_tab, cap * 8)); CK(hipMalloc(&d_tab32, cap * 8));
	CK(hipMalloc(&d_out, 8));
	CK(hipMemcpy(d_keys, keys.data(), n_rows * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_sel, sel.data(), sel.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_tab, tab.data(), cap * 8, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_tab32, tab32.data(), cap * 8, hipMemcpyHostToDevice));
	Args a {d_sel, d_keys, d_tab, cap - 1, sel.size(), d_out, 256};
	Args a32 = a; a32.tab = d_tab32;
	Args dense = a; dense.n = sel.size(); // the first 3.6 M rows, no selection
	for (int blocks : {512, 256}) {
		for (uint32_t unit : {256u, 2048u, 65536u}) {
			run<4, 1>("sel -> key -> 2 x 16 B probe (the pipeline's shape)", a, blocks, unit);
		}
	}
	run<4, 1 | 8>("sel -> key, no probe", a, 512, 2048);
	run<4, 8>("dense key, no probe", dense, 512, 2048);
	run<4, 0>("dense key -> 2 x 16 B probe", dense, 512, 2048);
	run<4, 1 | 2>("sel -> key -> 1 x 16 B probe", a, 512, 2048);
	run<4, 1 | 16>("sel -> key -> 8 B probe", a, 512, 2048);
	run<4, 1 | 2 | 4>("sel -> key -> 1 x 16 B probe, 32-bit hash", a32, 512, 2048);
	run<8, 1 | 2>("sel -> key -> 1 x 16 B probe", a, 512, 2048);
	run<8, 1>("sel -> key -> 2 x 16 B probe", a, 512, 2048);
	run<2, 1>("sel -> key -> 2 x 16 B probe", a, 512, 2048);
	run<1, 1>("sel -> key -> 2 x 16 B probe", a, 512, 2048);
	run<8, 1 | 16 | 4>("sel -> key -> 8 B probe, 32-bit hash", a32, 512, 2048);
	return 0;
}
