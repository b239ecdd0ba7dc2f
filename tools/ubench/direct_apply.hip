// microbenchmark: what PASS 1's op-side apply would cost -- per leader op H = 4 random counters of a 2 GB array:
//   gather      : read the four bytes (FOpTarget's probes today)
//   gather_store: read them and store a byte back to each that is below the target (the fused form)
//   blind_store : store four bytes from a mask, nothing read (the deferred form)
//   tile_stream : the array in and out through LDS tiles of 64 KB (tile_apply's streaming part alone)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void k_gather(const uint8_t* cnt, uint64_t m, uint64_t T, uint8_t* tgt)
{
	uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; if (t >= T) return;
	uint64_t h = mix(t + 1);
	unsigned c[4];
#pragma unroll
	for (int j = 0; j < 4; j++) c[j] = cnt[(h * (2 * j + 1) + (h >> (7 + j))) % m];
	unsigned mn = min(min(c[0], c[1]), min(c[2], c[3]));
	tgt[t] = (uint8_t)(mn + 1);
}
__global__ void k_gather_store(uint8_t* cnt, uint64_t m, uint64_t T, uint8_t* tgt)
{
	uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; if (t >= T) return;
	uint64_t h = mix(t + 1);
	unsigned c[4]; uint64_t p[4];
#pragma unroll
	for (int j = 0; j < 4; j++) { p[j] = (h * (2 * j + 1) + (h >> (7 + j))) % m; c[j] = cnt[p[j]]; }
	unsigned mn = min(min(c[0], c[1]), min(c[2], c[3]));
	unsigned tg = mn + 1 > 255 ? 255 : mn + 1;
	tgt[t] = (uint8_t)tg;
#pragma unroll
	for (int j = 0; j < 4; j++) if (c[j] < tg) cnt[p[j]] = (uint8_t)tg;
}
__global__ void k_blind_store(uint8_t* cnt, uint64_t m, uint64_t T, const uint8_t* tgt)
{
	uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; if (t >= T) return;
	unsigned tg = tgt[t]; if (!tg) return;
	uint64_t h = mix(t + 1);
#pragma unroll
	for (int j = 0; j < 4; j++) cnt[(h * (2 * j + 1) + (h >> (7 + j))) % m] = (uint8_t)tg;
}
__global__ void __launch_bounds__(1024) k_tile_stream(uint8_t* cnt, uint64_t m)
{
	__shared__ uint64_t l8[8192];
	uint64_t* g8 = (uint64_t*)(cnt + (uint64_t)blockIdx.x * 65536);
	for (int i = threadIdx.x; i < 8192; i += 1024) l8[i] = g8[i];
	__syncthreads();
	l8[(threadIdx.x * 37) & 8191] += 1;
	__syncthreads();
	for (int i = threadIdx.x; i < 8192; i += 1024) g8[i] = l8[i];
}
int main()
{
	const uint64_t m = 2ull << 30, T = 25ull << 20;
	uint8_t *cnt, *tgt;
	CK(hipMalloc(&cnt, m)); CK(hipMalloc(&tgt, T));
	CK(hipMemset(cnt, 1, m)); CK(hipMemset(tgt, 2, T));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const unsigned B = 256, G = (unsigned)((T + B - 1) / B);
	for (int rep = 0; rep < 3; rep++) {
		float ms;
		CK(hipEventRecord(a)); hipLaunchKernelGGL(k_gather, dim3(G), dim3(B), 0, 0, cnt, m, T, tgt); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
		printf("gather        %.3f ms\n", ms);
		CK(hipEventRecord(a)); hipLaunchKernelGGL(k_gather_store, dim3(G), dim3(B), 0, 0, cnt, m, T, tgt); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
		printf("gather_store  %.3f ms\n", ms);
		CK(hipEventRecord(a)); hipLaunchKernelGGL(k_blind_store, dim3(G), dim3(B), 0, 0, cnt, m, T, tgt); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
		printf("blind_store   %.3f ms\n", ms);
		CK(hipEventRecord(a)); hipLaunchKernelGGL(k_tile_stream, dim3((unsigned)(m / 65536)), dim3(1024), 0, 0, cnt, m); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
		printf("tile_stream   %.3f ms\n", ms);
	}
	return 0;
}
