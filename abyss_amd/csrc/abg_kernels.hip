// abg_kernels.hip -- the gfx950 kernels and the C ABI (include/abyss_amd.h).
//
// Kernels are grid-stride wrappers around the functors of abg_engine.h / abg_walk.h:
//   k_foreach      one item per lane (hash, claim, insert rounds, classify, predict, ...)
//   k_foreach_w    the same, launched one wave per workgroup so that the <=32768 walkers
//                  spread over all 256 CUs (each walker is a latency-bound pointer chase)
//   k_commit       one 256-thread workgroup: the ordered commit, cooperative per contig
// Launch geometry: wave = 64 lanes, workgroups of 256 (4 waves), grids capped at
// 256 CUs x 8 workgroups and grid-strided beyond that.
//
// Built with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>

#include "abg_host.h"
#include "abg_overlap.h"

#include <chrono>
#include <map>
#include <mutex>

namespace {

template <class F>
__global__ void __launch_bounds__(256) k_foreach(F f, uint64_t n)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	const uint32_t slot = (uint32_t)i;
	for (; i < n; i += stride) f(i, slot);
}
#ifndef ABG_CLS_WAVES
#define ABG_CLS_WAVES 3 // wavefronts per SIMD the one-item-per-lane kernels of k_foreach_w (FClassify) are compiled for at least: 170 VGPRs instead of the 180 the compiler takes unasked (two waves); 985 vs 1038 ms per configs[1] step, four and more waves lose to their spills
#endif
template <class F>
__global__ void __launch_bounds__(64, ABG_CLS_WAVES) k_foreach_w(F f, uint64_t n)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	const uint32_t slot = (uint32_t)i;
	for (; i < n; i += stride) f(i, slot);
}

// One walker per wavefront.  A unitig walk is a long, branchy, dependent chain; 64 of them
// in one wave would serialise on every divergent branch.  Instead all 64 lanes of a wave
// run ONE walker in lock step with identical state (wave-uniform control flow), and the
// parts with data parallelism use the lanes: each lane computes one Bloom probe position
// (k-mer x hash function) and a ballot classifies the 8 neighbours of the head; the
// trueBranch on-stack test scans 64 frames at a time; atomics go through lane 0.
// The walker's trueBranch stack (frames + keys) lives in WALK_LDS bytes of LDS.
#ifndef ABG_WALK_WAVES
#define ABG_WALK_WAVES 3 // wavefronts per SIMD the walker kernel is compiled for (register budget 512 / waves): 12 walkers per CU
#endif
#ifndef ABG_WALK_LDS
#define ABG_WALK_LDS 13312 // (160 KB / 12 walkers.  Two per SIMD with 20 KB each -- room for the walkers' neighbour-mask cache --
                           // measured 1.19 s per configs[1] step against 1.16 s for three with less: concurrency wins)
#endif
constexpr uint32_t WALK_LDS = ABG_WALK_LDS;
template <class F>
__global__ void __launch_bounds__(64, ABG_WALK_WAVES) k_walkers(F f, uint64_t n, unsigned long long* ticket)
{
	__shared__ __attribute__((aligned(16))) unsigned char lds[WALK_LDS];
	{
		// the walkers' neighbour-mask cache at the end of the block starts out empty (see FWalk)
		abg::MaskCache* mc = (abg::MaskCache*)(lds + WALK_LDS - sizeof(abg::MaskCache));
		for (unsigned i = threadIdx.x; i < abg::MC_N; i += 64) mc->valid[i] = 0;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	}
	// walks differ in length by orders of magnitude: waves draw candidates from a ticket
	// counter (in candidate order) instead of striding, so no wave is left with a queue of
	// long walks while others idle
	for (;;) {
		unsigned long long i = 0;
		if (threadIdx.x == 0) i = atomicAdd(ticket, 1ull);
		i = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(i >> 32)) << 32) |
		    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)i);
		if (i >= n) break;
		f(i, (uint32_t)blockIdx.x, (void*)lds, WALK_LDS, true);
	}
}

// one item per wavefront, all 64 lanes cooperate (f strides its inner loop by lane)
template <class F>
__global__ void __launch_bounds__(256) k_foreach_wave(F f, uint64_t n)
{
	const uint32_t lane = threadIdx.x & 63;
	uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	for (; w < n; w += nw) f(w, lane, 64u);
}

constexpr int COMMIT_THREADS = 256;
constexpr int SYNC_WORDS = 34; // up to 16 waves (1024 threads) + broadcast slot at [32]
struct DeviceSync {
	uint32_t* sh; // [SYNC_WORDS] shared words: one per wave + a broadcast slot
	abg::CommitDesc* dsc = nullptr; // [COMMIT_CHUNK] in LDS (commit kernel only)
	__device__ abg::CommitDesc* descs() { return dsc; }
	__device__ uint32_t tid() const { return threadIdx.x; }
	__device__ uint32_t nthreads() const { return blockDim.x; }
	__device__ void barrier() { __syncthreads(); }
	__device__ bool all(bool v) { return __syncthreads_and(v ? 1 : 0) != 0; }
	__device__ uint32_t sum(uint32_t v)
	{
		for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
		__syncthreads();
		if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
		__syncthreads();
		uint32_t t = 0;
		for (unsigned w = 0; w < blockDim.x / 64; w++) t += sh[w];
		__syncthreads();
		return t;
	}
	__device__ bool any(bool v) { return __syncthreads_or(v ? 1 : 0) != 0; }
	// ascending bitonic sort of keys[0, n) in place (the array has room for the next power of two)
	__device__ void sort_u32(uint32_t* keys, uint32_t n)
	{
		uint32_t P = 1;
		while (P < n) P <<= 1;
		for (uint32_t i = n + threadIdx.x; i < P; i += blockDim.x) keys[i] = 0xFFFFFFFFu;
		__syncthreads();
		for (uint32_t k = 2; k <= P; k <<= 1)
			for (uint32_t j = k >> 1; j > 0; j >>= 1) {
				for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
					const uint32_t x = i ^ j;
					if (x > i) {
						const uint32_t a = keys[i], b = keys[x];
						if ((a > b) == ((i & k) == 0)) { keys[i] = b; keys[x] = a; }
					}
				}
				__syncthreads();
			}
	}
	__device__ uint32_t bcast(uint32_t v)
	{
		__syncthreads();
		if (threadIdx.x == 0) sh[32] = v;
		__syncthreads();
		uint32_t r = sh[32];
		__syncthreads();
		return r;
	}
};
template <int NW>
__global__ void __launch_bounds__(COMMIT_THREADS) k_commit(abg::CommitEnv<NW> e, uint32_t c_begin, uint32_t c_end)
{
	__shared__ uint32_t sh[SYNC_WORDS];
	__shared__ abg::CommitDesc dsc[abg::COMMIT_CHUNK];
	DeviceSync sy{ sh, dsc };
	abg::commit_candidates<NW>(e, c_begin, c_end, sy);
}

// one workgroup per tile of the counter array (abg::TileEnv): F::FAST bytes of LDS, tiles strided over the grid
template <class F>
__global__ void __launch_bounds__(F::THREADS) k_tiles(F f, uint64_t n)
{
	__shared__ __attribute__((aligned(16))) unsigned char lds[F::FAST]; // (all of the 64 KB a workgroup may declare, for tile_apply)
	DeviceSync sy{ nullptr }; // (the tile procedures use barrier / any / sort_u32 only: no shared words)
	for (uint64_t tile = blockIdx.x; tile < n; tile += gridDim.x) {
		f(tile, (void*)lds, sy);
		__syncthreads();
	}
}

__global__ void __launch_bounds__(1024) k_insert_drain(abg::InsertDrainEnv e)
{
	__shared__ uint32_t sh[SYNC_WORDS];
	DeviceSync sy{ sh };
	abg::insert_drain(e, sy);
}

struct ProfEntry { double ms = 0; uint64_t launches = 0; };

struct HipBackend {
	static constexpr uint32_t ITEM_GROUP_LOG2 = 6; // launch(): a wavefront runs 64 consecutive items (k_foreach)
	int device = 0;
	bool good = false;
	std::string reason;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr; // side stream: work that may overlap the main stream's kernels (launch_side)
	hipEvent_t ev2a = nullptr, ev2b = nullptr;
	std::string side_name;
	bool side_pending = false;
	bool profiling = false;
	std::map<std::string, ProfEntry> prof;
	uint32_t cus = 256;
	unsigned long long* ticket = nullptr;

	explicit HipBackend(int dev = 0) : device(dev)
	{
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0) { reason = "no HIP device available (abyss_amd has no CPU fallback)"; return; }
		if (dev < 0 || dev >= n) { reason = "HIP device ordinal out of range"; return; }
		if (hipSetDevice(dev) != hipSuccess) { reason = "hipSetDevice failed"; return; }
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = (uint32_t)prop.multiProcessorCount;
		if (hipStreamCreate(&stream) != hipSuccess) { reason = "hipStreamCreate failed"; return; }
		if (const char* e = getenv("ABG_MEM_LIMIT_MB")) mem_limit = (size_t)strtoull(e, 0, 10) << 20;
		{
			// the side stream only fills gaps: lowest priority (ABG_SIDE_PRIORITY=0 turns that off)
			int lo = 0, hi = 0;
			hipDeviceGetStreamPriorityRange(&lo, &hi);
			const char* e = getenv("ABG_SIDE_PRIORITY");
			const bool low = !e || atoi(e) != 0;
			if (hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, low ? lo : hi) != hipSuccess) { reason = "hipStreamCreate failed"; return; }
		}
		hipEventCreate(&ev2a);
		hipEventCreate(&ev2b);
		good = true;
	}
	~HipBackend()
	{
		if (ticket) free(ticket);
		if (pin) hipHostFree(pin);
		for (int i = 0; i < 2; i++) { if (big_pin[i]) hipHostFree(big_pin[i]); if (big_ev[i]) hipEventDestroy(big_ev[i]); }
		for (int i = 0; i < 2; i++) { if (ahead_pin[i]) hipHostFree(ahead_pin[i]); if (ahead_ev[i]) hipEventDestroy(ahead_ev[i]); if (ahead[i].dev) hipFree(ahead[i].dev); }
		for (void* q : ahead_old) hipFree(q);
		if (ahead_stream) hipStreamDestroy(ahead_stream);
		if (cub_tmp) hipFree(cub_tmp);
		drop_cache();
		prof_drain(true);
		for (auto& q : prof_free) { hipEventDestroy(q.first); hipEventDestroy(q.second); }
		if (ev2a) hipEventDestroy(ev2a);
		if (ev2b) hipEventDestroy(ev2b);
		if (stream2) hipStreamDestroy(stream2);
		if (stream) hipStreamDestroy(stream);
	}
	bool ok() const { return good; }
	std::string why() const { return reason; }
	void bind_thread() { (void)hipSetDevice(device); } // a thread the library made itself: the device is per thread
	static void check(hipError_t e, const char* what)
	{
		if (e != hipSuccess) {
			(void)hipGetLastError();
			abg::fail_now(e == hipErrorOutOfMemory ? abg::FAIL_NOMEM : abg::FAIL_INTERNAL, std::string(what) + " failed: " + hipGetErrorString(e));
		}
	}
	// Scratch buffers come and go with every batch; hipMalloc / hipFree cost far more than the
	// kernels between them (and hipFree synchronises the device).  Blocks up to 256 MiB are
	// rounded up to a power of two and recycled; all work is ordered on one stream, so a block
	// may be handed out again while its previous user is still queued.
	static constexpr size_t CACHE_MAX_BLOCK = 256ull << 20, CACHE_MAX_TOTAL = 8ull << 30;
	std::map<size_t, std::vector<void*>> cache_free;
	std::map<void*, size_t> cache_size;
	size_t cache_held = 0;
	// for optional big buffers: NULL instead of aborting when the device has no room
	// ABG_MEM_LIMIT_MB: a device memory budget for this context (tests: "the device is too small" without
	// filling a 288 GB device); a request that would exceed it fails like a hipMalloc that found no room
	size_t mem_limit = 0, mem_used = 0;
	std::map<void*, size_t> mem_size;
	hipError_t dev_malloc(void** p, size_t n)
	{
		if (mem_limit && mem_used + n > mem_limit) return hipErrorOutOfMemory;
		const hipError_t e = hipMalloc(p, n);
		if (e == hipSuccess && mem_limit) { mem_used += n; mem_size[*p] = n; }
		return e;
	}
	void dev_free(void* p)
	{
		if (mem_limit) { auto it = mem_size.find(p); if (it != mem_size.end()) { mem_used -= it->second; mem_size.erase(it); } }
		hipFree(p);
	}
	void* try_alloc(size_t n)
	{
		void* p = nullptr;
		hipSetDevice(device);
		if (dev_malloc(&p, n ? n : 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
		if (getenv("ABG_MEM_DEBUG")) fprintf(stderr, "[mem] +%.2f GB (optional)\n", n / 1e9);
		return p;
	}
	void* alloc(size_t n)
	{
		void* p = nullptr;
		hipSetDevice(device);
		if (n > CACHE_MAX_BLOCK) {
			hipError_t e = dev_malloc(&p, n);
			if (e != hipSuccess) {
				// (give back what the block cache holds and try once more before giving up)
				(void)hipGetLastError();
				drop_cache();
				e = dev_malloc(&p, n);
			}
			if (e != hipSuccess) out_of_memory(n);
			static const bool mem_debug = getenv("ABG_MEM_DEBUG") != nullptr;
			if (mem_debug) {
				size_t fr = 0, tot = 0;
				(void)hipMemGetInfo(&fr, &tot);
				fprintf(stderr, "[mem] +%.2f GB, %.2f GB free\n", n / 1e9, fr / 1e9);
			}
			return p;
		}
		size_t sz = 256;
		while (sz < n) sz <<= 1;
		auto it = cache_free.find(sz);
		if (it != cache_free.end() && !it->second.empty()) {
			p = it->second.back();
			it->second.pop_back();
			cache_held -= sz;
			return p;
		}
		if (dev_malloc(&p, sz) != hipSuccess) {
			(void)hipGetLastError();
			drop_cache();
			if (dev_malloc(&p, sz) != hipSuccess) out_of_memory(sz);
		}
		cache_size[p] = sz;
		return p;
	}
	[[noreturn]] void out_of_memory(size_t n)
	{
		size_t fr = 0, tot = 0;
		(void)hipMemGetInfo(&fr, &tot);
		char msg[200];
		snprintf(msg, sizeof msg, "no device memory for a block of %.2f GB (%.2f of %.2f GB free%s)", n / 1e9, fr / 1e9, tot / 1e9,
		    mem_limit ? ", ABG_MEM_LIMIT_MB in force" : "");
		(void)hipGetLastError();
		abg::fail_now(abg::FAIL_NOMEM, msg);
	}
	// `span` bytes that answer to the addresses [base + lo, base + lo + span) of an array of `total` bytes no device holds as a
	// whole (the sliced counting filter): returns base
	void* alloc_window(uint64_t total, uint64_t lo, uint64_t span) { (void)total; return (char*)alloc(span) - lo; }
	void free_window(void* base, uint64_t total, uint64_t lo, uint64_t span) { (void)total; (void)span; free((char*)base + lo); }
	void free(void* p)
	{
		if (!p) return;
		auto it = cache_size.find(p);
		if (it != cache_size.end() && cache_held + it->second <= CACHE_MAX_TOTAL) {
			cache_free[it->second].push_back(p);
			cache_held += it->second;
			return;
		}
		if (it != cache_size.end()) cache_size.erase(it);
		hipStreamSynchronize(stream);
		dev_free(p);
	}
	void drop_cache()
	{
		hipStreamSynchronize(stream);
		for (auto& kv : cache_free)
			for (void* q : kv.second) { cache_size.erase(q); dev_free(q); }
		cache_free.clear();
		cache_held = 0;
	}
	void memset(void* p, int v, size_t n) { check(hipMemsetAsync(p, v, n, stream), "hipMemsetAsync"); }
	// Small transfers (counters read back after nearly every kernel of a batch, little tables going
	// up) pass through a pinned staging buffer: a copy to or from pageable memory is staged by the
	// runtime at several times the cost.
	static constexpr size_t PIN_BYTES = 1u << 16;
	void* pin = nullptr;
	void h2d(void* d, const void* s, size_t n)
	{
		if (!n) return;
		if (n <= PIN_BYTES && (pin || hipHostMalloc(&pin, PIN_BYTES, hipHostMallocDefault) == hipSuccess)) {
			memcpy(pin, s, n);
			check(hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, stream), "hipMemcpy H2D");
			check(hipStreamSynchronize(stream), "hipStreamSynchronize"); // (the buffer is reused right away)
			return;
		}
		// Large uploads (a chunk's packed reads, their offsets and lengths) come from pageable memory, which the runtime stages
		// at 5-50 ms per 64 MB: through two pinned buffers of our own instead, a piece copied in while the one before it is on
		// its way -- the rate of a memcpy.
		if (n >= (1u << 20) && big_pin_ok()) {
			const char* src = (const char*)s;
			size_t done = 0;
			for (int i = 0; done < n; i ^= 1) {
				const size_t m = n - done < BIG_PIN ? n - done : BIG_PIN;
				check(hipEventSynchronize(big_ev[i]), "hipEventSynchronize"); // (the buffer's last copy has left)
				memcpy(big_pin[i], src + done, m);
				check(hipMemcpyAsync((char*)d + done, big_pin[i], m, hipMemcpyHostToDevice, stream), "hipMemcpy H2D");
				check(hipEventRecord(big_ev[i], stream), "hipEventRecord");
				done += m;
			}
			check(hipStreamSynchronize(stream), "hipStreamSynchronize");
			return;
		}
		check(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream), "hipMemcpy H2D");
		check(hipStreamSynchronize(stream), "hipStreamSynchronize");
	}
	// A chunk's arrays uploaded AHEAD (abg_host.h, load_seqs_v): by the CALLING thread while the library's own thread is still busy with
	// the chunk before -- on a stream, through pinned buffers and into one of two device blocks that belong to this alone (nothing
	// here shares state with alloc / h2d / the main stream, which that other thread is using meanwhile).  `n` arrays, each placed on
	// a 256-byte boundary; at[i] = where array i went.  The block is the caller's until the second call after this one.  nullptr:
	// not available (the caller uploads the ordinary way).
	struct Ahead { void* dev = nullptr; size_t cap = 0; } ahead[2];
	std::vector<void*> ahead_old;
	int ahead_next = 0, ahead_state = 0; // state: 0 untried, 1 ready, -1 unavailable
	hipStream_t ahead_stream = nullptr; void* ahead_pin[2] = { nullptr, nullptr }; hipEvent_t ahead_ev[2] = { nullptr, nullptr };
	void* upload_ahead(const void* const* src, const size_t* bytes, int n, void** at)
	{
		if (ahead_state < 0 || getenv("ABG_NO_UPLOAD_AHEAD")) return nullptr;
		(void)hipSetDevice(device);
		if (ahead_state == 0) {
			ahead_state = -1;
			if (hipStreamCreateWithFlags(&ahead_stream, hipStreamNonBlocking) == hipSuccess &&
			    hipHostMalloc(&ahead_pin[0], BIG_PIN, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&ahead_pin[1], BIG_PIN, hipHostMallocDefault) == hipSuccess &&
			    hipEventCreateWithFlags(&ahead_ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ahead_ev[1], hipEventDisableTiming) == hipSuccess)
				ahead_state = 1;
			else { (void)hipGetLastError(); return nullptr; }
		}
		size_t total = 0;
		for (int i = 0; i < n; i++) total += (bytes[i] + 255) & ~(size_t)255;
		Ahead& a = ahead[ahead_next];
		if (total > a.cap) {
			// (a block that has become too small is kept until the backend goes: hipFree waits for the whole device, which the
			// other thread is keeping busy -- and the first chunks of a run double in size, so what is kept is less than what is used)
			const size_t a_cap_was = a.cap;
			if (a.dev) { ahead_old.push_back(a.dev); a.dev = nullptr; a.cap = 0; }
			const size_t cap = std::max(total + total / 8, 2 * a_cap_was); // (at least doubled: a handful of blocks ever)
			if (hipMalloc(&a.dev, cap) != hipSuccess) { (void)hipGetLastError(); a.dev = nullptr; return nullptr; }
			a.cap = cap;
		}
		ahead_next ^= 1;
		size_t off = 0;
		int b = 0;
		for (int i = 0; i < n; i++) {
			at[i] = (char*)a.dev + off;
			const char* s = (const char*)src[i];
			for (size_t done = 0; done < bytes[i]; b ^= 1) {
				const size_t m = bytes[i] - done < BIG_PIN ? bytes[i] - done : BIG_PIN;
				check(hipEventSynchronize(ahead_ev[b]), "hipEventSynchronize"); // (the buffer's last copy has left)
				memcpy(ahead_pin[b], s + done, m);
				check(hipMemcpyAsync((char*)at[i] + done, ahead_pin[b], m, hipMemcpyHostToDevice, ahead_stream), "hipMemcpy H2D");
				check(hipEventRecord(ahead_ev[b], ahead_stream), "hipEventRecord");
				done += m;
			}
			off += (bytes[i] + 255) & ~(size_t)255;
		}
		check(hipStreamSynchronize(ahead_stream), "hipStreamSynchronize");
		return a.dev;
	}
	static constexpr size_t BIG_PIN = 8u << 20;
	void* big_pin[2] = { nullptr, nullptr }; hipEvent_t big_ev[2] = { nullptr, nullptr }; int big_state = 0; // 0 untried, 1 ready, -1 unavailable
	bool big_pin_ok()
	{
		if (big_state == 0) {
			big_state = -1;
			if (!getenv("ABG_NO_PINNED_UPLOAD") &&
			    hipHostMalloc(&big_pin[0], BIG_PIN, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&big_pin[1], BIG_PIN, hipHostMallocDefault) == hipSuccess &&
			    hipEventCreateWithFlags(&big_ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&big_ev[1], hipEventDisableTiming) == hipSuccess)
				big_state = 1;
			else (void)hipGetLastError();
		}
		return big_state == 1;
	}
	void d2h(void* d, const void* s, size_t n)
	{
		if (!n) return;
		if (n <= PIN_BYTES && (pin || hipHostMalloc(&pin, PIN_BYTES, hipHostMallocDefault) == hipSuccess)) {
			check(hipMemcpyAsync(pin, s, n, hipMemcpyDeviceToHost, stream), "hipMemcpy D2H");
			check(hipStreamSynchronize(stream), "hipStreamSynchronize");
			memcpy(d, pin, n);
			prof_drain();
			return;
		}
		check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream), "hipMemcpy D2H");
		check(hipStreamSynchronize(stream), "hipStreamSynchronize");
		prof_drain(); // (the stream is idle: the launches' events have their times)
	}
	void sync() { check(hipStreamSynchronize(stream), "hipStreamSynchronize"); prof_drain(); }
	void d2d(void* d, const void* s, size_t n)
	{
		if (n) check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, stream), "hipMemcpy D2D");
	}
	void* stream_handle() const { return (void*)stream; }
	// out = the entries of `in` (or the indices 0..n-1 when in == NULL) whose flag is set, in
	// order (the same list on every rank of a partitioned run); *count_dev = how many
	void* cub_tmp = nullptr;
	size_t cub_tmp_bytes = 0;
	// the items whose flag byte is 1 (exactly: PASS 1 leaves a 2 on the ops its tiles have settled, abg::PEND_CANDIDATE), in order
	struct IsOne { __host__ __device__ bool operator()(uint8_t f) const { return f == 1; } };
	void compact_flagged(const uint32_t* in, const uint8_t* flags, uint64_t n, uint32_t* out, uint32_t* count_dev)
	{
		begin("compact");
		size_t need = 0;
		hipcub::CountingInputIterator<uint32_t> iota(0);
		hipcub::TransformInputIterator<bool, IsOne, const uint8_t*> fl(flags, IsOne());
		if (in) check(hipcub::DeviceSelect::Flagged(nullptr, need, in, fl, out, count_dev, (int)n, stream), "DeviceSelect");
		else check(hipcub::DeviceSelect::Flagged(nullptr, need, iota, fl, out, count_dev, (int)n, stream), "DeviceSelect");
		if (need > cub_tmp_bytes) {
			if (cub_tmp) { hipStreamSynchronize(stream); hipFree(cub_tmp); }
			cub_tmp_bytes = need * 2;
			check(hipMalloc(&cub_tmp, cub_tmp_bytes), "hipMalloc");
		}
		size_t bytes = cub_tmp_bytes;
		if (in) check(hipcub::DeviceSelect::Flagged(cub_tmp, bytes, in, fl, out, count_dev, (int)n, stream), "DeviceSelect");
		else check(hipcub::DeviceSelect::Flagged(cub_tmp, bytes, iota, fl, out, count_dev, (int)n, stream), "DeviceSelect");
		end("compact");
	}
	// in-place inclusive prefix sum over n 64-bit values
	void inclusive_sum_u64(uint64_t* data, uint64_t n)
	{
		if (!n) return;
		begin("scan");
		size_t need = 0;
		check(hipcub::DeviceScan::InclusiveSum(nullptr, need, data, data, (int)n, stream), "DeviceScan");
		if (need > cub_tmp_bytes) {
			if (cub_tmp) { hipStreamSynchronize(stream); hipFree(cub_tmp); }
			cub_tmp_bytes = need * 2;
			check(hipMalloc(&cub_tmp, cub_tmp_bytes), "hipMalloc");
		}
		size_t bytes = cub_tmp_bytes;
		check(hipcub::DeviceScan::InclusiveSum(cub_tmp, bytes, data, data, (int)n, stream), "DeviceScan");
		end("scan");
	}
	// (keys, values) sorted by key, stable; values 32-bit (AdjList's join: abg_overlap.h)
	void sort_pairs_u64_u32(const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n)
	{
		if (!n) return;
		begin("sort_pairs");
		size_t need = 0;
		check(hipcub::DeviceRadixSort::SortPairs(nullptr, need, kin, kout, vin, vout, (int)n, 0, 64, stream), "DeviceRadixSort");
		if (need > cub_tmp_bytes) {
			if (cub_tmp) { hipStreamSynchronize(stream); hipFree(cub_tmp); }
			cub_tmp_bytes = need * 2;
			check(hipMalloc(&cub_tmp, cub_tmp_bytes), "hipMalloc");
		}
		size_t bytes = cub_tmp_bytes;
		check(hipcub::DeviceRadixSort::SortPairs(cub_tmp, bytes, kin, kout, vin, vout, (int)n, 0, 64, stream), "DeviceRadixSort");
		end("sort_pairs");
	}
	uint32_t max_slots() const { return cus * 8 * 256; }
	uint64_t device_mem_bytes() const
	{
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
		return mem_limit && mem_limit < tot ? (uint64_t)mem_limit : (uint64_t)tot; // (ABG_MEM_LIMIT_MB: a smaller device, for the tests)
	}

	// Per-launch timing (abg_profile_enable): an event before and one behind every launch on the main stream, read when the host
	// waits for the stream anyway (d2h, sync) -- waiting for each launch's second event on the spot kept the host from queueing
	// ahead, ~1.5 % of a configs[1] step.
	struct ProfPending { hipEvent_t a, b; ProfEntry* p; };
	std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
	std::vector<ProfPending> prof_pending;
	std::pair<hipEvent_t, hipEvent_t> prof_cur{ nullptr, nullptr };
	void begin(const char*)
	{
		if (!profiling) return;
		if (prof_free.empty()) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); prof_free.push_back({ a, b }); }
		prof_cur = prof_free.back(); prof_free.pop_back();
		hipEventRecord(prof_cur.first, stream);
	}
	void end(const char* name)
	{
		check(hipGetLastError(), name);
		if (!profiling || !prof_cur.first) return;
		hipEventRecord(prof_cur.second, stream);
		prof_pending.push_back({ prof_cur.first, prof_cur.second, &prof[name] });
		prof_cur = { nullptr, nullptr };
		if (prof_pending.size() >= 8192) prof_drain();
	}
	void prof_drain(bool discard = false)
	{
		for (ProfPending& q : prof_pending) {
			if (!discard && hipEventSynchronize(q.b) == hipSuccess) {
				float ms = 0;
				if (hipEventElapsedTime(&ms, q.a, q.b) == hipSuccess) { q.p->ms += ms; q.p->launches++; }
			}
			prof_free.push_back({ q.a, q.b });
		}
		prof_pending.clear();
	}
	// A kernel launch on the main stream, timed when profiling is on: the two events ride on the dispatch itself
	// (hipExtLaunchKernelGGL) -- no event packets of their own on the stream, and what they bracket is the kernel alone.
	template <class K, class... A>
	void launch_kernel(const char* name, K kernel, dim3 grid, dim3 block, A... args)
	{
		if (!profiling) {
			hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
			check(hipGetLastError(), name);
			return;
		}
		if (prof_free.empty()) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); prof_free.push_back({ a, b }); }
		const std::pair<hipEvent_t, hipEvent_t> ev = prof_free.back(); prof_free.pop_back();
		hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev.first, ev.second, 0, args...);
		check(hipGetLastError(), name);
		prof_pending.push_back({ ev.first, ev.second, &prof[name] });
		if (prof_pending.size() >= 8192) prof_drain();
	}
	template <class F>
	void launch(uint64_t n, F f, const char* name)
	{
		if (!n) return;
		uint64_t blocks = (n + 255) / 256;
		uint64_t cap = (uint64_t)cus * 8;
		if (blocks > cap) blocks = cap;
		launch_kernel(name, k_foreach<F>, dim3((uint32_t)blocks), dim3(256), f, n);
	}
	template <class F>
	void launch_slots(uint64_t n, F f, uint32_t slots, const char* name)
	{
		if (!n) return;
		uint64_t blocks = (n + 63) / 64;
		uint64_t cap = slots / 64 ? slots / 64 : 1;
		if (blocks > cap) blocks = cap;
		launch_kernel(name, k_foreach_w<F>, dim3((uint32_t)blocks), dim3(64), f, n);
	}
	// The same on the side stream: it starts after everything queued on the main stream SO FAR
	// and runs next to whatever the main stream queues afterwards (queue it BEFORE the kernel it
	// is meant to overlap).  One at a time; sync_side() waits for it and books its time.
	template <class F>
	void launch_slots_side(uint64_t n, F f, uint32_t slots, const char* name)
	{
		if (!n) return;
		uint64_t blocks = (n + 63) / 64;
		uint64_t cap = slots / 64 ? slots / 64 : 1;
		if (blocks > cap) blocks = cap;
		hipEventRecord(ev2a, stream);              // order after the main stream's queue
		hipStreamWaitEvent(stream2, ev2a, 0);
		hipEventRecord(ev2a, stream2);
		hipLaunchKernelGGL(k_foreach_w<F>, dim3((uint32_t)blocks), dim3(64), 0, stream2, f, n);
		check(hipGetLastError(), name);
		hipEventRecord(ev2b, stream2);
		side_name = name;
		side_pending = true;
	}
	// A stretch of ordinary launches (launch, launch_tiles, memset) goes to the side stream instead:
	// everything between side_scope_begin() and side_scope_end() is queued there, unprofiled per
	// kernel; wait_side_scope() blocks until it is done and books the whole stretch under `name`.
	hipEvent_t evs0 = nullptr, evs1 = nullptr;
	bool scope_pending = false, scope_saved_prof = false;
	std::string scope_name;
	void side_scope_begin(const char* name)
	{
		if (!evs0) { hipEventCreate(&evs0); hipEventCreate(&evs1); }
		scope_name = name;
		hipEventRecord(evs0, stream);          // (after what the main stream holds so far)
		hipStreamWaitEvent(stream2, evs0, 0);
		std::swap(stream, stream2);
		scope_saved_prof = profiling; profiling = false;
		hipEventRecord(evs0, stream);
	}
	void side_scope_end()
	{
		hipEventRecord(evs1, stream);
		std::swap(stream, stream2);
		profiling = scope_saved_prof;
		scope_pending = true;
	}
	void wait_side_scope()
	{
		if (!scope_pending) return;
		check(hipEventSynchronize(evs1), "hipEventSynchronize");
		if (profiling) {
			float ms = 0;
			hipEventElapsedTime(&ms, evs0, evs1);
			ProfEntry& p = prof[scope_name];
			p.ms += ms;
			p.launches++;
		}
		scope_pending = false;
	}
	// whether what launch_slots_side queued has run (never blocks)
	bool side_done() { return !side_pending || hipEventQuery(ev2b) == hipSuccess; }
	void sync_side()
	{
		if (!side_pending) return;
		check(hipEventSynchronize(ev2b), "hipEventSynchronize");
		if (profiling) {
			float ms = 0;
			hipEventElapsedTime(&ms, ev2a, ev2b);
			ProfEntry& p = prof[side_name];
			p.ms += ms;
			p.launches++;
		}
		side_pending = false;
	}
	template <class F>
	void launch_wave(uint64_t n, F f, const char* name)
	{
		if (!n) return;
		uint64_t blocks = (n + 3) / 4;
		uint64_t cap = (uint64_t)cus * 8;
		if (blocks > cap) blocks = cap;
		launch_kernel(name, k_foreach_wave<F>, dim3((uint32_t)blocks), dim3(256), f, n);
	}
	template <class F>
	void launch_walkers(uint64_t n, F f, uint32_t slots, const char* name)
	{
		if (!n) return;
		uint64_t blocks = n < slots ? n : slots;
		if (!ticket) ticket = (unsigned long long*)alloc(8);
		check(hipMemsetAsync(ticket, 0, 8, stream), "hipMemsetAsync");
		launch_kernel(name, k_walkers<F>, dim3((uint32_t)blocks), dim3(64), f, n, ticket);
	}
	template <class F>
	void launch_tiles(uint64_t n, F f, const char* name)
	{
		if (!n) return;
		uint64_t blocks = n;
		const uint64_t cap = (uint64_t)cus * (F::FAST > 32768 ? 2 : 8);
		if (blocks > cap) blocks = cap;
		launch_kernel(name, k_tiles<F>, dim3((uint32_t)blocks), dim3(F::THREADS), f, n);
	}
	void launch_drain(abg::InsertDrainEnv e)
	{
		begin("insert_drain");
		hipLaunchKernelGGL(k_insert_drain, dim3(1), dim3(1024), 0, stream, e);
		end("insert_drain");
	}
	template <int NW>
	void launch_commit(abg::CommitEnv<NW> e, uint32_t c_begin, uint32_t c_end)
	{
		begin("commit");
		hipLaunchKernelGGL(k_commit<NW>, dim3(1), dim3(COMMIT_THREADS), 0, stream, e, c_begin, c_end);
		end("commit");
	}
};

typedef abg::Session<HipBackend> Sess;

// ---- RCCL communicator behind abg_comm.  librccl is opened at run time (the library stays
// loadable where RCCL is not installed; single-GPU use never touches it).  Collectives are
// enqueued on the engine's stream: no host synchronisation between a kernel, the exchange of
// its output and the kernel that consumes it.
struct RcclApi {
	void* h = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclBroadcast) Broadcast = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string why;
	bool load()
	{
		if (h) return true;
		// RCCL logs (its version banner included) go to stdout unless told otherwise -- the stream the
		// host binary writes its FASTA to
		setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);
		for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
			h = dlopen(name, RTLD_NOW | RTLD_LOCAL); // (never global: a second copy of RCCL, e.g. torch's, must not bind to this one)
			if (h) break;
		}
		if (!h) { why = std::string("cannot open librccl: ") + dlerror(); return false; }
#define ABG_SYM(f) f = (decltype(f))dlsym(h, "nccl" #f); if (!f) { why = "librccl lacks nccl" #f; h = nullptr; return false; }
		ABG_SYM(GetUniqueId) ABG_SYM(CommInitRank) ABG_SYM(CommDestroy) ABG_SYM(AllReduce) ABG_SYM(AllGather)
		ABG_SYM(Broadcast) ABG_SYM(GroupStart) ABG_SYM(GroupEnd) ABG_SYM(Send) ABG_SYM(Recv) ABG_SYM(GetErrorString)
#undef ABG_SYM
		return true;
	}
};
// RCCL prints a version banner to stdout when a communicator is created -- the stream the host
// binary writes its FASTA to and bench.py its JSON line.  While RCCL initialises, descriptor 1
// points at stderr; the banner (sitting in stdio's buffer) is flushed there before 1 comes back.
struct StdoutToStderr {
	int saved;
	StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
	~StdoutToStderr() { fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;
struct RcclComm { ncclComm_t comm = nullptr; int rank = 0, world = 1, device = 0; };

int rccl_fail(ncclResult_t r, const char* what)
{
	fprintf(stderr, "abyss_amd: %s: %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
	return -1;
}
int rccl_all_gather_v(void* user, void* buf, const uint64_t* counts, const uint64_t* displs, void* stream)
{
	RcclComm* c = (RcclComm*)user;
	hipStream_t st = (hipStream_t)stream;
	bool uniform = true;
	for (int q = 0; q < c->world; q++) uniform = uniform && counts[q] == counts[0] && displs[q] == displs[0] + (uint64_t)q * counts[0];
	ncclResult_t r;
	if (uniform) {
		// in place: each rank's part already sits at recvbuff + rank * count
		if (!counts[0]) return 0;
		char* base = (char*)buf + displs[0];
		r = g_rccl.AllGather(base + (uint64_t)c->rank * counts[0], base, counts[0], ncclUint8, c->comm, st);
		return r == ncclSuccess ? 0 : rccl_fail(r, "ncclAllGather");
	}
	// ragged parts: one broadcast per rank, fused into a single group
	r = g_rccl.GroupStart();
	if (r != ncclSuccess) return rccl_fail(r, "ncclGroupStart");
	for (int q = 0; q < c->world; q++) {
		if (!counts[q]) continue;
		char* part = (char*)buf + displs[q];
		r = g_rccl.Broadcast(part, part, counts[q], ncclUint8, q, c->comm, st);
		if (r != ncclSuccess) { g_rccl.GroupEnd(); return rccl_fail(r, "ncclBroadcast"); }
	}
	r = g_rccl.GroupEnd();
	return r == ncclSuccess ? 0 : rccl_fail(r, "ncclGroupEnd");
}
int rccl_all_reduce(void* user, void* buf, uint64_t count, int32_t dtype, int32_t op, void* stream)
{
	RcclComm* c = (RcclComm*)user;
	static const ncclDataType_t dt[3] = { ncclUint8, ncclUint32, ncclUint64 };
	static const ncclRedOp_t ops[3] = { ncclSum, ncclMax, ncclMin };
	if (dtype < 0 || dtype > 2 || op < 0 || op > 2) return -1;
	ncclResult_t r = g_rccl.AllReduce(buf, buf, count, dt[dtype], ops[op], c->comm, (hipStream_t)stream);
	return r == ncclSuccess ? 0 : rccl_fail(r, "ncclAllReduce");
}

// the routed exchanges: one ncclSend / ncclRecv pair per peer in ONE group -- over xGMI that is seven point-to-point
// transfers in flight at once, one per link, instead of a ring; the part a rank keeps for itself is a device copy
int rccl_all_to_all_v(void* user, const void* send, const uint64_t* sc, const uint64_t* sd, void* recv, const uint64_t* rc,
    const uint64_t* rd, void* stream)
{
	RcclComm* c = (RcclComm*)user;
	hipStream_t st = (hipStream_t)stream;
	if (sc[c->rank] != rc[c->rank]) return -1;
	if (sc[c->rank] && hipMemcpyAsync((char*)recv + rd[c->rank], (const char*)send + sd[c->rank], sc[c->rank], hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
	ncclResult_t r = g_rccl.GroupStart();
	if (r != ncclSuccess) return rccl_fail(r, "ncclGroupStart");
	for (int q = 0; q < c->world; q++) {
		if (q == c->rank) continue;
		if (sc[q]) { r = g_rccl.Send((const char*)send + sd[q], sc[q], ncclUint8, q, c->comm, st); if (r != ncclSuccess) { g_rccl.GroupEnd(); return rccl_fail(r, "ncclSend"); } }
		if (rc[q]) { r = g_rccl.Recv((char*)recv + rd[q], rc[q], ncclUint8, q, c->comm, st); if (r != ncclSuccess) { g_rccl.GroupEnd(); return rccl_fail(r, "ncclRecv"); } }
	}
	r = g_rccl.GroupEnd();
	return r == ncclSuccess ? 0 : rccl_fail(r, "ncclGroupEnd");
}

std::mutex g_err_mutex;
std::string g_create_error;

} // namespace

struct abg_ctx {
	Sess s;
	explicit abg_ctx(int dev) : s(dev) {}
};

namespace {
// The C ABI's promise -- a return code and abg_last_error(), never an exit: whatever the engine or the
// backend throws on its way (abg::Failure, std::bad_alloc) ends here.
template <class F>
int guarded(abg_ctx* ctx, F&& body, bool drain = true)
{
	try {
		// (the device's share of the last abg_load_seqs* call may still be running: Session::load_seqs_v)
		if (ctx && drain) ctx->s.drain();
		return body();
	} catch (const abg::Failure& f) {
		if (ctx) ctx->s.error = f.msg;
		return f.code;
	} catch (const std::bad_alloc&) {
		if (ctx) ctx->s.error = "host memory exhausted";
		return ABG_ENOMEM;
	} catch (const std::exception& e) {
		if (ctx) ctx->s.error = e.what();
		return ABG_EINTERNAL;
	}
}
} // namespace

extern "C" {

void abg_params_init(abg_params* p)
{
	memset(p, 0, sizeof *p);
	p->num_hashes = 4;
	p->min_cov = 2;
	p->trim = 0xFFFFFFFFu;
}

int abg_create(const abg_params* p, abg_ctx** out)
{
	if (!p || !out) return ABG_EINVAL;
	*out = nullptr;
	abg_ctx* c = nullptr;
	int rc = guarded(nullptr, [&]() { c = new abg_ctx(p->device); return ABG_OK; });
	if (rc != ABG_OK || !c) { std::lock_guard<std::mutex> g(g_err_mutex); g_create_error = "context construction failed"; return rc != ABG_OK ? rc : ABG_EINTERNAL; }
	rc = guarded(c, [&]() { return c->s.create(*p); });
	if (rc != ABG_OK) {
		std::lock_guard<std::mutex> g(g_err_mutex);
		g_create_error = c->s.error;
		delete c;
		return rc;
	}
	*out = c;
	return ABG_OK;
}
void abg_destroy(abg_ctx* ctx) { delete ctx; }
const char* abg_last_error(const abg_ctx* ctx)
{
	if (ctx) return ctx->s.error.c_str();
	return g_create_error.c_str();
}
int abg_contains_seq(abg_ctx* ctx, const char* seq, uint64_t len, uint32_t* pos_out, uint8_t* contains_out,
    uint64_t cap, uint64_t* n_out)
{
	if (!ctx || !seq || !n_out) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		return ctx->s.contains_seq(seq, len, pos_out, contains_out, cap, n_out);
	});
}

int abg_reset(abg_ctx* ctx)
{
	if (!ctx) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.eng->reset();
		ctx->s.be.sync();
		return ABG_OK;
	});
}

int abg_filter_size(const abg_ctx* ctx, uint64_t* counters)
{
	if (!ctx || !counters) return ABG_EINVAL;
	return guarded((abg_ctx*)ctx, [&]() -> int {
		*counters = ctx->s.eng->size();
		return ABG_OK;
	});
}
int abg_load_seqs(abg_ctx* ctx, const char* seqs, const uint64_t* offsets, uint64_t n)
{
	if (!ctx || (n && (!seqs || !offsets))) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		(void)hipSetDevice(ctx->s.be.device); // (the caller may be a thread that has not touched the device yet)
		return ctx->s.load_seqs(seqs, offsets, n);
	}, false);
}
int abg_load_seqs_v(abg_ctx* ctx, uint32_t nchunks, const char* const* seqs, const uint64_t* const* offsets, const uint64_t* n)
{
	if (!ctx || (nchunks && (!seqs || !offsets || !n))) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		for (uint32_t c = 0; c < nchunks; c++) if (n[c] && (!seqs[c] || !offsets[c])) return ABG_EINVAL;
		(void)hipSetDevice(ctx->s.be.device); // (the caller may be a thread that has not touched the device yet)
		return ctx->s.load_seqs_v(nchunks, seqs, offsets, n);
	}, false);
}
int abg_keep_reads(abg_ctx* ctx, int on, uint64_t expected_bases)
{
	if (!ctx) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		(void)hipSetDevice(ctx->s.be.device);
		return ctx->s.keep_reads(on, expected_bases);
	});
}
int abg_assemble_kept(abg_ctx* ctx, uint8_t* results, abg_contig_cb cb, void* user)
{
	if (!ctx) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		(void)hipSetDevice(ctx->s.be.device);
		return ctx->s.assemble_kept(results, cb, user);
	});
}
int abg_load_packed(abg_ctx* ctx, const uint32_t* d_words, const uint64_t* d_woff, const uint32_t* d_len, uint64_t n)
{
	if (!ctx || (n && (!d_words || !d_woff || !d_len))) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		return ctx->s.load_packed(d_words, d_woff, d_len, n);
	});
}
int abg_counting_stats(abg_ctx* ctx, uint64_t* popcount, uint64_t* filtered_popcount)
{
	if (!ctx) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		uint64_t a = 0, b = 0;
		ctx->s.eng->popcounts(&a, &b);
		if (popcount) *popcount = a;
		if (filtered_popcount) *filtered_popcount = b;
		return ABG_OK;
	});
}
int abg_counters_export(abg_ctx* ctx, uint8_t* host_out)
{
	if (!ctx || !host_out || ctx->s.eng->cascade_mode()) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.eng->counters_to_host(host_out);
		return ABG_OK;
	});
}
int abg_counters_import(abg_ctx* ctx, const uint8_t* host_in)
{
	if (!ctx || !host_in) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.eng->counters_from_host(host_in);
		return ABG_OK;
	});
}
int abg_visited_export(abg_ctx* ctx, uint8_t* host_out)
{
	if (!ctx || !host_out) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.be.d2h(host_out, ctx->s.eng->visited_dev_ro(), ctx->s.eng->visited_bytes());
		return ABG_OK;
	});
}
int abg_visited_import(abg_ctx* ctx, const uint8_t* host_in)
{
	if (!ctx || !host_in) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.be.h2d(ctx->s.eng->visited_dev(), host_in, ctx->s.eng->visited_bytes());
		return ABG_OK;
	});
}
int abg_assemble_seqs(abg_ctx* ctx, const char* seqs, const uint64_t* offsets, uint64_t n,
    uint8_t* results, abg_contig_cb cb, void* user)
{
	if (!ctx || (n && (!seqs || !offsets))) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		return ctx->s.assemble_seqs(seqs, offsets, n, results, cb, user);
	});
}
int abg_assemble_seqs_v(abg_ctx* ctx, uint32_t nchunks, const char* const* seqs, const uint64_t* const* offsets,
    const uint64_t* n, uint8_t* results, abg_contig_cb cb, void* user)
{
	if (!ctx || (nchunks && (!seqs || !offsets || !n))) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		for (uint32_t c = 0; c < nchunks; c++) if (n[c] && (!seqs[c] || !offsets[c])) return ABG_EINVAL;
		(void)hipSetDevice(ctx->s.be.device); // (the caller may be a thread that has not touched the device yet)
		return ctx->s.assemble_seqs_v(nchunks, seqs, offsets, n, results, cb, user);
	});
}
int abg_assemble_packed(abg_ctx* ctx, const uint32_t* d_words, const uint64_t* d_woff,
    const uint32_t* d_len, uint64_t n, uint8_t* results, abg_contig_cb cb, void* user)
{
	if (!ctx || (n && (!d_words || !d_woff || !d_len))) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		return ctx->s.assemble_packed(d_words, d_woff, d_len, n, results, cb, user);
	});
}
int abg_cascade_export(abg_ctx* ctx, uint32_t level, uint8_t* host_out)
{
	if (!ctx || !host_out || !ctx->s.eng->cascade_mode() || level >= ctx->s.eng->cascade_levels()) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.be.d2h(host_out, ctx->s.eng->cascade_level_dev(level), ctx->s.eng->size() / 8);
		return ABG_OK;
	});
}
int abg_get_counters(const abg_ctx* ctx, abg_counters* out)
{
	if (!ctx || !out) return ABG_EINVAL;
	return guarded((abg_ctx*)ctx, [&]() -> int {
		abg::Counters c = ctx->s.eng->counters();
		out->solid_reads = c.solid_reads;
		out->visited_reads = c.visited_reads;
		out->reads_processed = c.reads_processed;
		out->bases_assembled = c.bases_assembled;
		out->next_contig_id = c.contig_id;
		return ABG_OK;
	});
}
int abg_set_counters(abg_ctx* ctx, const abg_counters* in)
{
	if (!ctx || !in) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		abg::Counters c;
		c.solid_reads = in->solid_reads;
		c.visited_reads = in->visited_reads;
		c.reads_processed = in->reads_processed;
		c.bases_assembled = in->bases_assembled;
		c.contig_id = in->next_contig_id;
		ctx->s.eng->set_counters(c);
		return ABG_OK;
	});
}
int abg_hash_seq(abg_ctx* ctx, const char* seq, uint64_t len, uint32_t* pos_out, uint64_t* hashes_out,
    uint64_t cap, uint64_t* n_out)
{
	if (!ctx || !seq || !n_out) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		return ctx->s.hash_seq(seq, len, pos_out, hashes_out, cap, n_out);
	});
}
int abg_attach_comm(abg_ctx* ctx, const abg_comm* comm)
{
	if (!ctx || !comm) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		return ctx->s.attach_comm(*comm);
	});
}
int abg_share_reads(abg_ctx* ctx, const uint32_t* d_words, const uint64_t* d_woff, const uint32_t* d_len,
    uint64_t n_local, const uint32_t** g_words, const uint64_t** g_woff, const uint32_t** g_len, uint64_t* n_total)
{
	if (!ctx || !g_words || !g_woff || !g_len || !n_total || (n_local && (!d_words || !d_woff || !d_len))) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		return ctx->s.share_reads(d_words, d_woff, d_len, n_local, g_words, g_woff, g_len, n_total);
	});
}
int abg_rccl_unique_id(uint8_t id[128])
{
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
	if (!id) return ABG_EINVAL;
	std::lock_guard<std::mutex> g(g_rccl_mutex);
	if (!g_rccl.load()) { g_create_error = g_rccl.why; return ABG_ENODEV; }
	ncclUniqueId u;
	StdoutToStderr quiet;
	ncclResult_t r = g_rccl.GetUniqueId(&u);
	if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return ABG_EINTERNAL; }
	memcpy(id, &u, 128);
	return ABG_OK;
}
int abg_rccl_comm_create(const uint8_t id[128], int32_t rank, int32_t world, int32_t device, abg_comm* out)
{
	if (!id || !out || world < 1 || rank < 0 || rank >= world) return ABG_EINVAL;
	std::lock_guard<std::mutex> g(g_rccl_mutex);
	if (!g_rccl.load()) { g_create_error = g_rccl.why; return ABG_ENODEV; }
	if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return ABG_ENODEV; }
	ncclUniqueId u;
	memcpy(&u, id, 128);
	RcclComm* c = new RcclComm;
	c->rank = rank; c->world = world; c->device = device;
	ncclResult_t r;
	{
		StdoutToStderr quiet;
		r = g_rccl.CommInitRank(&c->comm, world, u, rank);
	}
	if (r != ncclSuccess) { g_create_error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete c; return ABG_EINTERNAL; }
	memset(out, 0, sizeof *out);
	out->rank = rank; out->world = world; out->stream_ordered = 1; out->user = c; out->struct_size = (int32_t)sizeof *out;
	out->all_gather_v = rccl_all_gather_v;
	out->all_reduce = rccl_all_reduce;
	out->all_to_all_v = rccl_all_to_all_v;
	return ABG_OK;
}
int abg_rccl_comm_destroy(abg_comm* comm)
{
	if (!comm || !comm->user || comm->all_reduce != rccl_all_reduce) return ABG_EINVAL;
	RcclComm* c = (RcclComm*)comm->user;
	if (c->comm) g_rccl.CommDestroy(c->comm);
	delete c;
	comm->user = nullptr;
	return ABG_OK;
}
int abg_dev_alloc(abg_ctx* ctx, uint64_t bytes, void** out)
{
	if (!ctx || !out) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		*out = ctx->s.be.try_alloc(bytes);
		if (!*out) { ctx->s.error = "device allocation failed"; return ABG_ENOMEM; }
		return ABG_OK;
	});
}
int abg_dev_free(abg_ctx* ctx, void* ptr)
{
	if (!ctx) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		if (ptr) { ctx->s.be.sync(); hipFree(ptr); }
		return ABG_OK;
	});
}
int abg_dev_copy(abg_ctx* ctx, void* dst, const void* src, uint64_t n, int32_t kind)
{
	if (!ctx || (n && (!dst || !src)) || kind < 0 || kind > 2) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		if (kind == 0) ctx->s.be.h2d(dst, src, n);
		else if (kind == 1) ctx->s.be.d2h(dst, src, n);
		else { ctx->s.be.d2d(dst, src, n); ctx->s.be.sync(); }
		return ABG_OK;
	});
}
int abg_output_graph_seqs(abg_ctx* ctx, const char* seqs, const uint64_t* offsets, uint64_t n,
    abg_text_cb cb, void* user, uint64_t* nodes, uint64_t* edges)
{
	if (!ctx || (n && (!seqs || !offsets))) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		return ctx->s.output_graph_seqs(seqs, offsets, n, cb, user, nodes, edges);
	});
}
int abg_profile_enable(abg_ctx* ctx, int on)
{
	if (!ctx) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.be.prof_drain();
		ctx->s.be.profiling = on != 0;
		return ABG_OK;
	});
}
int abg_profile_reset(abg_ctx* ctx)
{
	if (!ctx) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.be.prof_drain(true);
		ctx->s.be.prof.clear();
		return ABG_OK;
	});
}
int abg_profile_get(abg_ctx* ctx, const char* name, double* total_ms, uint64_t* launches)
{
	if (!ctx || !name) return ABG_EINVAL;
	return guarded(ctx, [&]() -> int {
		ctx->s.be.prof_drain();
		auto it = ctx->s.be.prof.find(name);
		double ms = 0;
		uint64_t n = 0;
		if (it != ctx->s.be.prof.end()) { ms = it->second.ms; n = it->second.launches; }
		if (total_ms) *total_ms = ms;
		if (launches) *launches = n;
		return ABG_OK;
	});
}
int abg_get_stats(const abg_ctx* ctx, abg_stats* out)
{
	if (!ctx || !out) return ABG_EINVAL;
	return guarded((abg_ctx*)ctx, [&]() -> int {
		auto s = ctx->s.eng->stats();
		out->insert_rounds = s.insert_rounds;
		out->walk_rounds = s.rounds;
		out->candidates = s.candidates;
		out->walked = s.walked;
		out->rewalked = s.rewalked;
		out->commit_breaks = s.breaks; out->commit_rounds = s.commit_rounds; out->generated = s.generated;
		out->bulk_calls = s.bulk_calls; out->bulk_steps = s.bulk_steps; out->lin_steps = s.lin_steps; out->guide_slots = s.guide_slots; out->chain_steps = s.chain_steps; out->batch_cuts = s.batch_cuts; out->overflows = s.overflows;
		out->memo_hits = s.memo_hits; out->memo_adds = s.memo_adds;
		out->tiled_ops = s.tiled_ops; out->tiled_pending = s.tiled_pending; out->tile_overflows = s.tile_overflows;
		out->cls_covered_reads = s.cls_covered_reads; out->archive_bases = s.archive_bases; out->cls_decided_reads = s.cls_decided_reads;
		out->counter_bytes_held = ctx->s.eng->counter_bytes_held();
		return ABG_OK;
	});
}

// ---- the AdjList stage: k-1 overlaps of contig ends (abg_overlap.h) ----
} // extern "C"
struct abg_overlap {
	HipBackend be;
	abg::OverlapJoin<HipBackend> join;
	std::string error;
	explicit abg_overlap(int dev) : be(dev), join(be) {}
};
namespace {
std::string g_overlap_create_error;
template <class F>
int guarded_ov(abg_overlap* o, F&& body)
{
	try {
		return body();
	} catch (const abg::Failure& f) {
		o->error = f.msg;
		return f.code;
	} catch (const std::bad_alloc&) {
		o->error = "host memory exhausted";
		return ABG_ENOMEM;
	} catch (const std::exception& e) {
		o->error = e.what();
		return ABG_EINTERNAL;
	}
}
} // namespace
extern "C" {
int abg_overlap_create(int device, abg_overlap** out)
{
	if (!out) return ABG_EINVAL;
	*out = nullptr;
	abg_overlap* o = nullptr;
	try {
		o = new abg_overlap(device);
	} catch (const std::exception& e) {
		g_overlap_create_error = e.what();
		return ABG_EINTERNAL;
	}
	if (!o->be.ok()) {
		g_overlap_create_error = o->be.why();
		delete o;
		return ABG_ENODEV;
	}
	*out = o;
	return ABG_OK;
}
void abg_overlap_destroy(abg_overlap* o) { delete o; }
const char* abg_overlap_last_error(const abg_overlap* o) { return o ? o->error.c_str() : g_overlap_create_error.c_str(); }
int abg_overlap_join(abg_overlap* o, uint32_t overlap, uint64_t n_contigs, const uint64_t* head_keys, const uint64_t* tail_keys,
    int strand_specific, uint64_t* n_edges)
{
	if (!o || !n_edges || (n_contigs && (!head_keys || !tail_keys))) return ABG_EINVAL;
	if (overlap < 1 || overlap > 32 * abg::OV_MAX_WORDS) { o->error = "overlap length out of range (1..256)"; return ABG_EINVAL; }
	if (n_contigs >= (1ull << 30)) { o->error = "too many contigs (vertex ids are 32-bit)"; return ABG_EINVAL; }
	return guarded_ov(o, [&]() -> int {
		o->join.run(overlap, n_contigs, head_keys, tail_keys, strand_specific != 0);
		*n_edges = o->join.edges();
		return ABG_OK;
	});
}
int abg_overlap_edges(abg_overlap* o, uint64_t* offsets, uint32_t* targets)
{
	if (!o) return ABG_EINVAL;
	return guarded_ov(o, [&]() -> int {
		o->join.fetch(offsets, targets);
		return ABG_OK;
	});
}
int abg_overlap_profile(abg_overlap* o, int on)
{
	if (!o) return ABG_EINVAL;
	o->be.prof_drain();
	o->be.profiling = on != 0;
	return ABG_OK;
}
int abg_overlap_profile_get(abg_overlap* o, const char* name, double* total_ms, uint64_t* launches)
{
	if (!o || !name) return ABG_EINVAL;
	o->be.prof_drain();
	auto it = o->be.prof.find(name);
	if (total_ms) *total_ms = it != o->be.prof.end() ? it->second.ms : 0;
	if (launches) *launches = it != o->be.prof.end() ? it->second.launches : 0;
	return ABG_OK;
}

} // extern "C"
