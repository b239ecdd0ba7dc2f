// abg_rr.hip -- gfx950 kernels and C ABI of the RResolver read filter (include/abyss_amd.h, abg_rr_*; logic in abg_rr.h).
//
//   k_rr_insert    one read prefix per lane.  The host hands over fixed-stride ASCII records (a read's first
//                  r + extract - 1 characters, padded with 'N'); a workgroup copies its records from HBM into LDS with
//                  coalesced 16-byte loads (a lane walking its own 128-byte record in global memory would touch 64
//                  cache lines per wave-load), the rows padded to an odd number of dwords so that 64 lanes reading
//                  "their" dword j hit 64 different banks; every lane then rolls ntHash over its row and sets
//                  hash_num bits per r-mer with fire-and-forget atomic ORs.
//   k_rr_contains  one candidate sequence per wavefront: lane L hashes the r-mers L, L + 64, ... from scratch (a
//                  sequence is shorter than 2r, so at most r of them), probes the filter, and the wave's count is a
//                  popcount of the ballot.
//   k_rr_popcount  bits set (occupancy and FPR of the -v report).
// Bound: the insert kernel is bound by the random 4-byte atomics (hash_num per r-mer; a filter beyond the 256 MB
// Infinity Cache moves a 64-byte sector each); its algorithmic bytes per read are stride + extract * hash_num / 8.
//
// Built with the rest of the library: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/abyss_amd.h"
#include "abg_rr.h"

namespace {

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_rr_insert(abg::RRParams p, const uint4* __restrict__ recs, uint32_t stride, uint32_t span,
    uint64_t n, uint32_t* __restrict__ bits)
{
	extern __shared__ uint32_t lds[];
	const uint32_t q = stride / 16;      // uint4 per record
	const uint32_t lw = stride / 4 + 1;  // dwords per LDS row: odd, so lane t's dword j sits in bank (t * lw + j) % 64, all different
	const uint64_t groups = (n + THREADS - 1) / THREADS;
	for (uint64_t g = blockIdx.x; g < groups; g += gridDim.x) {
		const uint64_t base = g * THREADS;
		const uint32_t cnt = (uint32_t)(n - base < (uint64_t)THREADS ? n - base : THREADS);
		const uint4* src = recs + base * q;
		const uint32_t total = cnt * q;
		for (uint32_t i = threadIdx.x; i < total; i += THREADS) {
			const uint4 v = src[i];
			const uint32_t rec = i / q, w = (i - rec * q) * 4;
			uint32_t* d = lds + rec * lw + w;
			d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
		}
		__syncthreads();
		if (threadIdx.x < cnt) {
			const uint32_t* row = lds + threadIdx.x * lw;
			abg::rr_insert_record(p, [&](uint32_t i) { return (row[i >> 2] >> (8 * (i & 3))) & 0xFFu; }, span, bits);
		}
		__syncthreads();
	}
}

// records too long for an LDS row: each lane reads its own record where it lies
__global__ void __launch_bounds__(256) k_rr_insert_direct(abg::RRParams p, const unsigned char* __restrict__ recs, uint32_t stride,
    uint32_t span, uint64_t n, uint32_t* __restrict__ bits)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n; i += step) {
		const unsigned char* s = recs + i * stride;
		abg::rr_insert_record(p, [&](uint32_t j) { return (unsigned)s[j]; }, span, bits);
	}
}

__global__ void __launch_bounds__(256) k_rr_contains(abg::RRParams p, const unsigned char* __restrict__ seqs, const uint64_t* __restrict__ off,
    uint64_t n, const uint32_t* __restrict__ bits, uint32_t* __restrict__ found)
{
	const uint32_t lane = threadIdx.x & 63;
	uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	for (; w < n; w += nw) {
		const uint64_t a = off[w], len = off[w + 1] - a;
		const unsigned char* s = seqs + a;
		uint32_t cnt = 0;
		if (len >= p.r) {
			const uint64_t m = len - p.r + 1;
			for (uint64_t at = lane; at < ((m + 63) & ~63ull); at += 64) {
				const bool hit = at < m && abg::rr_contains_at(p, [&](uint32_t i) { return (unsigned)s[i]; }, (uint32_t)at, bits) == 1;
				cnt += (uint32_t)__popcll(__ballot(hit));
			}
		}
		if (lane == 0) found[w] = cnt;
	}
}

__global__ void __launch_bounds__(256) k_rr_popcount(const uint4* __restrict__ words, uint64_t n16, unsigned long long* out)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	unsigned long long c = 0;
	for (; i < n16; i += step) {
		const uint4 v = words[i];
		c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
	}
	for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
	if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

struct Prof { double ms = 0; uint64_t launches = 0; };

} // namespace

struct abg_rr {
	int device = 0;
	hipStream_t stream = nullptr;
	abg::RRParams p;
	uint64_t bytes = 0;
	uint32_t* bits = nullptr; // the filter: `bytes` bytes (+ padding to 16)
	// two staging slots: pinned host memory the reads' prefixes are packed into, device memory they are copied to, and the
	// event that says the slot's kernel is done with both
	static constexpr size_t SLOT = 32u << 20;
	char* pin[2] = { nullptr, nullptr };
	char* dev[2] = { nullptr, nullptr };
	hipEvent_t done[2] = { nullptr, nullptr };
	bool busy[2] = { false, false };
	int next = 0;
	void* qdev = nullptr; size_t qcap = 0; // queries: sequences, offsets, counts
	unsigned long long* d_count = nullptr;
	uint32_t cus = 256;
	bool profiling = false;
	std::map<std::string, Prof> prof;
	std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
	std::string error;
	~abg_rr()
	{
		(void)hipSetDevice(device);
		if (stream) (void)hipStreamSynchronize(stream);
		for (auto& e : pending) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
		for (int i = 0; i < 2; i++) {
			if (pin[i]) (void)hipHostFree(pin[i]);
			if (dev[i]) (void)hipFree(dev[i]);
			if (done[i]) (void)hipEventDestroy(done[i]);
		}
		if (qdev) (void)hipFree(qdev);
		if (d_count) (void)hipFree(d_count);
		if (bits) (void)hipFree(bits);
		if (stream) (void)hipStreamDestroy(stream);
	}
};

namespace {

std::string g_rr_create_error;

bool rr_ok(abg_rr* f, hipError_t e, const char* what)
{
	if (e == hipSuccess) return true;
	(void)hipGetLastError();
	f->error = std::string(what) + " failed: " + hipGetErrorString(e);
	return false;
}
int rr_code_of(hipError_t e) { return e == hipErrorOutOfMemory ? ABG_ENOMEM : ABG_EINTERNAL; }

struct Timed { // brackets one launch with events when profiling
	abg_rr* f; const char* name; hipEvent_t a = nullptr, b = nullptr;
	Timed(abg_rr* f, const char* name) : f(f), name(name)
	{
		if (!f->profiling) return;
		(void)hipEventCreate(&a); (void)hipEventCreate(&b);
		(void)hipEventRecord(a, f->stream);
	}
	~Timed()
	{
		if (!a) return;
		(void)hipEventRecord(b, f->stream);
		f->pending.push_back({ name, { a, b } });
	}
};
void prof_drain(abg_rr* f)
{
	for (auto& e : f->pending) {
		float ms = 0;
		(void)hipEventSynchronize(e.second.second);
		if (hipEventElapsedTime(&ms, e.second.first, e.second.second) == hipSuccess) { f->prof[e.first].ms += ms; f->prof[e.first].launches++; }
		(void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second);
	}
	f->pending.clear();
}

// packs records [a, b) of the caller's reads into a staging slot: the first `span` characters, the rest of the row 'N'
void pack_rows(char* dst, uint32_t stride, uint32_t span, const char* seqs, const uint64_t* offsets, const uint64_t* rows, uint64_t a, uint64_t b)
{
	for (uint64_t i = a; i < b; i++) {
		const uint64_t r = rows[i], o = offsets[r], len = offsets[r + 1] - o;
		const uint32_t take = (uint32_t)std::min<uint64_t>(len, span);
		char* d = dst + (i - a) * stride;
		memcpy(d, seqs + o, take);
		memset(d + take, 'N', stride - take);
	}
}

int launch_insert(abg_rr* f, int slot, uint32_t stride, uint32_t span, uint64_t n)
{
	Timed t(f, "rr_insert");
	const uint32_t lw = stride / 4 + 1;
	const uint32_t grid_cap = f->cus * 8;
	auto grid = [&](int threads) { return (unsigned)std::min<uint64_t>((n + threads - 1) / threads, grid_cap); };
	if ((size_t)256 * lw * 4 <= 48u << 10) // (48 KB a workgroup: three per CU beside each other)
		k_rr_insert<256><<<grid(256), 256, 256 * lw * 4, f->stream>>>(f->p, (const uint4*)f->dev[slot], stride, span, n, f->bits);
	else if ((size_t)128 * lw * 4 <= 48u << 10)
		k_rr_insert<128><<<grid(128), 128, 128 * lw * 4, f->stream>>>(f->p, (const uint4*)f->dev[slot], stride, span, n, f->bits);
	else if ((size_t)64 * lw * 4 <= 64u << 10)
		k_rr_insert<64><<<grid(64), 64, 64 * lw * 4, f->stream>>>(f->p, (const uint4*)f->dev[slot], stride, span, n, f->bits);
	else
		k_rr_insert_direct<<<grid(256), 256, 0, f->stream>>>(f->p, (const unsigned char*)f->dev[slot], stride, span, n, f->bits);
	const hipError_t e = hipGetLastError();
	if (!rr_ok(f, e, "the insert kernel launch")) return rr_code_of(e);
	return ABG_OK;
}

} // namespace

extern "C" {

int abg_rr_create(int device, uint64_t bytes, uint32_t hash_num, uint32_t r, abg_rr** out)
{
	if (!out) return ABG_EINVAL;
	*out = nullptr;
	if (bytes == 0 || hash_num == 0 || hash_num > abg::RR_MAX_HASHES || r == 0) { g_rr_create_error = "bad filter size, hash count or r"; return ABG_EINVAL; }
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); g_rr_create_error = "no HIP device available (abyss_amd has no CPU fallback)"; return ABG_ENODEV; }
	if (device < 0 || device >= n) { g_rr_create_error = "HIP device ordinal out of range"; return ABG_ENODEV; }
	abg_rr* f = new abg_rr;
	f->device = device;
	f->bytes = (bytes + 7) / 8 * 8; // btllib::BloomFilter rounds its size up to whole 64-bit words
	f->p = abg::make_rr_params(r, hash_num, f->bytes);
	hipError_t e = hipSetDevice(device);
	hipDeviceProp_t prop;
	if (e == hipSuccess && hipGetDeviceProperties(&prop, device) == hipSuccess) f->cus = (uint32_t)prop.multiProcessorCount;
	if (e == hipSuccess) e = hipStreamCreate(&f->stream);
	if (e == hipSuccess) e = hipMalloc((void**)&f->bits, (f->bytes + 15) / 16 * 16);
	if (e == hipSuccess) e = hipMemsetAsync(f->bits, 0, (f->bytes + 15) / 16 * 16, f->stream);
	if (e == hipSuccess) e = hipMalloc((void**)&f->d_count, 8);
	for (int i = 0; i < 2 && e == hipSuccess; i++) {
		e = hipHostMalloc((void**)&f->pin[i], abg_rr::SLOT, hipHostMallocDefault);
		if (e == hipSuccess) e = hipMalloc((void**)&f->dev[i], abg_rr::SLOT);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&f->done[i], hipEventDisableTiming);
	}
	if (e != hipSuccess) {
		(void)hipGetLastError();
		g_rr_create_error = std::string("creating the read filter failed: ") + hipGetErrorString(e);
		const int rc = rr_code_of(e);
		delete f;
		return rc;
	}
	*out = f;
	return ABG_OK;
}

void abg_rr_destroy(abg_rr* f) { delete f; }
const char* abg_rr_last_error(const abg_rr* f) { return f ? f->error.c_str() : g_rr_create_error.c_str(); }

int abg_rr_bytes(const abg_rr* f, uint64_t* bytes)
{
	if (!f || !bytes) return ABG_EINVAL;
	*bytes = f->bytes;
	return ABG_OK;
}

int abg_rr_clear(abg_rr* f)
{
	if (!f) return ABG_EINVAL;
	(void)hipSetDevice(f->device);
	const hipError_t e = hipMemsetAsync(f->bits, 0, (f->bytes + 15) / 16 * 16, f->stream);
	return rr_ok(f, e, "clearing the filter") ? ABG_OK : rr_code_of(e);
}

int abg_rr_insert_seqs(abg_rr* f, const char* seqs, const uint64_t* offsets, uint64_t n, uint32_t max_bases, const uint32_t* lengths,
    uint32_t n_lengths, uint64_t* n_inserted)
{
	if (!f || (n && (!seqs || !offsets)) || (n_lengths && !lengths)) return ABG_EINVAL;
	if (max_bases == 0 || max_bases > abg::RR_MAX_SPAN) { f->error = "max_bases out of range (1..4096)"; return ABG_EINVAL; }
	(void)hipSetDevice(f->device);
	// the reads of the wanted lengths that are long enough to hold an r-mer (BloomFilters.cpp:182-193)
	std::vector<uint64_t> rows;
	rows.reserve(n);
	for (uint64_t i = 0; i < n; i++) {
		const uint64_t len = offsets[i + 1] - offsets[i];
		bool want = n_lengths == 0;
		for (uint32_t j = 0; j < n_lengths && !want; j++) want = len == lengths[j];
		if (n_inserted && want) ++*n_inserted;
		if (want && len >= f->p.r) rows.push_back(i);
	}
	const uint32_t span = max_bases, stride = (span + 15) / 16 * 16;
	const uint64_t per_slot = abg_rr::SLOT / stride;
	static const unsigned pack_threads = []() { const char* e = getenv("ABG_RR_PACK_THREADS"); return e ? (unsigned)std::max(1, atoi(e)) : 4u; }();
	for (uint64_t a = 0; a < rows.size(); a += per_slot) {
		const uint64_t b = std::min<uint64_t>(rows.size(), a + per_slot), cnt = b - a;
		const int s = f->next;
		f->next ^= 1;
		if (f->busy[s]) {
			const hipError_t e = hipEventSynchronize(f->done[s]);
			if (!rr_ok(f, e, "waiting for a staging slot")) return rr_code_of(e);
			f->busy[s] = false;
		}
		{
			const unsigned T = (unsigned)std::min<uint64_t>(pack_threads, cnt / 4096 + 1);
			std::vector<std::thread> pool;
			const uint64_t per = (cnt + T - 1) / T;
			for (unsigned t = 1; t < T; t++) {
				const uint64_t x = a + std::min(cnt, t * per), y = a + std::min(cnt, (t + 1) * per);
				if (x < y) pool.emplace_back([=]() { pack_rows(f->pin[s] + (x - a) * stride, stride, span, seqs, offsets, rows.data(), x, y); });
			}
			pack_rows(f->pin[s], stride, span, seqs, offsets, rows.data(), a, a + std::min(cnt, per));
			for (auto& th : pool) th.join();
		}
		hipError_t e = hipMemcpyAsync(f->dev[s], f->pin[s], cnt * stride, hipMemcpyHostToDevice, f->stream);
		if (!rr_ok(f, e, "copying reads to the device")) return rr_code_of(e);
		const int rc = launch_insert(f, s, stride, span, cnt);
		if (rc != ABG_OK) return rc;
		e = hipEventRecord(f->done[s], f->stream);
		if (!rr_ok(f, e, "hipEventRecord")) return rr_code_of(e);
		f->busy[s] = true;
	}
	return ABG_OK;
}

int abg_rr_contains_seqs(abg_rr* f, const char* seqs, const uint64_t* offsets, uint64_t n, uint32_t* found)
{
	if (!f || (n && (!seqs || !offsets || !found))) return ABG_EINVAL;
	if (n == 0) return ABG_OK;
	(void)hipSetDevice(f->device);
	const uint64_t a = offsets[0], total = offsets[n] - a;
	const size_t off_at = (total + 15) / 16 * 16, cnt_at = off_at + (n + 1) * 8, need = cnt_at + n * 4;
	if (need > f->qcap) {
		if (f->qdev) { (void)hipStreamSynchronize(f->stream); (void)hipFree(f->qdev); f->qdev = nullptr; f->qcap = 0; }
		const size_t cap = std::max<size_t>(need, 1u << 20);
		const hipError_t e = hipMalloc(&f->qdev, cap);
		if (!rr_ok(f, e, "device memory for the queries")) return rr_code_of(e);
		f->qcap = cap;
	}
	char* base = (char*)f->qdev;
	std::vector<uint64_t> rel(n + 1);
	for (uint64_t i = 0; i <= n; i++) rel[i] = offsets[i] - a;
	hipError_t e = hipMemcpyAsync(base, seqs + a, total, hipMemcpyHostToDevice, f->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(base + off_at, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, f->stream);
	if (!rr_ok(f, e, "copying the queries to the device")) return rr_code_of(e);
	{
		Timed t(f, "rr_contains");
		const unsigned grid = (unsigned)std::min<uint64_t>((n + 3) / 4, (uint64_t)f->cus * 8);
		k_rr_contains<<<grid, 256, 0, f->stream>>>(f->p, (const unsigned char*)base, (const uint64_t*)(base + off_at), n, f->bits, (uint32_t*)(base + cnt_at));
		e = hipGetLastError();
		if (!rr_ok(f, e, "the query kernel launch")) return rr_code_of(e);
	}
	e = hipMemcpyAsync(found, base + cnt_at, n * 4, hipMemcpyDeviceToHost, f->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(f->stream); // (also keeps `rel` alive until its copy is done)
	if (!rr_ok(f, e, "reading the counts back")) return rr_code_of(e);
	return ABG_OK;
}

int abg_rr_popcount(abg_rr* f, uint64_t* bits_set)
{
	if (!f || !bits_set) return ABG_EINVAL;
	(void)hipSetDevice(f->device);
	hipError_t e = hipMemsetAsync(f->d_count, 0, 8, f->stream);
	if (e == hipSuccess) {
		Timed t(f, "rr_popcount");
		const uint64_t n16 = (f->bytes + 15) / 16;
		k_rr_popcount<<<(unsigned)std::min<uint64_t>((n16 + 255) / 256, (uint64_t)f->cus * 8), 256, 0, f->stream>>>((const uint4*)f->bits, n16, f->d_count);
		e = hipGetLastError();
	}
	unsigned long long c = 0;
	if (e == hipSuccess) e = hipMemcpyAsync(&c, f->d_count, 8, hipMemcpyDeviceToHost, f->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(f->stream);
	if (!rr_ok(f, e, "counting the filter's bits")) return rr_code_of(e);
	*bits_set = c;
	return ABG_OK;
}

int abg_rr_export(abg_rr* f, uint8_t* host_out)
{
	if (!f || !host_out) return ABG_EINVAL;
	(void)hipSetDevice(f->device);
	hipError_t e = hipMemcpyAsync(host_out, f->bits, f->bytes, hipMemcpyDeviceToHost, f->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(f->stream);
	return rr_ok(f, e, "copying the filter to the host") ? ABG_OK : rr_code_of(e);
}

int abg_rr_sync(abg_rr* f)
{
	if (!f) return ABG_EINVAL;
	(void)hipSetDevice(f->device);
	const hipError_t e = hipStreamSynchronize(f->stream);
	f->busy[0] = f->busy[1] = false;
	return rr_ok(f, e, "hipStreamSynchronize") ? ABG_OK : rr_code_of(e);
}

int abg_rr_profile(abg_rr* f, int on)
{
	if (!f) return ABG_EINVAL;
	f->profiling = on != 0;
	return ABG_OK;
}
int abg_rr_profile_get(abg_rr* f, const char* name, double* total_ms, uint64_t* launches)
{
	if (!f || !name) return ABG_EINVAL;
	(void)hipSetDevice(f->device);
	prof_drain(f);
	auto it = f->prof.find(name);
	if (total_ms) *total_ms = it == f->prof.end() ? 0 : it->second.ms;
	if (launches) *launches = it == f->prof.end() ? 0 : it->second.launches;
	return ABG_OK;
}

} // extern "C"
