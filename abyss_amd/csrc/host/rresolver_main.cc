// abyss-rresolver-short -- drop-in for the reference's RResolver (RResolver/RResolverShort.cpp), the rule abyss-pe runs after
// AdjList in Bloom mode (bin/abyss-pe:581-585): the Bloom filter of the reads' r-mers is built and asked on the GPU through
// abg_rr_* (include/abyss_amd.h), everything else is rresolver_core.h.  No CPU fallback: without a HIP device the program fails.
#include "rresolver_core.h"

#include "abyss_amd.h"

#include <future>
#include <unistd.h>

namespace {

// btllib::KmerBloomFilter as RResolver uses it, over abg_rr_*
struct GpuFilter : abgrr::ReadFilter {
	int device;
	abg_rr* f = nullptr;
	bool timing;
	// The first HIP call of a process pays for the runtime's start (~0.1 s on the GPU boxes): a filter of a few bytes made and dropped
	// on a thread of its own while the graph and the contigs are read; create() waits for it.  (What it says does not matter: without
	// a device the real create() fails and reports.)
	// (The reader's error paths leave through exit() like the reference's: exit() then waits for that thread -- wait_for_warm, as
	// AdjList does -- so that the static destructors never run beside a HIP runtime that is still starting.)
	static std::future<void>& warm() { static std::future<void> w; return w; }
	static void wait_for_warm() { if (warm().valid()) warm().wait(); }
	explicit GpuFilter(int device) : device(device), timing(getenv("ABG_RR_TIMING") != nullptr)
	{
		if (getenv("ABG_RR_NO_WARM")) return;
		(void)warm(); // (made before the handler is registered: destroyed after it has run)
		atexit(wait_for_warm);
		warm() = std::async(std::launch::async, [device]() { abg_rr* t = nullptr; if (abg_rr_create(device, 64, 7, 32, &t) == ABG_OK) abg_rr_destroy(t); });
	}
	~GpuFilter() override { wait_for_warm(); abg_rr_destroy(f); }
	[[noreturn]] void fail(const char* what)
	{
		fprintf(stderr, ABG_RR_PROGRAM ": %s: %s\n", what, abg_rr_last_error(f));
		exit(EXIT_FAILURE);
	}
	void create(uint64_t bytes, unsigned hash_num, unsigned r) override
	{
		wait_for_warm();
		report();
		abg_rr_destroy(f);
		f = nullptr;
		if (abg_rr_create(device, bytes, hash_num, r, &f) != ABG_OK) fail("creating the read Bloom filter");
		if (timing) abg_rr_profile(f, 1);
	}
	void insert(const char* seqs, const uint64_t* offsets, uint64_t n, uint32_t max_bases, const uint32_t* lengths, uint32_t nlen) override
	{
		if (abg_rr_insert_seqs(f, seqs, offsets, n, max_bases, lengths, nlen, nullptr) != ABG_OK) fail("loading reads into the Bloom filter");
	}
	void contains(const char* seqs, const uint64_t* offsets, uint64_t n, uint32_t* found) override
	{
		if (abg_rr_contains_seqs(f, seqs, offsets, n, found) != ABG_OK) fail("querying the Bloom filter");
	}
	uint64_t popcount() override
	{
		uint64_t c = 0;
		if (abg_rr_popcount(f, &c) != ABG_OK) fail("counting the Bloom filter's bits");
		return c;
	}
	uint64_t bytes() override
	{
		uint64_t b = 0;
		abg_rr_bytes(f, &b);
		return b;
	}
	void report()
	{
		if (!f || !timing) return;
		for (const char* name : { "rr_insert", "rr_contains", "rr_popcount" }) {
			double ms = 0;
			uint64_t launches = 0;
			abg_rr_profile_get(f, name, &ms, &launches);
			fprintf(stderr, "[timing] %-12s %9.3f ms  %llu launches\n", name, ms, (unsigned long long)launches);
		}
	}
};

} // namespace

int main(int argc, char** argv)
{
	abgrr::Options opt;
	int status = 0;
	if (!abgrr::parse_options(argc, argv, opt, &status)) return status;
	GpuFilter filter(opt.device);
	abgrr::Resolver resolver(opt, filter);
	status = resolver.run();
	filter.report();
	fflush(NULL);
	if (!getenv("ABG_ORDERLY_EXIT")) _exit(status); // (as abyss-bloom-dbg: the output is written; the kernel reclaims the device faster than we can)
	return status;
}
