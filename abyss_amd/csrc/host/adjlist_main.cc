// AdjList -- drop-in for the reference's AdjList (AdjList/AdjList.cpp), the stage abyss-pe runs on
// the unitigs of abyss-bloom-dbg: the overlaps of exactly k-1 bases come from abg_overlap_join on
// the GPU (include/abyss_amd.h), everything else is adjlist_core.h.  No CPU fallback: without a
// HIP device the program fails.
#include "adjlist_core.h"

#include "abyss_amd.h"

#include <future>
#include <unistd.h>

// The device comes up on a thread of its own while the contigs are read, and the reader's error paths (a bad character, a
// duplicate id, ...) leave through exit() like the reference's: exit() then waits here for that thread, so that the static
// destructors never run beside a HIP runtime that is still starting.
static std::future<int> g_device_up;
static void wait_for_device() { if (g_device_up.valid()) g_device_up.wait(); }

int main(int argc, char** argv)
{
	abgadj::Options opt;
	int device = 0, status = 0;
	if (!abgadj::parse_options(argc, argv, opt, &device, &status)) return status;
	// (the device comes up -- HIP start-up, a quarter of a second -- while the contigs are read)
	abg_overlap* ov = nullptr;
	atexit(wait_for_device);
	g_device_up = std::async(std::launch::async, [&]() { return abg_overlap_create(device, &ov); });
	std::future<int>& up = g_device_up;
	const bool timing = getenv("ABG_ADJ_TIMING") != nullptr;
	auto device_ready = [&]() {
		if (!up.valid()) return;
		if (up.get() != ABG_OK) {
			fprintf(stderr, ABG_ADJ_PROGRAM ": %s\n", abg_overlap_last_error(nullptr));
			exit(EXIT_FAILURE);
		}
		if (timing) abg_overlap_profile(ov, 1);
	};
	abgadj::Join join = [&](uint32_t overlap, uint64_t n, const uint64_t* head, const uint64_t* tail, bool ss,
	                        std::vector<uint64_t>& off, std::vector<uint32_t>& tgt) {
		uint64_t ne = 0;
		device_ready();
		int r = abg_overlap_join(ov, overlap, n, head, tail, ss ? 1 : 0, &ne);
		if (r == ABG_OK) {
			off.assign(2 * n + 1, 0);
			tgt.assign(ne, 0);
			r = abg_overlap_edges(ov, off.data(), tgt.data());
		}
		if (r != ABG_OK) {
			fprintf(stderr, ABG_ADJ_PROGRAM ": %s\n", abg_overlap_last_error(ov));
			exit(EXIT_FAILURE);
		}
	};
	status = abgadj::run(opt, join, stdout);
	device_ready(); // (no contigs, no join: still the device's verdict)
	if (timing) {
		for (const char* name : { "overlap_keys", "sort_pairs", "overlap_count", "scan", "overlap_fill" }) {
			double ms = 0;
			uint64_t launches = 0;
			abg_overlap_profile_get(ov, name, &ms, &launches);
			fprintf(stderr, "[timing] %-14s %8.3f ms  %llu launches\n", name, ms, (unsigned long long)launches);
		}
	}
	fflush(NULL);
	if (!getenv("ABG_ORDERLY_EXIT")) _exit(status); // (as abyss-bloom-dbg: the output is written; the kernel reclaims the device faster than we can)
	abg_overlap_destroy(ov);
	return status;
}
