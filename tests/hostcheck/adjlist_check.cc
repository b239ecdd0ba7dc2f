// adjlist_check.cc -- TEST INFRASTRUCTURE ONLY.
// The host side of the drop-in AdjList (abyss_amd/csrc/host/adjlist_core.h: options, reader,
// suffix-array overlaps, writers) over the k-1 join run SERIALLY by tests/hostcheck
// (hc_overlap_join: the product's device logic through the serial backend), so that the
// "-m 'not gpu'" suite can compare whole AdjList outputs with the reference's on a machine
// without a GPU.  Never shipped; the product binary calls abg_overlap_join on the GPU.
#include "../../abyss_amd/csrc/host/adjlist_core.h"

extern "C" int hc_overlap_join(uint32_t overlap, uint64_t n, const uint64_t* head, const uint64_t* tail, int ss, uint64_t* off, uint32_t* tgt, uint64_t* ne);

int main(int argc, char** argv)
{
	abgadj::Options opt;
	int device = 0, status = 0;
	if (!abgadj::parse_options(argc, argv, opt, &device, &status)) return status;
	abgadj::Join join = [&](uint32_t overlap, uint64_t n, const uint64_t* head, const uint64_t* tail, bool ss,
	                        std::vector<uint64_t>& off, std::vector<uint32_t>& tgt) {
		uint64_t ne = 0;
		off.assign(2 * n + 1, 0);
		hc_overlap_join(overlap, n, head, tail, ss, off.data(), nullptr, &ne);
		tgt.assign(ne, 0);
		hc_overlap_join(overlap, n, head, tail, ss, off.data(), tgt.data(), &ne);
	};
	return abgadj::run(opt, join, stdout);
}
