// rresolver_check.cc -- TEST INFRASTRUCTURE ONLY.
// The host side of the drop-in abyss-rresolver-short (abyss_amd/csrc/host/rresolver_core.h: options, graph reader and surgery,
// read statistics, path support, writers) over the read filter run SERIALLY on the CPU -- the product's device logic of
// abyss_amd/csrc/abg_rr.h, one lane at a time -- so that the "-m 'not gpu'" suite can compare whole runs with the reference's on a
// machine without a GPU.  Never shipped; the product binary calls abg_rr_* on the GPU.
//   rresolver_check [options as abyss-rresolver-short]
//   rresolver_check --filter BYTES H R SPAN OUT   reads on stdin, one per line: the filter's array written to OUT (parity tests)
#include "../../abyss_amd/csrc/host/rresolver_core.h"
#include "../../abyss_amd/csrc/abg_rr.h"

namespace {

struct SerialFilter : abgrr::ReadFilter {
	abg::RRParams p;
	std::vector<uint32_t> bits;
	uint64_t nbytes = 0;
	void create(uint64_t bytes, unsigned hash_num, unsigned r) override
	{
		nbytes = (bytes + 7) / 8 * 8;
		p = abg::make_rr_params(r, hash_num, nbytes);
		bits.assign((nbytes + 3) / 4, 0);
	}
	void insert(const char* seqs, const uint64_t* off, uint64_t n, uint32_t max_bases, const uint32_t* lengths, uint32_t nlen) override
	{
		for (uint64_t i = 0; i < n; i++) {
			const uint64_t len = off[i + 1] - off[i];
			bool want = nlen == 0;
			for (uint32_t j = 0; j < nlen && !want; j++) want = len == lengths[j];
			if (!want || len < p.r) continue;
			const unsigned char* s = (const unsigned char*)seqs + off[i];
			abg::rr_insert_record(p, [&](uint32_t j) { return (unsigned)s[j]; }, (uint32_t)std::min<uint64_t>(len, max_bases), bits.data());
		}
	}
	void contains(const char* seqs, const uint64_t* off, uint64_t n, uint32_t* found) override
	{
		for (uint64_t i = 0; i < n; i++)
			found[i] = abg::rr_contains_share(p, (const unsigned char*)seqs + off[i], off[i + 1] - off[i], 0, 1, bits.data());
	}
	uint64_t popcount() override { uint64_t c = 0; for (uint32_t w : bits) c += (uint64_t)__builtin_popcount(w); return c; }
	uint64_t bytes() override { return nbytes; }
};

} // namespace

int main(int argc, char** argv)
{
	if (argc == 7 && !strcmp(argv[1], "--filter")) {
		SerialFilter f;
		f.create(strtoull(argv[2], 0, 10), (unsigned)atoi(argv[3]), (unsigned)atoi(argv[4]));
		const uint32_t span = (uint32_t)atoi(argv[5]);
		std::string line;
		char buf[1 << 16];
		while (fgets(buf, sizeof buf, stdin)) {
			line = buf;
			while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
			const uint64_t off[2] = { 0, line.size() };
			f.insert(line.data(), off, 1, span, nullptr, 0);
		}
		FILE* o = fopen(argv[6], "wb");
		if (!o) return 1;
		fwrite(f.bits.data(), 1, f.nbytes, o);
		fclose(o);
		printf("%llu %llu\n", (unsigned long long)f.popcount(), (unsigned long long)f.nbytes);
		return 0;
	}
	abgrr::Options opt;
	int status = 0;
	if (!abgrr::parse_options(argc, argv, opt, &status)) return status;
	SerialFilter filter;
	abgrr::Resolver resolver(opt, filter);
	return resolver.run();
}
