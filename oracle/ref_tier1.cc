// ref_tier1.cc -- harness around the reference's OWN headers, compiled from where
// they lie under /root/reference into oracle/_ref/tier1 (never copied).
// TEST INFRASTRUCTURE ONLY.  Emits golden vectors used to pin oracle/abg_oracle.c:
//   tier1 tables                       -> seedTab / msTab consistency table
//   tier1 hash K H SEQ                 -> per valid k-mer: pos and H hashes (ntHashIterator-free:
//                                         NTC64 + NTE64 exactly as BloomDBG/RollingHash.h uses them)
//   tier1 counters M K H KC < seqs     -> inserts every k-mer of every line into a
//                                         CountingBloomFilter<uint8_t>(M,H,K,KC) and dumps the raw array
#include "vendor/nthash/nthash.hpp"
#include "vendor/btl_bloomfilter/CountingBloomFilter.hpp"
#include "vendor/btl_bloomfilter/BloomFilter.hpp"
#include "Bloom/HashAgnosticCascadingBloom.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <sstream>
#include <fstream>

static bool acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

template <typename F>
static void each_kmer(const std::string& s, unsigned k, unsigned H, F f)
{
	if (s.size() < k) return;
	uint64_t fh = 0, rh = 0, hashes[32];
	bool roll = false;
	for (size_t pos = 0; pos + k <= s.size();) {
		size_t bad = std::string::npos;
		for (size_t i = pos; i < pos + k; i++) if (!acgt(s[i])) bad = i;
		if (bad != std::string::npos) { roll = false; pos = bad + 1; continue; }
		uint64_t h;
		if (!roll) { h = NTC64(s.c_str() + pos, k, fh, rh); roll = true; }
		else h = NTC64((unsigned char)s[pos - 1], (unsigned char)s[pos + k - 1], k, fh, rh);
		hashes[0] = h;
		for (unsigned i = 1; i < H; i++) hashes[i] = NTE64(h, k, i);
		f(pos, hashes);
		pos++;
	}
}

int main(int argc, char** argv)
{
	if (argc >= 2 && !strcmp(argv[1], "tables")) {
		for (int c = 0; c < 256; c++) {
			printf("%d %llu", c, (unsigned long long)seedTab[c]);
			for (int n = 0; n < 200; n += 7)
				printf(" %llu", (unsigned long long)(msTab31l[c][n % 31] | msTab33r[c][n % 33]));
			printf("\n");
		}
		return 0;
	}
	if (argc >= 5 && !strcmp(argv[1], "hash")) {
		unsigned k = atoi(argv[2]), H = atoi(argv[3]);
		std::string s(argv[4]);
		each_kmer(s, k, H, [&](size_t pos, const uint64_t* h) {
			printf("%zu", pos);
			for (unsigned i = 0; i < H; i++) printf(" %llu", (unsigned long long)h[i]);
			printf("\n");
		});
		return 0;
	}
	if (argc >= 6 && !strcmp(argv[1], "counters")) {
		size_t m = strtoull(argv[2], 0, 10);
		unsigned k = atoi(argv[3]), H = atoi(argv[4]), kc = atoi(argv[5]);
		CountingBloomFilter<uint8_t> bloom(m, H, k, kc);
		std::string line;
		while (std::getline(std::cin, line))
			each_kmer(line, k, H, [&](size_t, const uint64_t* h) { bloom.insert(h); });
		if (argc >= 7) bloom.storeFilter(argv[6]);
		fprintf(stderr, "size=%zu popcount=%zu filtered=%zu\n", bloom.size(), bloom.popCount(), bloom.filtered_popcount());
		for (size_t i = 0; i < bloom.size(); i++) putchar(bloom[i]);
		return 0;
	}
	if (argc >= 6 && !strcmp(argv[1], "cascade")) {
		// tier1 cascade LEVEL_BITS K H LEVELS [file] < seqs : HashAgnosticCascadingBloom as built by
		// `abyss-bloom build -t rolling-hash -l LEVELS` (Bloom/bloom.cc:585-602); dumps the last level
		size_t bits = strtoull(argv[2], 0, 10);
		unsigned k = atoi(argv[3]), H = atoi(argv[4]), L = atoi(argv[5]);
		HashAgnosticCascadingBloom bloom(bits, H, L, k);
		std::string line;
		while (std::getline(std::cin, line))
			each_kmer(line, k, H, [&](size_t, const uint64_t* h) { bloom.insert(h); });
		fprintf(stderr, "size=%zu popcount=%zu\n", bloom.size(), bloom.popcount());
		if (argc >= 7) { std::ofstream f(argv[6], std::ios::binary); f << bloom; }
		std::ostringstream os;
		os << bloom;
		std::string all = os.str();
		size_t pos = all.find("[HeaderEnd]\n");
		fwrite(all.data() + pos + 12, 1, all.size() - pos - 12, stdout);
		return 0;
	}
	fprintf(stderr, "usage: tier1 tables | hash K H SEQ | counters M K H KC [file] < seqs\n");
	return 1;
}
