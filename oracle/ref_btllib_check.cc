// oracle/ref_btllib_check.cc -- TEST INFRASTRUCTURE ONLY: prints what oracle/shim/btllib/ computes, for tests/test_rresolver_oracle.py.
//   btllib_check hash K H SEQ       one line per ACGT-only k-mer of SEQ: position, then its H hash values
//   btllib_check bloom BYTES H K    sequences on stdin, one per line: inserts the lines before an empty line, then prints for
//                                   every later line how many of its k-mers the filter holds; last line: the filter's popcount
//   btllib_check bloomdump BYTES H K SPAN OUT   sequences on stdin, one per line: inserts the first SPAN characters of every line
//                                   that long at least K (RResolver/BloomFilters.cpp:191-193), writes the filter's array to OUT
#include "btllib/bloom_filter.hpp"

#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

int main(int argc, char** argv)
{
	if (argc == 5 && !strcmp(argv[1], "hash")) {
		const unsigned k = (unsigned)atoi(argv[2]), h = (unsigned)atoi(argv[3]);
		btllib::NtHash nt(std::string(argv[4]), h, k);
		while (nt.roll()) {
			printf("%zu", nt.get_pos());
			for (unsigned i = 0; i < h; i++) printf(" %" PRIu64, nt.hashes()[i]);
			printf("\n");
		}
		return 0;
	}
	if (argc == 5 && !strcmp(argv[1], "bloom")) {
		btllib::KmerBloomFilter bf((size_t)strtoull(argv[2], 0, 10), (unsigned)atoi(argv[3]), (unsigned)atoi(argv[4]));
		std::string line;
		bool querying = false;
		while (std::getline(std::cin, line)) {
			if (line.empty()) { querying = true; continue; }
			if (!querying) bf.insert(line);
			else printf("%u\n", bf.contains(line));
		}
		printf("%" PRIu64 " %zu\n", bf.get_pop_cnt(), bf.get_bytes());
		return 0;
	}
	if (argc == 7 && !strcmp(argv[1], "bloomdump")) {
		btllib::KmerBloomFilter bf((size_t)strtoull(argv[2], 0, 10), (unsigned)atoi(argv[3]), (unsigned)atoi(argv[4]));
		const size_t span = (size_t)strtoull(argv[5], 0, 10);
		std::string line;
		while (std::getline(std::cin, line)) {
			const std::string seq = line.substr(0, span);
			if (seq.size() >= bf.get_k()) bf.insert(seq);
		}
		FILE* o = fopen(argv[6], "wb");
		if (!o) return 1;
		for (size_t i = 0; i < bf.get_bytes(); i++) fputc(bf.get_bloom_filter().data()[i].load(), o);
		fclose(o);
		printf("%" PRIu64 " %zu\n", bf.get_pop_cnt(), bf.get_bytes());
		return 0;
	}
	fprintf(stderr, "usage: btllib_check hash K H SEQ | bloom BYTES H K | bloomdump BYTES H K SPAN OUT\n");
	return 2;
}
