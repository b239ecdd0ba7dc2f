// oracle/ref_btllib_check.cc -- TEST INFRASTRUCTURE ONLY: prints what oracle/shim/btllib/ computes, for tests/test_rresolver_oracle.py.
//   btllib_check hash K H SEQ       one line per ACGT-only k-mer of SEQ: position, then its H hash values
//   btllib_check bloom BYTES H K    sequences on stdin, one per line: inserts the lines before an empty line, then prints for
//                                   every later line how many of its k-mers the filter holds; last line: the filter's popcount
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
	fprintf(stderr, "usage: btllib_check hash K H SEQ | bloom BYTES H K\n");
	return 2;
}
