// bloom_main.cc -- `abyss-bloom build` on the MI355X: the two Bloom-filter builders of the
// reference's Bloom/bloom.cc that share the abyss-bloom-dbg hot path,
//   build -t counting      (bloom.cc:605-620)  -> [BTLCountingBloomFilter_v1] file for abyss-bloom-dbg -i
//   build -t rolling-hash  (bloom.cc:585-602)  -> last level of a HashAgnosticCascadingBloom,
//                                                [BTLBloomFilter_v1] file
// over the C ABI (include/abyss_amd.h).  Other abyss-bloom commands (union, intersect, info,
// compare, graph, kmers, trim) and `-t konnector` filters are not part of this path.
#include "../../../include/abyss_amd.h"
#include "fasta_reader.h"
#include "si_bytes.h"

#include <algorithm>
#include <thread>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <string>
#include <vector>

#define PROGRAM "abyss-bloom"

static abghost::ReaderOptions ropt;
static const struct option longopts[] = {
	{ "bloom-size", required_argument, NULL, 'b' }, { "threads", required_argument, NULL, 'j' },
	{ "kmer", required_argument, NULL, 'k' }, { "num-hashes", required_argument, NULL, 'H' },
	{ "levels", required_argument, NULL, 'l' }, { "chastity", no_argument, &ropt.chastityFilter, 1 },
	{ "no-chastity", no_argument, &ropt.chastityFilter, 0 }, { "trim-masked", no_argument, &ropt.trimMasked, 1 },
	{ "no-trim-masked", no_argument, &ropt.trimMasked, 0 }, { "trim-quality", required_argument, NULL, 'q' },
	{ "bloom-type", required_argument, NULL, 't' }, { "standard-quality", no_argument, &ropt.qualityOffset, 33 },
	{ "illumina-quality", no_argument, &ropt.qualityOffset, 64 }, { "verbose", no_argument, NULL, 'v' },
	{ NULL, 0, NULL, 0 }
};

static void check(int rc, abg_ctx* ctx, const char* what)
{
	if (rc == ABG_OK) return;
	fprintf(stderr, PROGRAM ": %s: %s\n", what, abg_last_error(ctx));
	exit(EXIT_FAILURE);
}

int main(int argc, char** argv)
{
	if (argc < 2 || strcmp(argv[1], "build")) {
		fprintf(stderr, "Usage: " PROGRAM " build -t counting|rolling-hash -k N -b N [-H N] [-l N] <OUTPUT_BLOOM_FILE> <READS>...\n"
		                "(only `build -t counting|rolling-hash' is provided by this GPU build)\n");
		return EXIT_FAILURE;
	}
	optind = 2;
	uint64_t bloomSize = 500ull << 20; // bloom.cc: default 500M
	unsigned k = 0, H = 1, levels = 1;
	std::string type = "konnector";
	int verbose = 0;
	for (int c; (c = getopt_long(argc, argv, "b:B:j:k:H:l:q:t:v", longopts, NULL)) != -1;) {
		switch (c) {
		case 'b': if (!si_to_bytes(optarg, &bloomSize)) { fprintf(stderr, PROGRAM ": invalid option: `-b%s'\n", optarg); return EXIT_FAILURE; } break;
		case 'k': k = (unsigned)atoi(optarg); break;
		case 'H': H = (unsigned)atoi(optarg); break;
		case 'l': levels = (unsigned)atoi(optarg); break;
		case 'q': ropt.qualityThreshold = atoi(optarg); break;
		case 't': type = optarg; break;
		case 'v': verbose++; break;
		case 'B': case 'j': break;
		case '?': return EXIT_FAILURE;
		}
	}
	if (type != "counting" && type != "rolling-hash") {
		fprintf(stderr, PROGRAM ": this build provides `-t counting' and `-t rolling-hash' only (saw `%s')\n", type.c_str());
		return EXIT_FAILURE;
	}
	if (k == 0) { fprintf(stderr, PROGRAM ": missing mandatory option `-k'\n"); return EXIT_FAILURE; }
	if (type == "counting" && levels > 1) { fprintf(stderr, PROGRAM ": `-l' is not supported when using `-t counting'\n"); return EXIT_FAILURE; }
	if (argc - optind < 2) { fprintf(stderr, PROGRAM ": missing arguments\n"); return EXIT_FAILURE; }
	std::string outputPath = argv[optind++];

	abg_params p;
	abg_params_init(&p);
	p.k = k; p.num_hashes = H; p.min_cov = 0; p.verbose = verbose;
	if (type == "counting") {
		p.counters = bloomSize; // CountingBloomFilter<uint8_t>(bytes, H, k, 0), bloom.cc:610
	} else {
		uint64_t bits = bloomSize * 8 / levels; // roundUpToMultiple(bits / levels, 64), bloom.cc:590
		if (bits % 64) bits += 64 - bits % 64;
		p.counters = bits;
		p.cascade_levels = levels;
	}
	abg_ctx* ctx = NULL;
	if (abg_create(&p, &ctx) != ABG_OK) { fprintf(stderr, PROGRAM ": %s\n", abg_last_error(NULL)); return EXIT_FAILURE; }
	uint64_t size = 0;
	abg_filter_size(ctx, &size);
	std::string id, comment, seq, seqs;
	std::vector<uint64_t> off{ 0 };
	for (int i = optind; i < argc; i++) { // BloomDBG::loadFile for each file, bloom.cc:596-597,613-614
		if (verbose) fprintf(stderr, "Reading `%s'...\n", argv[i]);
		abghost::SequenceReader in(argv[i], ropt, std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
		while (in.read(id, comment, seq)) {
			seqs += seq; off.push_back(seqs.size());
			if (seqs.size() >= (256u << 20)) { check(abg_load_seqs(ctx, seqs.data(), off.data(), off.size() - 1), ctx, "load"); seqs.clear(); off.assign(1, 0); }
		}
		if (off.size() > 1) { check(abg_load_seqs(ctx, seqs.data(), off.data(), off.size() - 1), ctx, "load"); seqs.clear(); off.assign(1, 0); }
	}
	FILE* f = fopen(outputPath.c_str(), "wb");
	if (!f) { fprintf(stderr, "error: `%s': %s\n", outputPath.c_str(), strerror(errno)); return EXIT_FAILURE; }
	if (type == "counting") {
		// CountingBloomFilter::storeHeader + raw counters (CountingBloomFilter.hpp:344-379), key order as cpptoml emits it
		std::vector<uint8_t> cnt(size);
		check(abg_counters_export(ctx, cnt.data()), ctx, "export");
		fprintf(stderr, "Writing a %llu byte filter to %s on disk.\n", (unsigned long long)size, outputPath.c_str());
		fprintf(f, "[BTLCountingBloomFilter_v1]\n\tBloomFilterSize = %llu\n\tHashNum = %u\n\tKmerSize = %u\n"
		           "\tBloomFilterSizeInBytes = %llu\n\tBitsPerCounter = 8\n[HeaderEnd]\n",
		    (unsigned long long)size, H, k, (unsigned long long)size);
		fwrite(cnt.data(), 1, cnt.size(), f);
	} else {
		// operator<< of the last level (HashAgnosticCascadingBloom.h:143-150; BloomFilter::writeHeader, BloomFilter.hpp:261-294)
		std::vector<uint8_t> bits(size / 8);
		check(abg_cascade_export(ctx, levels - 1, bits.data()), ctx, "export");
		fprintf(f, "[BTLBloomFilter_v1]\n\tnEntry = 0\n\tdFPR = 0.0000000000000000\n\tEntry = 0\n"
		           "\tBloomFilterSizeInBytes = %llu\n\tBloomFilterSize = %llu\n\tHashNum = %u\n\tKmerSize = %u\n[HeaderEnd]\n",
		    (unsigned long long)(size / 8), (unsigned long long)size, H, k);
		fwrite(bits.data(), 1, bits.size(), f);
	}
	fclose(f);
	abg_destroy(ctx);
	return EXIT_SUCCESS;
}
