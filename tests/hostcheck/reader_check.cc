// reader_check.cc -- TEST INFRASTRUCTURE ONLY: dumps the records produced by the drop-in
// binary's reader (abyss_amd/csrc/host/fasta_reader.h) as "id<TAB>sequence" lines, to be
// compared with oracle/_ref/ref_reader (the reference's own FastaReader) and with the
// golden file tests/golden/reader_*.tsv.
#include "../../abyss_amd/csrc/host/fasta_reader.h"
#include <cstdio>
int main(int argc, char** argv)
{
	abghost::ReaderOptions o;
	const char* path = NULL;
	unsigned threads = 1;
	bool count_only = false, prefetch = false, blocks = false;
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "-q")) o.qualityThreshold = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-Q")) o.internalQThreshold = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--no-chastity")) o.chastityFilter = 0;
		else if (!strcmp(argv[i], "--no-trim-masked")) o.trimMasked = 0;
		else if (!strcmp(argv[i], "--illumina-quality")) o.qualityOffset = 64;
		else if (!strcmp(argv[i], "-j")) threads = (unsigned)atoi(argv[++i]); // SequenceReader: blocks parsed in parallel
		else if (!strcmp(argv[i], "--count")) count_only = true;
		else if (!strcmp(argv[i], "--blocks")) blocks = true;     // the records wholesale (SequenceReader::next_block), as the host binary's PASS 1 takes them
		else if (!strcmp(argv[i], "--prefetch")) prefetch = true; // compressed input inflated ahead (abghost::Prefetch)
		else path = argv[i];
	}
	if (prefetch) abghost::Prefetch::get().start({ path });
	abghost::SequenceReader in(path, o, threads);
	std::string id, comment, seq;
	unsigned long long n = 0, bases = 0, sum = 0;
	if (blocks && in.has_blocks()) {
		abghost::SequenceReader::Block b;
		while (in.next_block(b))
			for (size_t i = 0; i < b.seq_end.size(); i++) {
				const size_t ia = i ? b.id_end[i - 1] : 0, sa = i ? b.seq_end[i - 1] : 0;
				printf("%s\t%s\n", b.ids.substr(ia, b.id_end[i] - ia).c_str(), b.seqs.substr(sa, b.seq_end[i] - sa).c_str());
			}
		return 0;
	}
	while (in.read(id, comment, seq)) {
		if (!count_only) { printf("%s\t%s\n", id.c_str(), seq.c_str()); continue; }
		n++; bases += seq.size();
		for (char c : id) sum = sum * 131 + (unsigned char)c;
		for (char c : seq) sum = sum * 131 + (unsigned char)c;
	}
	if (count_only) printf("%llu records, %llu bases, checksum %llu\n", n, bases, sum);
	return 0;
}
