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
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "-q")) o.qualityThreshold = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-Q")) o.internalQThreshold = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--no-chastity")) o.chastityFilter = 0;
		else if (!strcmp(argv[i], "--no-trim-masked")) o.trimMasked = 0;
		else if (!strcmp(argv[i], "--illumina-quality")) o.qualityOffset = 64;
		else path = argv[i];
	}
	abghost::FastaReader in(path, o);
	std::string id, comment, seq;
	while (in.read(id, comment, seq)) printf("%s\t%s\n", id.c_str(), seq.c_str());
	return 0;
}
