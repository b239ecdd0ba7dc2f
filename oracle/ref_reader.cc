// ref_reader.cc -- dumps what the reference's OWN FastaReader (DataLayer/FastaReader.cpp,
// compiled from /root/reference into oracle/_ref/ref_reader) hands to BloomDBG for a file:
// one "id<TAB>sequence" line per record, with the FOLD_CASE flag BloomDBG uses
// (BloomIO.h:61, bloom-dbg.h:917).  TEST INFRASTRUCTURE ONLY: pins the host-side reader
// of the drop-in binary (abyss_amd/csrc/host/fasta_reader.h).
// usage: ref_reader [-q N] [-Q N] [--no-chastity] [--no-trim-masked] [--illumina-quality] FILE
#include "config.h"
#include "DataLayer/FastaReader.h"
#include "DataLayer/Options.h"
#include <cstdlib>
#include <cstring>
#include <iostream>
int main(int argc, char** argv)
{
	const char* path = NULL;
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "-q")) opt::qualityThreshold = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-Q")) opt::internalQThreshold = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--no-chastity")) opt::chastityFilter = 0;
		else if (!strcmp(argv[i], "--no-trim-masked")) opt::trimMasked = 0;
		else if (!strcmp(argv[i], "--illumina-quality")) opt::qualityOffset = 64;
		else path = argv[i];
	}
	FastaReader in(path, FastaReader::FOLD_CASE);
	for (FastaRecord rec; in >> rec;)
		std::cout << rec.id << '\t' << rec.seq << '\n';
	return 0;
}
