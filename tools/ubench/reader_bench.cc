// how fast SequenceReader hands out the blocks of a FASTQ file (no device): tools/ubench/reader_bench FILE [threads] [-q N]
#include "../../abyss_amd/csrc/host/fasta_reader.h"
#include <chrono>
int main(int argc, char** argv)
{
	abghost::ReaderOptions o;
	unsigned threads = argc > 2 ? atoi(argv[2]) : 8;
	if (argc > 3) o.qualityThreshold = atoi(argv[3]);
	auto t0 = std::chrono::steady_clock::now();
	abghost::SequenceReader in(argv[1], o, threads);
	abghost::SequenceReader::Block b;
	uint64_t n = 0, bases = 0, idb = 0, h = 1469598103934665603ull;
	while (in.next_block(b)) {
		n += b.seq_end.size(); bases += b.seqs.size(); idb += b.ids.size() + b.comments.size();
		for (size_t i = 0; i < b.seqs.size(); i += 4099) h = (h ^ (unsigned char)b.seqs[i]) * 1099511628211ull;
		for (size_t i = 0; i < b.ids.size(); i += 997) h = (h ^ (unsigned char)b.ids[i]) * 1099511628211ull;
		h = (h ^ b.seq_end.size() ^ (b.seq_end.empty() ? 0 : b.seq_end.back()) ^ (b.id_end.empty() ? 0 : b.id_end.back()) ^ (b.com_end.empty() ? 0 : b.com_end.back())) * 1099511628211ull;
	}
	double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("%llu records, %llu bases, %llu id bytes, digest %016llx, %.3f s\n", (unsigned long long)n, (unsigned long long)bases, (unsigned long long)idb, (unsigned long long)h, s);
}
