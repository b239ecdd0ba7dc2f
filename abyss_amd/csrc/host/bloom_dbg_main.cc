// bloom_dbg_main.cc -- `abyss-bloom-dbg`, the drop-in host binary for the abyss-pe
// pipeline: the command line, messages, exit codes and FASTA output of the reference's
// BloomDBG/bloom-dbg.cc (options table :128,142-175; main :389-558) on top of the C ABI of
// libabyss_amd.so (include/abyss_amd.h).  All assembly work happens on the GPU; this file
// only parses options, reads sequence files and prints records.
//
// Every option of the reference's table is served, -g (GraphViz dump, abg_output_graph_seqs) included.
#include "../../../include/abyss_amd.h"
#include "fasta_reader.h"
#include "si_bytes.h"

#include <csignal>
#include <sys/wait.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/prctl.h>
#include <sys/wait.h>

#include <algorithm>
#include <cmath>
#include <memory>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <getopt.h>
#include <unistd.h>
#include <string>
#include <vector>

#define PROGRAM "abyss-bloom-dbg"

static const char VERSION_MESSAGE[] =
    PROGRAM " (ABySS) 2.3.10 [abyss_amd: MI355X-native unitig stage]\n"
            "Interface of the ABySS program written by Ben Vandervalk, Shaun Jackman, Hamid Mohamadi,\n"
            "Justin Chu, and Anthony Raymond.\n";

static const char USAGE_MESSAGE[] =
    "Usage: " PROGRAM " -b <bloom_size> -H <bloom_hashes> -k <kmer_size> \\\n"
    "    [options] <FASTQ> [FASTQ]... > assembly.fasta\n"
    "\n"
    "Perform a de Bruijn graph assembly of the given FASTQ files on an AMD MI355X.\n"
    "\n"
    "  -b  --bloom-size=N           overall memory budget in bytes; suffixes k, M, G [required]\n"
    "      --chastity / --no-chastity   discard unchaste reads [default] / keep them\n"
    "      --help                   display this help and exit\n"
    "      --gpus=N                 [extension] spread the job over N MI355X of this node: the counting\n"
    "                               filter is range-partitioned over them (RCCL over xGMI), output as on one\n"
    "  -H  --num-hashes=N           number of Bloom filter hash functions [4]\n"
    "  -i  --input-bloom=FILE       load the counting Bloom filter from FILE\n"
    "  -j, --threads=N              host threads parsing FASTQ input [all, up to 16]; the GPU does the assembly\n"
    "      --trim-masked / --no-trim-masked\n"
    "  -k, --kmer=N                 the size of a k-mer [<=192]\n"
    "      --kc=N                   ignore k-mers having a count < N [2]\n"
    "  -o, --out=FILE               write the contigs to FILE [STDOUT]\n"
    "  -q, --trim-quality=N         trim bases from the ends of reads whose quality is less than N\n"
    "  -Q, --mask-quality=N         mask all low quality bases as `N'\n"
    "      --standard-quality / --illumina-quality\n"
    "  -t, --trim-length=N          max branch length to trim, in k-mers [k]\n"
    "  -T, --trace-file=FILE        write debugging info about each contig to FILE\n"
    "      --read-log=FILE          write outcome of processing each read to FILE\n"
    "  -v, --verbose                display verbose output\n"
    "      --version                output version information and exit\n";

enum { OPT_HELP = 1, OPT_VERSION, QR_SEED, MIN_KMER_COV, CHECKPOINT, KEEP_CHECKPOINT, CHECKPOINT_PREFIX, READ_LOG, OPT_GPUS };
static abghost::ReaderOptions ropt;
// --gpus: the other ranks.  If one of them dies, the rest would wait for it in a collective for
// ever: the SIGCHLD handler of rank 0 takes the whole job down instead.
static pid_t g_children[ABG_MAX_RANKS];
static volatile sig_atomic_t g_nchildren = 0, g_children_left = 0;
static void on_sigchld(int)
{
	int st;
	pid_t pid;
	// (only the ranks are reaped here: the decompressors of compressed inputs are children too, and
	// their reader waits for them itself)
	for (int i = 0; i < g_nchildren; i++) {
		if (g_children[i] <= 0) continue;
		pid = waitpid(g_children[i], &st, WNOHANG);
		if (pid <= 0) continue;
		g_children[i] = -1;
		g_children_left--;
		if (WIFEXITED(st) && WEXITSTATUS(st) == 0) continue;
		static const char msg[] = "abyss-bloom-dbg: a rank of the multi-GPU run failed\n";
		if (write(2, msg, sizeof msg - 1)) {}
		for (int j = 0; j < g_nchildren; j++) if (g_children[j] > 0) kill(g_children[j], SIGKILL);
		_exit(EXIT_FAILURE);
	}
}
static const char shortopts[] = "b:C:g:H:i:j:k:K:o:q:Q:R:s:t:T:v";
static const struct option longopts[] = {
	{ "bloom-size", required_argument, NULL, 'b' }, { "min-coverage", required_argument, NULL, 'c' },
	{ "cov-track", required_argument, NULL, 'C' }, { "chastity", no_argument, &ropt.chastityFilter, 1 },
	{ "no-chastity", no_argument, &ropt.chastityFilter, 0 }, { "checkpoint", required_argument, NULL, CHECKPOINT },
	{ "keep-checkpoint", no_argument, NULL, KEEP_CHECKPOINT }, { "checkpoint-prefix", required_argument, NULL, CHECKPOINT_PREFIX },
	{ "graph", required_argument, NULL, 'g' }, { "gpus", required_argument, NULL, OPT_GPUS }, { "num-hashes", required_argument, NULL, 'H' },
	{ "input-bloom", required_argument, NULL, 'i' }, { "help", no_argument, NULL, OPT_HELP },
	{ "threads", required_argument, NULL, 'j' }, { "trim-masked", no_argument, &ropt.trimMasked, 1 },
	{ "no-trim-masked", no_argument, &ropt.trimMasked, 0 }, { "kmer", required_argument, NULL, 'k' },
	{ "kc", required_argument, NULL, MIN_KMER_COV }, { "single-kmer", required_argument, NULL, 'K' },
	{ "out", required_argument, NULL, 'o' }, { "trim-quality", required_argument, NULL, 'q' },
	{ "mask-quality", required_argument, NULL, 'Q' }, { "standard-quality", no_argument, &ropt.qualityOffset, 33 },
	{ "illumina-quality", no_argument, &ropt.qualityOffset, 64 }, { "qr-seed", required_argument, NULL, QR_SEED },
	{ "read-log", required_argument, NULL, READ_LOG }, { "ref", required_argument, NULL, 'R' },
	{ "spaced-seed", required_argument, NULL, 's' }, { "trim-length", required_argument, NULL, 't' },
	{ "trace-file", required_argument, NULL, 'T' }, { "verbose", no_argument, NULL, 'v' },
	{ "version", no_argument, NULL, OPT_VERSION }, { NULL, 0, NULL, 0 }
};

// SIToBytes, Common/StringUtil.h:181-219: k/M/G are powers of 1024

struct Chunk { // one batch of sequences for the C ABI; the ids likewise as one string and their ends
	std::string seqs;
	std::vector<uint64_t> off{ 0 };
	std::string idbuf;
	std::vector<uint64_t> id_end;
	void add(const std::string& id, const std::string& s)
	{
		seqs += s; off.push_back(seqs.size());
		idbuf += id; id_end.push_back(idbuf.size());
	}
	// a parser thread's block of records in one go (SequenceReader::next_block)
	void add_block(const abghost::SequenceReader::Block& b)
	{
		const uint64_t sb = seqs.size(), ib = idbuf.size();
		seqs += b.seqs; idbuf += b.ids;
		off.reserve(off.size() + b.seq_end.size()); id_end.reserve(id_end.size() + b.id_end.size());
		for (size_t e : b.seq_end) off.push_back(sb + e);
		for (size_t e : b.id_end) id_end.push_back(ib + e);
	}
	// ... taken over as it is: the block's strings become the chunk's (no copy of the bases)
	void take_block(abghost::SequenceReader::Block& b)
	{
		seqs = std::move(b.seqs); idbuf = std::move(b.ids); id_end.assign(b.id_end.begin(), b.id_end.end());
		off.clear(); off.reserve(b.seq_end.size() + 1); off.push_back(0);
		off.insert(off.end(), b.seq_end.begin(), b.seq_end.end());
	}
	void drop_seqs() { std::string().swap(seqs); std::vector<uint64_t>().swap(off); } // (the ids stay: n(), id())
	size_t n() const { return id_end.size(); }
	std::string id(size_t i) const { const uint64_t a = i ? id_end[i - 1] : 0; return idbuf.substr(a, id_end[i] - a); }
	size_t bytes() const { return seqs.size() + idbuf.size() + 16 * n(); }
	void clear() { seqs.clear(); off.assign(1, 0); idbuf.clear(); id_end.clear(); }
};

struct Output {
	FILE* out; FILE* trace; const Chunk* chunk; unsigned k;
	FILE* checkpoint = NULL; // duplicate FASTA output for checkpoints (bloom-dbg.h:607-609,919-926)
	// one assemble call over several chunks (abg_assemble_seqs_v): read index -> (chunk, index in it)
	const std::vector<Chunk>* chunks = NULL;
	std::vector<uint64_t> first; // [chunks + 1] index of each chunk's first read
	void locate(uint64_t read_index, const Chunk*& c, uint64_t& i) const
	{
		if (!chunks) { c = chunk; i = read_index; return; }
		const size_t q = (size_t)(std::upper_bound(first.begin(), first.end(), read_index) - first.begin()) - 1;
		c = &(*chunks)[q]; i = read_index - first[q];
	}
};
static const char* ext_str(int c)
{
	static const char* s[] = { "AMBI_IN", "AMBI_OUT", "DEAD_END", "CYCLE", "LENGTH_LIMIT" };
	return s[c];
}
static void on_contig(void* user, const abg_contig* c)
{
	Output* o = (Output*)user;
	const Chunk* ch; uint64_t ri;
	o->locate(c->read_index, ch, ri);
	const std::string rid = ch->id(ri);
	if (!c->redundant) // printContig, bloom-dbg.h:455-487
	{
		fprintf(o->out, ">%llu %u %u read:%s\n%s\n", (unsigned long long)c->contig_id, c->length, c->coverage, rid.c_str(), c->seq);
		if (o->checkpoint)
			fprintf(o->checkpoint, ">%llu %u %u read:%s\n%s\n", (unsigned long long)c->contig_id, c->length, c->coverage, rid.c_str(), c->seq);
	}
	if (o->trace) { // ContigRecord operator<<, bloom-dbg.h:229-254
		if (c->redundant) fputs("NA\t", o->trace); else fprintf(o->trace, "%llu\t", (unsigned long long)c->contig_id);
		fprintf(o->trace, "%u\t%d\t%s\t", c->length, c->redundant, rid.c_str());
		if (c->left_ext > 0) fprintf(o->trace, "%s\t%u\t", ext_str(c->left_code), c->left_ext); else fputs("NA\tNA\t", o->trace);
		if (c->right_ext > 0) fprintf(o->trace, "%s\t%u\t", ext_str(c->right_code), c->right_ext); else fputs("NA\tNA\t", o->trace);
		uint64_t a = ch->off[ri];
		fprintf(o->trace, "READ\t%u\t%.*s\n", o->k, (int)o->k, ch->seqs.c_str() + a + c->seed_pos);
	}
}
static void check(int rc, abg_ctx* ctx, const char* what)
{
	if (rc == ABG_OK) return;
	fprintf(stderr, PROGRAM ": %s: %s\n", what, abg_last_error(ctx));
	exit(EXIT_FAILURE);
}

// SpacedSeed::kmerPair (BloomDBG/SpacedSeed.h:18-26): K ones, a gap, K ones
static std::string spaced_seed_kmer_pair(unsigned k, unsigned K)
{
	std::string seed(k, '0');
	for (unsigned i = 0; i < K && i < k; i++) seed[i] = seed[k - 1 - i] = '1';
	return seed;
}
// SpacedSeed::qrSeed (SpacedSeed.h:40-52): position i is '0' when i is a quadratic residue mod len
static std::string spaced_seed_qr(unsigned len)
{
	std::string seed(len, '1');
	for (unsigned long i = 0; i < len; i++)
		for (unsigned long j = 1; j < len; j++)
			if (j * j % len == i) { seed[i] = '0'; break; }
	return seed;
}
// SpacedSeed::qrSeedPair (SpacedSeed.h:64-73): a QR seed, a gap, the mirrored QR seed
static std::string spaced_seed_qr_pair(unsigned k, unsigned len)
{
	std::string seed(k, '0'), q = spaced_seed_qr(len);
	for (unsigned i = 0; i < len && i < k; i++) seed[i] = seed[k - 1 - i] = q[i];
	return seed;
}

// ---- Bloom files and checkpoints (BloomDBG/Checkpoint.h) ----------------------------------------
static bool file_readable(const std::string& path) { FILE* f = fopen(path.c_str(), "rb"); if (f) fclose(f); return f != NULL; }
static void copy_file(const std::string& from, const std::string& to) // copyFile, Common/IOUtil.h
{
	FILE* a = fopen(from.c_str(), "rb"); FILE* b = fopen(to.c_str(), "wb");
	if (!a || !b) { fprintf(stderr, "error: `%s': %s\n", (a ? to : from).c_str(), strerror(errno)); exit(EXIT_FAILURE); }
	char buf[1 << 16]; size_t n;
	while ((n = fread(buf, 1, sizeof buf, a)) > 0) fwrite(buf, 1, n, b);
	fclose(a); fclose(b);
}
// header + payload of a [BTLCountingBloomFilter_v1] / [BTLBloomFilter_v1] file
// (CountingBloomFilter.hpp:262-329, BloomFilter.hpp:105-178)
static void read_bloom_file(const std::string& path, const char* magic, uint64_t& size, unsigned& hn, unsigned& ks,
    std::vector<uint8_t>& payload, bool bits)
{
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
	char* line = NULL; size_t cap = 0; ssize_t n;
	if ((n = getline(&line, &cap, f)) <= 0 || strncmp(line, magic, strlen(magic))) {
		fprintf(stderr, "ERROR: magic string does not match (likely version mismatch)\n"); exit(EXIT_FAILURE);
	}
	uint64_t bytes = 0; bool end = false;
	size = 0; hn = 0; ks = 0;
	while ((n = getline(&line, &cap, f)) > 0) {
		if (!strncmp(line, "[HeaderEnd]", 11)) { end = true; break; }
		char key[64]; unsigned long long v;
		if (sscanf(line, " %63[A-Za-z] = %llu", key, &v) == 2) {
			if (!strcmp(key, "BloomFilterSize")) size = v; else if (!strcmp(key, "HashNum")) hn = (unsigned)v;
			else if (!strcmp(key, "KmerSize")) ks = (unsigned)v; else if (!strcmp(key, "BloomFilterSizeInBytes")) bytes = v;
		}
	}
	if (!end || !size) { fprintf(stderr, "ERROR: pre-built bloom filter does not have the correct header end.\n"); exit(EXIT_FAILURE); }
	payload.resize(bytes ? bytes : (bits ? size / 8 : size));
	if (fread(payload.data(), 1, payload.size(), f) != payload.size()) { fprintf(stderr, "error: `%s': short read\n", path.c_str()); exit(EXIT_FAILURE); }
	fclose(f); free(line);
}
static void write_counting_file(abg_ctx* ctx, const std::string& path, unsigned k, unsigned H) // operator<<, CountingBloomFilter.hpp:344-379
{
	uint64_t size = 0; abg_filter_size(ctx, &size);
	std::vector<uint8_t> cnt(size);
	if (abg_counters_export(ctx, cnt.data()) != ABG_OK) { fprintf(stderr, PROGRAM ": %s\n", abg_last_error(ctx)); exit(EXIT_FAILURE); }
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
	fprintf(f, "[BTLCountingBloomFilter_v1]\n\tBloomFilterSize = %llu\n\tHashNum = %u\n\tKmerSize = %u\n"
	           "\tBloomFilterSizeInBytes = %llu\n\tBitsPerCounter = 8\n[HeaderEnd]\n",
	    (unsigned long long)size, H, k, (unsigned long long)size);
	fwrite(cnt.data(), 1, cnt.size(), f);
	fclose(f);
}
static void write_visited_file(abg_ctx* ctx, const std::string& path, unsigned k, unsigned H) // BloomFilter::writeHeader, BloomFilter.hpp:261-294
{
	uint64_t size = 0; abg_filter_size(ctx, &size);
	std::vector<uint8_t> bits(size / 8);
	if (abg_visited_export(ctx, bits.data()) != ABG_OK) { fprintf(stderr, PROGRAM ": %s\n", abg_last_error(ctx)); exit(EXIT_FAILURE); }
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
	fprintf(f, "[BTLBloomFilter_v1]\n\tnEntry = 0\n\tdFPR = 0.0000000000000000\n\tEntry = 0\n"
	           "\tBloomFilterSizeInBytes = %llu\n\tBloomFilterSize = %llu\n\tHashNum = %u\n\tKmerSize = %u\n[HeaderEnd]\n",
	    (unsigned long long)(size / 8), (unsigned long long)size, H, k);
	fwrite(bits.data(), 1, bits.size(), f);
	fclose(f);
}
static const char CK_FASTA[] = ".contigs.fa", CK_COUNTERS[] = ".counters.tsv", CK_DBG[] = ".dbg.bloom",
                  CK_VISITED[] = ".visited.bloom", CK_TMP[] = ".tmp";
static bool checkpoint_exists(const std::string& prefix) // Checkpoint.h:130-144
{
	return file_readable(prefix + CK_FASTA) && file_readable(prefix + CK_DBG) && file_readable(prefix + CK_VISITED) &&
	       file_readable(prefix + CK_COUNTERS);
}
// ABG_HOST_TIMING=1: wall time of the process so far at every phase boundary, on stderr
static double host_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static const double g_t0 = host_now();
static double g_in_load = 0, g_in_asm = 0;
static void host_mark(const char* what)
{
	static const bool on = getenv("ABG_HOST_TIMING") != NULL;
	if (on) fprintf(stderr, "[host %.3f s] %s (in abg_load_seqs %.3f s, in abg_assemble_seqs %.3f s)\n", host_now() - g_t0, what, g_in_load, g_in_asm);
}
static void do_rename(const std::string& a, const std::string& b, int verbose)
{
	if (verbose) fprintf(stderr, "\tMoving `%s' to `%s'\n", a.c_str(), b.c_str());
	if (rename(a.c_str(), b.c_str()) != 0) { perror("Error renaming file"); abort(); }
}
// createCheckpoint, Checkpoint.h:31-127
static void create_checkpoint(abg_ctx* ctx, const std::string& prefix, unsigned k, unsigned H, int verbose)
{
	if (verbose) fprintf(stderr, "Writing checkpoint data...\n");
	const std::string dbg = prefix + CK_DBG, vis = prefix + CK_VISITED, cnt = prefix + CK_COUNTERS, fa = prefix + CK_FASTA;
	if (verbose) fprintf(stderr, "\tWriting Bloom filter de Bruijn graph to `%s'\n", (dbg + CK_TMP).c_str());
	write_counting_file(ctx, dbg + CK_TMP, k, H);
	if (verbose) fprintf(stderr, "\tWriting visited k-mers Bloom to `%s'\n", (vis + CK_TMP).c_str());
	write_visited_file(ctx, vis + CK_TMP, k, H);
	if (verbose) fprintf(stderr, "\tWriting assembly counters to `%s'\n", (cnt + CK_TMP).c_str());
	abg_counters c;
	abg_get_counters(ctx, &c);
	FILE* f = fopen((cnt + CK_TMP).c_str(), "w");
	if (!f) { fprintf(stderr, "error: `%s': %s\n", (cnt + CK_TMP).c_str(), strerror(errno)); exit(EXIT_FAILURE); }
	fprintf(f, "solid_reads\tprocessed_reads\tbases_assembled\tnext_contig_id\n%llu\t%llu\t%llu\t%llu\n", // AssemblyCounters.h:32-52
	    (unsigned long long)c.solid_reads, (unsigned long long)c.reads_processed, (unsigned long long)c.bases_assembled,
	    (unsigned long long)c.next_contig_id);
	fclose(f);
	if (verbose) fprintf(stderr, "\tCopying `%s to `%s'\n", (fa + CK_TMP).c_str(), fa.c_str());
	copy_file(fa + CK_TMP, fa);
	do_rename(dbg + CK_TMP, dbg, verbose);
	do_rename(vis + CK_TMP, vis, verbose);
	do_rename(cnt + CK_TMP, cnt, verbose);
}
static void remove_checkpoint(const std::string& prefix, int verbose) // removeCheckpointData, Checkpoint.h:245-281
{
	if (verbose) fprintf(stderr, "Removing checkpoint files...\n");
	for (const char* ext : { CK_DBG, CK_VISITED, CK_COUNTERS, CK_FASTA })
		for (const char* tmp : { "", CK_TMP }) {
			std::string path = prefix + ext + tmp;
			if (file_readable(path) && remove(path.c_str()) != 0) { perror("Error removing file"); abort(); }
		}
}

int main(int argc, char** argv)
{
	abg_params p;
	abg_params_init(&p);
	std::string bloomPath, outputPath, tracePath, readLogPath, covTrackPath, refPath, graphPath;
	unsigned gpus = 1;
	unsigned threads = 0; // -j: host threads parsing FASTQ (the GPU does the assembly); 0 = as many as the machine has, up to 16
	int verbose = 0;
	bool die = false;
	unsigned K = 0, qr = 0;
	uint64_t readsPerCheckpoint = 0; bool keepCheckpoint = false; std::string checkpointPrefix = "bloom-dbg-checkpoint";
	std::string spaced;
	for (int c; (c = getopt_long(argc, argv, shortopts, longopts, NULL)) != -1;) {
		bool bad = false;
		char* end = NULL;
		switch (c) {
		case '?': die = true; break;
		case 'b': bad = !si_to_bytes(optarg, &p.bloom_bytes); break;
		case 'H': p.num_hashes = (uint32_t)strtoul(optarg, &end, 10); bad = *end; break;
		case 'i': bloomPath = optarg; break;
		case 'j': threads = (unsigned)strtoul(optarg, &end, 10); bad = *end; break;
		case 'k': p.k = (uint32_t)strtoul(optarg, &end, 10); bad = *end; break;
		case 'K': qr = 0; spaced.clear(); K = (unsigned)strtoul(optarg, &end, 10); bad = *end; break; // resetSpacedSeedParams, AssemblyParams.h:96-100
		case 'o': outputPath = optarg; break;
		case 'q': ropt.qualityThreshold = (int)strtol(optarg, &end, 10); bad = *end; break;
		case 'Q': ropt.internalQThreshold = (int)strtol(optarg, &end, 10); bad = *end; break;
		case 's': K = 0; qr = 0; spaced = optarg; break;
		case 't': p.trim = (uint32_t)strtoul(optarg, &end, 10); bad = *end; break;
		case 'T': tracePath = optarg; break;
		case 'v': ++verbose; break;
		case OPT_HELP: fputs(USAGE_MESSAGE, stdout); exit(EXIT_SUCCESS);
		case OPT_VERSION: fputs(VERSION_MESSAGE, stdout); exit(EXIT_SUCCESS);
		case MIN_KMER_COV: p.min_cov = (uint32_t)strtoul(optarg, &end, 10); bad = *end; break;
		case QR_SEED: K = 0; spaced.clear(); qr = (unsigned)strtoul(optarg, &end, 10); bad = *end; break;
		case READ_LOG: readLogPath = optarg; break;
		case 'C': covTrackPath = optarg; break;
		case 'R': refPath = optarg; break;
		case CHECKPOINT: readsPerCheckpoint = strtoull(optarg, &end, 10); bad = *end; break;
		case KEEP_CHECKPOINT: keepCheckpoint = true; break;
		case CHECKPOINT_PREFIX: checkpointPrefix = optarg; break;
		case 'g': graphPath = optarg; break;
		case OPT_GPUS: gpus = (unsigned)strtoul(optarg, &end, 10); bad = *end || gpus < 1 || gpus > ABG_MAX_RANKS; break;
		}
		if (bad) { // bloom-dbg.cc:472-475
			fprintf(stderr, PROGRAM ": invalid option: `-%c%s'\n", (char)c, optarg);
			exit(EXIT_FAILURE);
		}
	}
	if (bloomPath.empty() && p.bloom_bytes == 0) { fprintf(stderr, PROGRAM ": missing mandatory option `-b'\n"); die = true; }
	if (bloomPath.empty() && p.k == 0) { fprintf(stderr, PROGRAM ": missing mandatory option `-k'\n"); die = true; }
	if (p.k > 0 && K > 0 && K > p.k / 2) { fprintf(stderr, PROGRAM ": value of `-K' must be <= k/2\n"); die = true; }
	if (p.k > 0 && qr > 0 && (qr < 11 || qr > p.k / 2)) { fprintf(stderr, PROGRAM ": value of `--qr-seed' must be >= 11 and <= k/2\n"); die = true; }
	if (!covTrackPath.empty() && refPath.empty()) { fprintf(stderr, PROGRAM ": you must specify a reference with `-R' when using `-C'\n"); die = true; } // bloom-dbg.cc:512-516
	if (p.num_hashes > ABG_MAX_HASHES) { fprintf(stderr, PROGRAM ": number of hash functions (`-H`) must be <= %d\n", ABG_MAX_HASHES); die = true; }
	if (argc - optind < 1) { fprintf(stderr, PROGRAM ": missing input file arguments\n"); die = true; }
	if (die) { fprintf(stderr, "Try `%s --help' for more information.\n", PROGRAM); exit(EXIT_FAILURE); }
	if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	// (parser threads: more than 32 make the run SLOWER -- three runs each on a 256-thread host, the full configs[1] files: 1,176 ms at 32,
	// 1,191 at 24, 1,221 at 16, 1,277 at 12, 1,414 at 8, ~1,300 at 64 and more: the threads that drive the device wait for a core)
	threads = std::min(threads, 32u);

	// initGlobals (bloom-dbg.cc:214-233) + SpacedSeed.h:18-75
	std::string mask;
	if (K > 0) mask = spaced_seed_kmer_pair(p.k, K);
	else if (qr > 0) mask = spaced_seed_qr_pair(p.k, qr);
	else mask = spaced;
	if (!mask.empty()) {
		// MaskedKmer::setMask, MaskedKmer.h:25-48
		if (mask.size() != p.k) { fprintf(stderr, "error: spaced seed must be exactly k bits long\n"); exit(EXIT_FAILURE); }
		if (mask.find_first_not_of("01") != std::string::npos) { fprintf(stderr, "error: spaced seed must contain only '0's or '1's\n"); exit(EXIT_FAILURE); }
		if (mask.front() != '1' || mask.back() != '1') { fprintf(stderr, "error: spaced seed must begin and end with '1's\n"); exit(EXIT_FAILURE); }
		p.spaced_seed = mask.c_str();
	}

	// -i: [BTLCountingBloomFilter_v1] header + raw counters (CountingBloomFilter.hpp:262-329,344-379)
	std::vector<uint8_t> prebuilt;
	if (!bloomPath.empty()) {
		uint64_t size = 0; unsigned hn = 0, ks = 0;
		read_bloom_file(bloomPath, "[BTLCountingBloomFilter_v1]", size, hn, ks, prebuilt, false);
		p.k = ks; p.num_hashes = hn; p.counters = size; // bloom-dbg.cc:320-322
	}
	// --gpus N [extension]: one process per GPU.  The ranks are forked BEFORE anything touches the HIP
	// runtime; rank 0 makes the RCCL id and pipes it to the others; every rank then reads the same
	// input and makes the same library calls (the library partitions the filter and splits the
	// device work, include/abyss_amd.h); only rank 0 writes output.
	unsigned rank = 0;
	std::vector<pid_t> children;
	abg_comm comm;
	memset(&comm, 0, sizeof comm);
	const bool use_comm = gpus > 1 || (getenv("ABG_FORCE_DIST") && atoi(getenv("ABG_FORCE_DIST")));
	if (gpus > 1 && readsPerCheckpoint) { fprintf(stderr, PROGRAM ": --checkpoint is not available with --gpus\n"); exit(EXIT_FAILURE); }
	if (gpus > 1) // every rank reads every input: one shared stdin would be split between them
		for (int i = optind; i < argc; i++)
			if (!strcmp(argv[i], "-")) { fprintf(stderr, PROGRAM ": standard input (`-') cannot be read with --gpus\n"); exit(EXIT_FAILURE); }
	if (use_comm) {
		std::vector<int> rd(gpus, -1), wr(gpus, -1);
		for (unsigned r = 1; r < gpus; r++) {
			int fd[2];
			if (pipe(fd)) { perror("pipe"); exit(EXIT_FAILURE); }
			rd[r] = fd[0]; wr[r] = fd[1];
		}
		fflush(NULL);
		if (gpus > 1) {
			struct sigaction sa;
			memset(&sa, 0, sizeof sa);
			sa.sa_handler = on_sigchld;
			sa.sa_flags = SA_RESTART;
			sigaction(SIGCHLD, &sa, NULL);
		}
		for (unsigned r = 1; r < gpus; r++) {
			sigset_t block, old;
			sigemptyset(&block); sigaddset(&block, SIGCHLD);
			sigprocmask(SIG_BLOCK, &block, &old); // the child is on the list before its exit can be seen
			pid_t pid = fork();
			if (pid < 0) { perror("fork"); exit(EXIT_FAILURE); }
			if (pid == 0) { rank = r; children.clear(); g_nchildren = 0; signal(SIGCHLD, SIG_DFL); sigprocmask(SIG_SETMASK, &old, NULL); break; }
			children.push_back(pid);
			g_children[g_nchildren] = pid; g_nchildren = g_nchildren + 1; g_children_left = g_children_left + 1;
			sigprocmask(SIG_SETMASK, &old, NULL);
		}
		uint8_t ident[128];
		if (rank == 0) {
			if (abg_rccl_unique_id(ident) != ABG_OK) { fprintf(stderr, PROGRAM ": %s\n", abg_last_error(NULL)); exit(EXIT_FAILURE); }
			for (unsigned r = 1; r < gpus; r++)
				if (write(wr[r], ident, sizeof ident) != (ssize_t)sizeof ident) { perror("write"); exit(EXIT_FAILURE); }
		} else {
			// (only rank 0 holds the writing ends: if it dies before sending, the read below sees end of file)
			for (unsigned r = 1; r < gpus; r++) { close(wr[r]); wr[r] = -1; }
			if (read(rd[rank], ident, sizeof ident) != (ssize_t)sizeof ident) { fprintf(stderr, PROGRAM ": rank %u did not receive the communicator id\n", rank); exit(EXIT_FAILURE); }
			// the other ranks compute along and stay silent
			if (!getenv("ABG_RANK_STDERR")) { if (!freopen("/dev/null", "w", stderr)) exit(EXIT_FAILURE); }
			if (!outputPath.empty()) outputPath = "/dev/null";
			if (!freopen("/dev/null", "w", stdout)) exit(EXIT_FAILURE);
			for (std::string* path : { &tracePath, &readLogPath, &covTrackPath, &graphPath }) if (!path->empty()) *path = "/dev/null";
		}
		for (unsigned r = 1; r < gpus; r++) { close(rd[r]); if (wr[r] >= 0) close(wr[r]); }
		if (abg_rccl_comm_create(ident, (int32_t)rank, (int32_t)gpus, (int32_t)rank, &comm) != ABG_OK) { fprintf(stderr, PROGRAM ": %s\n", abg_last_error(NULL)); exit(EXIT_FAILURE); }
		p.device = (int32_t)rank;
	}
	// One GPU: the work is done in a child process.  The parent waits for one byte -- the exit status, which the child writes once every
	// output is flushed and closed -- and leaves with it, so that the caller (abyss-pe's next rule) goes on while the kernel takes the
	// worker's ~25 GB of device mappings down, a quarter of a second that neither `_exit` nor freeing the memory by hand avoids.  A child
	// that ends without sending the byte (any failure: they all leave through exit()) is waited for and its status passed on.
	// ABG_FOREGROUND=1 keeps everything in this process; so does --gpus, whose ranks are processes of their own already.  (A pipeline
	// that starts another GPU stage right behind this one at a large B should set ABG_FOREGROUND: until the worker is gone its device
	// memory is not free for the next stage.)
	int done_fd = -1;
	if (!use_comm && !getenv("ABG_FOREGROUND")) {
		int fd[2];
		// (close-on-exec: the decompressors the worker starts later must not hold the writing end open -- the parent's read below
		// would outlast a worker that died)
		if (pipe2(fd, O_CLOEXEC) == 0) {
			fflush(NULL);
			const pid_t parent = getpid();
			const pid_t pid = fork();
			if (pid > 0) {
				close(fd[1]);
				unsigned char st = 0;
				ssize_t got;
				while ((got = read(fd[0], &st, 1)) < 0 && errno == EINTR) {}
				if (got == 1) _exit(st);
				int ws = 0;
				while (waitpid(pid, &ws, 0) < 0 && errno == EINTR) {}
				_exit(WIFEXITED(ws) ? WEXITSTATUS(ws) : 128 + (WIFSIGNALED(ws) ? WTERMSIG(ws) : 1));
			} else if (pid == 0) {
				close(fd[0]);
				done_fd = fd[1];
				prctl(PR_SET_PDEATHSIG, SIGTERM); // (a parent that is killed takes the worker with it)
				if (getppid() != parent) _exit(EXIT_FAILURE); // (... also one killed between the fork and the line above)
			} else { close(fd[0]); close(fd[1]); } // (no child: carry on here)
		}
	}
	p.verbose = verbose;
	// The first window of the first input is read and parsed while the context comes up (HIP start-up and the
	// filters' memory: 0.1-0.2 s).  Plain files only: a compressed one waits for the prefetch below.
	std::unique_ptr<abghost::SequenceReader> primed;
	int primed_arg = -1;
	if (bloomPath.empty() && readsPerCheckpoint == 0 && optind < argc && strcmp(argv[optind], ":") && strcmp(argv[optind], "-")) {
		const char* prog; const char* flag;
		struct stat st;
		if (!abghost::Prefetch::compressed(argv[optind], &prog, &flag) && stat(argv[optind], &st) == 0 && S_ISREG(st.st_mode)) {
			primed.reset(new abghost::SequenceReader(argv[optind], ropt, threads));
			primed->prime();
			primed_arg = optind;
		}
	}
	abg_ctx* ctx = NULL;
	if (abg_create(&p, &ctx) != ABG_OK) { fprintf(stderr, PROGRAM ": %s\n", abg_last_error(NULL)); exit(EXIT_FAILURE); }
	host_mark("context created");
	if (use_comm) check(abg_attach_comm(ctx, &comm), ctx, "communicator");
	const uint32_t trim = p.trim == 0xFFFFFFFFu ? p.k : p.trim;
	if (verbose) {
		fprintf(stderr, "Assembling with k-mer size %u\n", p.k);
		if (!mask.empty()) fprintf(stderr, "Using spaced seed %s\n", mask.c_str());
		fprintf(stderr, "Assembly parameters:\n\tK-mer size (-k): %u\n\tK-mer coverage threshold (--kc): %u\n"
		    "\tMax branch trim length (-t): %u\n\tBloom size in bytes (-b): %llu\n\tBloom hash functions (-H): %u\n",
		    p.k, p.min_cov, trim, (unsigned long long)p.bloom_bytes, p.num_hashes);
	}
	FILE* out = stdout;
	if (!outputPath.empty() && !(out = fopen(outputPath.c_str(), "w"))) { fprintf(stderr, "error: `%s': %s\n", outputPath.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
	size_t CHUNK_BASES = 256u << 20;
	if (const char* e = getenv("ABG_CHUNK_MB")) CHUNK_BASES = (size_t)std::max(1, atoi(e)) << 20; // (measurements)
	int first_asm = optind;
	std::string id, comment, seq;
	Chunk chunk;
	uint64_t counters = 0;
	abg_filter_size(ctx, &counters);
	// The reference reads its input twice (loadBloomFilter, then assemble).  When both passes run
	// over the same files, the parsed records of pass 1 are kept for pass 2 as long as they fit in
	// a quarter of the machine's memory: parsing is what the host spends its time on.
	std::vector<Chunk> kept;
	bool keep = prebuilt.empty();
	for (int i = optind; i < argc; ++i) if (!strcmp(argv[i], ":") || !strcmp(argv[i], "-")) keep = false;
	// ... and when nothing needs the bases on the host afterwards (the -T trace prints a read's seed k-mer),
	// they are kept where PASS 1 put them instead: packed, on the device (abg_keep_reads), and the parser's
	// blocks go to the library as they lie (abg_load_seqs_v) -- `kept` then only holds the ids.
	bool packed_keep = keep && tracePath.empty() && readsPerCheckpoint == 0 && !getenv("ABG_NO_PACKED_KEEP");
	if (packed_keep) {
		uint64_t expect = 0;
		for (int i = optind; i < argc; ++i) {
			struct stat st;
			const char* prog; const char* flag;
			if (stat(argv[i], &st) == 0 && S_ISREG(st.st_mode)) expect += (uint64_t)st.st_size * (abghost::Prefetch::compressed(argv[i], &prog, &flag) ? 3 : 1);
		}
		if (abg_keep_reads(ctx, 1, expect / 2) != ABG_OK) packed_keep = false; // (more than the device should hold: the host keeps them)
	}
	const bool packed_path = packed_keep; // (how PASS 1's chunks are fed; packed_keep may be given up on the way)
	std::vector<Chunk> cur_v, loading_v; // packed_path: the chunk being filled / loaded, as the parser's blocks
	size_t cur_bases = 0;
	size_t kept_bytes = 0;
	size_t keep_limit = (size_t)sysconf(_SC_PHYS_PAGES) / 4 * (size_t)sysconf(_SC_PAGE_SIZE) / gpus; // (every rank keeps its own copy)
	if (const char* e = getenv("ABG_KEEP_LIMIT_BYTES")) keep_limit = (size_t)strtoull(e, NULL, 10); // (tests: the host gives up keeping)
	// PASS 1 of a chunk runs on a thread of its own while the reader parses the next chunk (one chunk
	// in flight: the chunks go in in order)
	std::thread loader;
	Chunk loading;
	int load_rc = ABG_OK; // what the loader thread's call returned: looked at on THIS thread once it is joined
	auto load_done = [&]() {
		if (!loader.joinable()) return;
		loader.join();
		check(load_rc, ctx, "load"); // (exit() from the loader thread would run the static destructors beside live threads)
		host_mark("chunk loaded");
		if (packed_path) {
			// (the bases stay on the device; the host keeps the ids -- and gives the whole arrangement up when even those outgrow its share
			// of the machine's memory: the device store may hold a billion reads, every rank of a --gpus run keeps its own copy of the ids)
			for (Chunk& c : loading_v) {
				if (!packed_keep) continue;
				c.drop_seqs();
				kept_bytes += c.idbuf.size() + 8 * c.n();
				kept.push_back(std::move(c));
			}
			loading_v.clear();
			if (packed_keep && kept_bytes > keep_limit) {
				check(abg_keep_reads(ctx, 0, 0), ctx, "keep"); // (drops the device store; PASS 2 reads the input again, as the reference does)
				packed_keep = false; keep = false;
				kept.clear(); kept.shrink_to_fit();
			}
			return;
		}
		if (keep) {
			kept_bytes += loading.bytes();
			if (kept_bytes > keep_limit) { keep = false; kept.clear(); kept.shrink_to_fit(); }
			else { kept.push_back(std::move(loading)); loading = Chunk(); return; }
		}
		loading = Chunk();
	};
	auto loaded = [&]() {
		load_done();
		loading = std::move(chunk);
		chunk = Chunk();
		chunk.seqs.reserve(loading.seqs.size() + (64u << 20)); // (the next one grows to about the same size: no copies on the way)
		chunk.off.reserve(loading.off.size() + 1024); chunk.id_end.reserve(loading.id_end.size() + 1024); chunk.idbuf.reserve(loading.idbuf.size() + (1u << 20));
		loader = std::thread([&]() {
			const double tl = host_now();
			load_rc = abg_load_seqs(ctx, loading.seqs.data(), loading.off.data(), loading.n());
			g_in_load += host_now() - tl;
		});
	};
	// (the first chunks of a kept-reads run are small, so that the device starts while the parser has
	// read a fraction of a second's worth; then they double up to CHUNK_BASES)
	size_t chunk_target = 32u << 20;
	if (const char* e = getenv("ABG_FIRST_CHUNK_MB")) chunk_target = (size_t)std::max(1, atoi(e)) << 20;
	auto loaded_v = [&]() {
		load_done();
		loading_v = std::move(cur_v);
		cur_v.clear(); cur_bases = 0;
		chunk_target = std::min(CHUNK_BASES, chunk_target * 2);
		loader = std::thread([&]() {
			std::vector<const char*> sv; std::vector<const uint64_t*> ov; std::vector<uint64_t> nv;
			for (const Chunk& c : loading_v) { sv.push_back(c.seqs.data()); ov.push_back(c.off.data()); nv.push_back(c.n()); }
			const double tl = host_now();
			load_rc = abg_load_seqs_v(ctx, (uint32_t)sv.size(), sv.data(), ov.data(), nv.data());
			g_in_load += host_now() - tl;
		});
	};
	// --checkpoint=N (bloom-dbg.h:1012-1077, Checkpoint.h): the state is saved every N reads, and a
	// run that finds a complete set of checkpoint files picks up from it (bloom-dbg.cc:546-547).
	// (Resuming restores what the files hold: both filters, the four counters, the contigs so far;
	// like the reference it starts with an empty contigEndKmers set.)
	const bool ckpt = readsPerCheckpoint > 0;
	const bool resume = ckpt && checkpoint_exists(checkpointPrefix);
	uint64_t skip_reads = 0;
	if (ckpt) keep = false;
	if (resume) {
		if (verbose) fprintf(stderr, "Resuming from last checkpoint...\n");
		std::vector<uint8_t> payload; uint64_t size = 0; unsigned hn = 0, ks = 0;
		if (verbose) fprintf(stderr, "\tReading Bloom filter de Bruijn graph from `%s'\n", (checkpointPrefix + CK_DBG).c_str());
		read_bloom_file(checkpointPrefix + CK_DBG, "[BTLCountingBloomFilter_v1]", size, hn, ks, payload, false);
		if (size != counters || hn != p.num_hashes || ks != p.k) { fprintf(stderr, PROGRAM ": checkpoint does not match -k/-b/-H\n"); exit(EXIT_FAILURE); }
		check(abg_counters_import(ctx, payload.data()), ctx, "import");
		if (verbose) fprintf(stderr, "\tReading reading visited k-mers Bloom from `%s'\n", (checkpointPrefix + CK_VISITED).c_str());
		read_bloom_file(checkpointPrefix + CK_VISITED, "[BTLBloomFilter_v1]", size, hn, ks, payload, true);
		if (size != counters) { fprintf(stderr, PROGRAM ": checkpoint does not match -k/-b/-H\n"); exit(EXIT_FAILURE); }
		check(abg_visited_import(ctx, payload.data()), ctx, "import");
		if (verbose) fprintf(stderr, "\tReading index of next input read from `%s'\n", (checkpointPrefix + CK_COUNTERS).c_str());
		abg_counters c; memset(&c, 0, sizeof c);
		{
			FILE* f = fopen((checkpointPrefix + CK_COUNTERS).c_str(), "r");
			unsigned long long a = 0, b = 0, d = 0, e = 0;
			if (!f || fscanf(f, "%*[^\n]\n%llu\t%llu\t%llu\t%llu", &a, &b, &d, &e) != 4) { fprintf(stderr, "error: `%s': malformed\n", (checkpointPrefix + CK_COUNTERS).c_str()); exit(EXIT_FAILURE); }
			fclose(f);
			c.solid_reads = a; c.reads_processed = b; c.bases_assembled = d; c.next_contig_id = e;
		}
		check(abg_set_counters(ctx, &c), ctx, "counters");
		skip_reads = c.reads_processed;
		if (verbose) fprintf(stderr, "\tAdvancing to read index %llu in input reads...\n", (unsigned long long)skip_reads);
		if (verbose) fprintf(stderr, "\tCopying `%s' to `%s'\n", (checkpointPrefix + CK_FASTA).c_str(), (checkpointPrefix + CK_FASTA + CK_TMP).c_str());
		copy_file(checkpointPrefix + CK_FASTA, checkpointPrefix + CK_FASTA + CK_TMP);
		if (verbose) fprintf(stderr, "\tOutputting previously assembled contigs from `%s'\n", (checkpointPrefix + CK_FASTA).c_str());
		FILE* prev = fopen((checkpointPrefix + CK_FASTA).c_str(), "rb");
		char buf[1 << 16]; size_t nb;
		while (prev && (nb = fread(buf, 1, sizeof buf, prev)) > 0) fwrite(buf, 1, nb, out);
		if (prev) fclose(prev);
	} else if (prebuilt.empty()) {
		// PASS 1: loadBloomFilter, BloomIO.h:97-118 (a ":" argument separates load and assembly files)
		if (threads > 1 && !ckpt) { // compressed inputs inflate side by side, ahead of the reader (Prefetch)
			std::vector<std::string> ins;
			for (int i = optind; i < argc && strcmp(argv[i], ":"); ++i) ins.push_back(argv[i]);
			abghost::Prefetch::get().start(ins, gpus);
		}
		chunk.seqs.reserve(CHUNK_BASES + (64u << 20));
		for (int i = optind; i < argc; ++i) {
			if (!strcmp(argv[i], ":")) { first_asm = i + 1; break; }
			if (verbose) fprintf(stderr, "Reading `%s'...\n", argv[i]);
			std::unique_ptr<abghost::SequenceReader> own;
			if (i == primed_arg && primed) own = std::move(primed); else own.reset(new abghost::SequenceReader(argv[i], ropt, threads));
			abghost::SequenceReader& in = *own;
			uint64_t n = 0;
			if (packed_path) {
				if (in.has_blocks()) {
					abghost::SequenceReader::Block blk;
					while (in.next_block(blk)) {
						n += blk.seq_end.size();
						cur_v.emplace_back();
						cur_v.back().take_block(blk);
						cur_bases += cur_v.back().seqs.size();
						if (cur_bases >= chunk_target) loaded_v();
					}
				} else {
					while (in.read(id, comment, seq)) {
						if (cur_v.empty() || cur_v.back().seqs.size() >= (64u << 20)) cur_v.emplace_back();
						cur_v.back().add(id, seq); n++;
						cur_bases += seq.size();
						if (cur_bases >= chunk_target) loaded_v();
					}
				}
				if (!cur_v.empty()) loaded_v();
			} else if (in.has_blocks()) { // the parser threads' records wholesale
				abghost::SequenceReader::Block blk;
				while (in.next_block(blk)) {
					chunk.add_block(blk); n += blk.seq_end.size();
					if (chunk.seqs.size() >= CHUNK_BASES) loaded();
				}
			} else {
				while (in.read(id, comment, seq)) {
					chunk.add(id, seq); n++;
					if (chunk.seqs.size() >= CHUNK_BASES) loaded();
				}
			}
			if (chunk.n()) loaded();
			if (verbose) fprintf(stderr, "Loaded %llu reads from `%s` into Bloom filter\n", (unsigned long long)n, argv[i]);
		}
		load_done();
		// (while reads are kept, the device's share of the LAST load call still runs on the library's thread and its
		// failure would come back from whichever call is next: this one waits for it and is checked)
		{ uint64_t m = 0; check(abg_filter_size(ctx, &m), ctx, "load"); }
	} else {
		if (prebuilt.size() != counters) { fprintf(stderr, PROGRAM ": Bloom file size does not match its header\n"); exit(EXIT_FAILURE); }
		check(abg_counters_import(ctx, prebuilt.data()), ctx, "import");
	}
	if (verbose) {
		uint64_t pop = 0, filt = 0;
		check(abg_counting_stats(ctx, &pop, &filt), ctx, "counting filter statistics");
		fprintf(stderr, "Bloom filter FPR: %.3g%%\n", 100.0 * pow((double)pop / (double)counters, p.num_hashes));
		fprintf(stderr, "Counting Bloom filter stats:\n\t#counters               = %llu\n\t#size (B)               = %llu\n"
		    "\tthreshold               = %u\n\tpopcount                = %llu\n\tFPR                     = %.3g%%\n",
		    (unsigned long long)counters, (unsigned long long)counters, p.min_cov, (unsigned long long)filt,
		    100.0 * pow((double)filt / (double)counters, p.num_hashes));
		fprintf(stderr, "Trimming branches %u k-mers or shorter\n", trim);
	}
	// -C / -R: writeCovTrack (bloom-dbg.cc:200-201, bloom-dbg.h:1251-1334): a variableStep WIG with 1
	// where the reference's k-mer is in the solid filter, 0 where it is not
	if (!covTrackPath.empty() && !refPath.empty()) {
		FILE* wig = fopen(covTrackPath.c_str(), "w");
		if (!wig) { fprintf(stderr, "error: `%s': %s\n", covTrackPath.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		if (verbose) fprintf(stderr, "Writing 0/1 k-mer coverage track for `%s` to `%s`\n", refPath.c_str(), covTrackPath.c_str());
		abghost::FastaReader ref(refPath, ropt);
		std::vector<uint32_t> pos; std::vector<uint8_t> val;
		while (ref.read(id, comment, seq)) {
			pos.resize(seq.size() + 1); val.resize(seq.size() + 1);
			uint64_t n = 0;
			check(abg_contains_seq(ctx, seq.data(), seq.size(), pos.data(), val.data(), pos.size(), &n), ctx, "contains");
			size_t blockStart = 1, blockLength = 0; unsigned blockVal = 0;
			for (uint64_t i = 0; i < n; i++) {
				if (i == 0 || val[i] != blockVal) {
					if (i) fprintf(wig, "variableStep chrom=%s span=%zu\n%zu %u\n", id.c_str(), blockLength, blockStart, blockVal);
					blockStart = (size_t)pos[i] + 1; blockLength = 1; blockVal = val[i]; // WIG coordinates are 1-based
				} else {
					blockLength++;
				}
			}
			if (blockLength > 0) fprintf(wig, "variableStep chrom=%s span=%zu\n%zu %u\n", id.c_str(), blockLength, blockStart, blockVal);
		}
		fclose(wig);
	}
	// -g: outputGraph (bloom-dbg.cc:203-211, bloom-dbg.h:1171-1242) over the assembly input files.  (The
	// reference writes it after the assembly; it reads nothing but the solid filter, so the order is free.)
	if (!graphPath.empty()) {
		FILE* gv = fopen(graphPath.c_str(), "w");
		if (!gv) { fprintf(stderr, "error: `%s': %s\n", graphPath.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		if (verbose) fprintf(stderr, "Generating GraphViz output...\n");
		fputs("digraph g {\n", gv);
		uint64_t nodes = 0, edges = 0, nreads = 0;
		Chunk gc;
		auto graph = [&]() {
			if (!gc.n()) return;
			uint64_t a = 0, b = 0;
			check(abg_output_graph_seqs(ctx, gc.seqs.data(), gc.off.data(), gc.n(),
			    [](void* u, const char* text, uint64_t len) { fwrite(text, 1, len, (FILE*)u); }, gv, &a, &b), ctx, "graph");
			nodes += a; edges += b;
			gc.clear();
		};
		for (int i = first_asm; i < argc; ++i) {
			if (!strcmp(argv[i], ":")) continue;
			abghost::SequenceReader in(argv[i], ropt, threads);
			while (in.read(id, comment, seq)) {
				gc.add(id, seq); nreads++;
				if (gc.seqs.size() >= CHUNK_BASES) graph();
			}
		}
		graph();
		fputs("}\n", gv);
		if (fclose(gv)) { fprintf(stderr, "error: `%s': %s\n", graphPath.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		if (verbose) fprintf(stderr, "processed %llu reads (k-mers visited: %llu, edges visited: %llu)\nGraphViz generation complete\n",
		    (unsigned long long)nreads, (unsigned long long)nodes, (unsigned long long)edges);
	}
	// PASS 2: assemble, bloom-dbg.h:900-951,972-1089
	FILE* trace = NULL; FILE* readlog = NULL;
	if (!tracePath.empty()) {
		if (!(trace = fopen(tracePath.c_str(), "w"))) { fprintf(stderr, "error: `%s': %s\n", tracePath.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		fputs("contig_id\tlength\tredundant\tread_id\tleft_result\tleft_extension\tright_result\tright_extension\tseed_type\tseed_length\tseed\n", trace);
	}
	if (!readLogPath.empty()) {
		if (!(readlog = fopen(readLogPath.c_str(), "w"))) { fprintf(stderr, "error: `%s': %s\n", readLogPath.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		fputs("read_id\tresult\n", readlog);
	}
	static const char* rr[] = { "NA", "SHORTER_THAN_K", "NON_ACGT", "BLUNT_END", "NOT_SOLID", "ALL_KMERS_VISITED", "ALL_BRANCH_KMERS_VISITED", "GENERATED_CONTIGS" };
	Output o{ out, trace, &chunk, p.k };
	if (ckpt) {
		const std::string path = checkpointPrefix + CK_FASTA + CK_TMP;
		if (!(o.checkpoint = fopen(path.c_str(), resume ? "a" : "w"))) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
	}
	uint64_t untilCheckpoint = readsPerCheckpoint;
	std::vector<uint8_t> results;
	auto assemble = [&](Chunk& c) {
		if (!c.n()) return;
		o.chunk = &c;
		results.assign(c.n(), 0);
		const double ta = host_now();
		check(abg_assemble_seqs(ctx, c.seqs.data(), c.off.data(), c.n(), results.data(), on_contig, &o), ctx, "assemble");
		g_in_asm += host_now() - ta;
		host_mark("chunk assembled");
		if (readlog) for (size_t i = 0; i < c.n(); i++) fprintf(readlog, "%s\t%s\n", c.id(i).c_str(), rr[results[i]]);
	};
	auto flush = [&]() { assemble(chunk); chunk.clear(); };
	if (packed_keep) {
		// the reads PASS 1 left on the device, all of them in one pass
		o.chunks = &kept; o.first.assign(1, 0);
		for (const Chunk& c : kept) o.first.push_back(o.first.back() + c.n());
		results.assign(o.first.back(), 0);
		const double ta = host_now();
		const int rc = abg_assemble_kept(ctx, results.data(), on_contig, &o);
		g_in_asm += host_now() - ta;
		if (rc == ABG_EAGAIN) {
			// (the device had no room to keep them after all -- said before PASS 2 touched anything: the input is
			// read again, as the reference reads it; every other failure, ABG_ENOMEM inside PASS 2 included, is final)
			if (verbose) fprintf(stderr, "%s; reading the input again\n", abg_last_error(ctx));
			o.chunks = NULL; kept.clear(); keep = false; first_asm = optind;
		} else {
			check(rc, ctx, "assemble");
			host_mark("kept reads assembled");
			if (readlog)
				for (size_t q = 0; q < kept.size(); q++)
					for (size_t i = 0; i < kept[q].n(); i++) fprintf(readlog, "%s\t%s\n", kept[q].id(i).c_str(), rr[results[o.first[q] + i]]);
			o.chunks = NULL;
			first_asm = argc; // nothing left to read
		}
	} else if (keep && !kept.empty()) {
		// the records kept from PASS 1, all of them in one pass (one guide, one walk schedule)
		std::vector<const char*> sv; std::vector<const uint64_t*> ov; std::vector<uint64_t> nv;
		o.chunks = &kept; o.first.assign(1, 0);
		for (const Chunk& c : kept) { sv.push_back(c.seqs.data()); ov.push_back(c.off.data()); nv.push_back(c.n()); o.first.push_back(o.first.back() + c.n()); }
		results.assign(o.first.back(), 0);
		const double ta = host_now();
		check(abg_assemble_seqs_v(ctx, (uint32_t)kept.size(), sv.data(), ov.data(), nv.data(), results.data(), on_contig, &o), ctx, "assemble");
		g_in_asm += host_now() - ta;
		host_mark("kept chunks assembled");
		if (readlog)
			for (size_t q = 0; q < kept.size(); q++)
				for (size_t i = 0; i < kept[q].n(); i++) fprintf(readlog, "%s\t%s\n", kept[q].id(i).c_str(), rr[results[o.first[q] + i]]);
		o.chunks = NULL;
		kept.clear();
		first_asm = argc; // nothing left to read
	}
	for (int i = first_asm; i < argc; ++i) {
		if (!strcmp(argv[i], ":")) continue;
		abghost::SequenceReader in(argv[i], ropt, threads);
		while (in.read(id, comment, seq)) {
			if (skip_reads) { skip_reads--; continue; }
			chunk.add(id, seq);
			if (ckpt && --untilCheckpoint == 0) {
				flush();
				fflush(o.checkpoint);
				create_checkpoint(ctx, checkpointPrefix, p.k, p.num_hashes, verbose);
				untilCheckpoint = readsPerCheckpoint;
			} else if (chunk.seqs.size() >= CHUNK_BASES) {
				flush();
			}
		}
	}
	flush();
	if (o.checkpoint) fclose(o.checkpoint);
	if (verbose) {
		abg_counters c;
		abg_get_counters(ctx, &c);
		fprintf(stderr, "Processed %llu reads, solid reads: %llu (%.3g%%), visited reads: %llu (%.3g%%)\n",
		    (unsigned long long)c.reads_processed, (unsigned long long)c.solid_reads, 100.0f * c.solid_reads / c.reads_processed,
		    (unsigned long long)c.visited_reads, 100.0f * c.visited_reads / c.reads_processed);
		fprintf(stderr, "Assembled %llu bp in %llu contigs\nAssembly complete\n", (unsigned long long)c.bases_assembled,
		    (unsigned long long)c.next_contig_id);
	}
	host_mark("assembly complete");
	if (getenv("ABG_PRINT_STATS")) { // engine work counters (tests assert which code paths a run took)
		abg_stats st;
		memset(&st, 0, sizeof st);
		abg_get_stats(ctx, &st);
		fprintf(stderr, "abyss_amd stats: insert_rounds=%llu walk_rounds=%llu candidates=%llu rewalked=%llu commit_rounds=%llu "
		    "batch_cuts=%llu overflows=%llu bulk_steps=%llu lin_steps=%llu chain_steps=%llu memo_hits=%llu memo_adds=%llu\n",
		    (unsigned long long)st.insert_rounds, (unsigned long long)st.walk_rounds, (unsigned long long)st.candidates,
		    (unsigned long long)st.rewalked, (unsigned long long)st.commit_rounds, (unsigned long long)st.batch_cuts,
		    (unsigned long long)st.overflows, (unsigned long long)st.bulk_steps, (unsigned long long)st.lin_steps,
		    (unsigned long long)st.chain_steps, (unsigned long long)st.memo_hits, (unsigned long long)st.memo_adds);
	}
	if (ckpt && !keepCheckpoint) remove_checkpoint(checkpointPrefix, verbose);
	if (trace) fclose(trace);
	if (readlog) fclose(readlog);
	if (out != stdout) fclose(out); else fflush(stdout);
	host_mark("output closed");
	if (!use_comm && g_children_left <= 0 && !getenv("ABG_ORDERLY_EXIT")) {
		// Everything this process was asked for is written and flushed.  Tearing the context down (tens of GB of device
		// allocations handed back one by one, then the HIP runtime's own static destructors) is a quarter of a second the
		// pipeline waits for nothing: the kernel reclaims a process's device memory faster than the process can.
		fflush(NULL);
		if (done_fd >= 0) {
			// (everything is written: the descriptors go first, so that a reader of our output sees its end, then the word to the parent)
			close(1); close(2);
			const unsigned char ok = EXIT_SUCCESS;
			if (write(done_fd, &ok, 1) != 1) _exit(EXIT_FAILURE);
		}
		_exit(EXIT_SUCCESS);
	}
	abg_destroy(ctx);
	host_mark("context destroyed");
	if (use_comm) abg_rccl_comm_destroy(&comm);
	// (the handler reaps the other ranks and ends the job if one of them fails)
	while (rank == 0 && g_children_left > 0) usleep(1000);
	return EXIT_SUCCESS;
}
