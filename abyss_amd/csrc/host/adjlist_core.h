// adjlist_core.h -- host side of the drop-in `AdjList` (the stage abyss-pe runs on the unitig
// FASTA right after abyss-bloom-dbg, bin/abyss-pe:575-577): options, contig reading, the
// suffix-array overlaps of fewer than k-1 bases, and the graph writers.  The join of contig ends
// that overlap by exactly k-1 bases -- the part that touches every contig -- is the caller's
// `Join` (abg_overlap_join on the GPU in the product binary; tests/hostcheck substitutes the same
// device logic run serially).
//
// Reference behaviour restated here (ABySS 2.3.10):
//   AdjList/AdjList.cpp:36-133,323-395  options, usage, -m handling
//   AdjList/AdjList.cpp:137-189         addOverlapsSA: overlaps of [m, k-1) bases among blunt vertices
//   AdjList/AdjList.cpp:192-230         readContigs (FOLD_CASE, flattenAmbiguityCodes, `LEN COV` comment)
//   Common/SuffixArray.h                suffixes of length >= m, sorted by (strcmp, vertex)
//   Common/Sequence.h:50-73             flattenAmbiguityCodes
//   Graph/AdjIO.h:32-66  DotIO.h:14-101  GfaIO.h:15-211  AsqgIO.h:13-70  SAMIO.h:18-70   writers
//   Graph/GraphUtil.h:27-64, Common/Histogram.cpp:45-95   -v statistics
#pragma once

#include "fasta_reader.h"
#include "graph_writers.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <getopt.h>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>
#include <cmath>

namespace abgadj {

#define ABG_ADJ_PROGRAM "AdjList"
#define ABG_ADJ_VERSION ABG_IO_VERSION

using abgio::ADJ; using abgio::ASQG; using abgio::DOT; using abgio::GFA1; using abgio::GFA2; using abgio::SAM; // Graph/Options.h (the ones AdjList offers)

struct Options {
	unsigned k = 0, singleKmer = 0, minOverlap = 50;
	int format = ADJ, ss = 0, verbose = 0;
	std::vector<std::string> files;
	std::string commandLine;
};

// k-1 join: (overlap, n, head keys, tail keys, ss) -> CSR (offsets [2n+1], targets); see include/abyss_amd.h
typedef std::function<void(uint32_t, uint64_t, const uint64_t*, const uint64_t*, bool, std::vector<uint64_t>&, std::vector<uint32_t>&)> Join;

struct Edge { uint32_t v; int d; };

struct Graph {
	unsigned k = 0;
	std::vector<std::string> name;               // per contig
	std::vector<unsigned> length, coverage;      // per contig
	std::vector<uint64_t> head, tail;            // per contig, W words each: first / last k-1 bases
	uint32_t W = 0;
	std::vector<uint64_t> off;                   // k-1 overlaps, CSR over the source vertex
	std::vector<uint32_t> tgt;
	std::vector<std::vector<Edge>> extra;        // shorter overlaps of blunt vertices (empty unless -m < k-1)
	uint64_t n() const { return name.size(); }
	uint64_t nv() const { return 2 * n(); }
	uint64_t degree(uint64_t u) const
	{
		const uint64_t d = off[u + 1] - off[u];
		return d || extra.empty() ? d : extra[u].size();
	}
	template <class F> void for_out(uint64_t u, F f) const
	{
		if (off[u + 1] > off[u] || extra.empty()) {
			for (uint64_t e = off[u]; e < off[u + 1]; e++) f(tgt[e], -(int)(k - 1));
		} else {
			for (const Edge& e : extra[u]) f(e.v, e.d);
		}
	}
	uint64_t edges() const
	{
		uint64_t e = tgt.size();
		for (auto& x : extra) e += x.size();
		return e;
	}
	// (what abgio's writers ask of a graph)
	bool removed(uint64_t) const { return false; }
	const std::string& cname(uint64_t u) const { return name[u >> 1]; }
	unsigned len(uint64_t u) const { return length[u >> 1]; }
	unsigned cov(uint64_t u) const { return coverage[u >> 1]; }
};

static const char USAGE_MESSAGE[] =
"Usage: " ABG_ADJ_PROGRAM " -k<kmer> [OPTION]... [FILE]...\n"
"Find overlaps of [m,k) bases. Contigs may be read from FILE(s)\n"
"or standard input. Output is written to standard output.\n"
"Overlaps of exactly k-1 bases are found by a sort-and-join on the GPU.\n"
"Overlaps of fewer than k-1 bases are found using a suffix array.\n"
"\n"
" Options:\n"
"\n"
"  -k, --kmer=N          the length of a k-mer\n"
"  -m, --min-overlap=M   require a minimum overlap of M bases [50]\n"
"                        value of 0 is interpreted as k - 1\n"
"      --adj             output the graph in ADJ format [default]\n"
"      --asqg            output the graph in ASQG format\n"
"      --dot             output the graph in GraphViz format\n"
"      --gfa             output the graph in GFA1 format\n"
"      --gfa1            output the graph in GFA1 format\n"
"      --gfa2            output the graph in GFA2 format\n"
"      --gv              output the graph in GraphViz format\n"
"      --sam             output the graph in SAM format\n"
"      --SS              expect contigs to be oriented correctly\n"
"      --no-SS           no assumption about contig orientation\n"
"  -v, --verbose         display verbose output\n"
"      --help            display this help and exit\n"
"      --version         output version information and exit\n"
"      --gpu=N           HIP device ordinal [0]\n"
"\n"
"-K (paired de Bruijn graph) and --db are not supported by this build.\n";

// AdjList.cpp:323-378.  Returns false when the caller should exit with `*status`.
inline bool parse_options(int argc, char** argv, Options& o, int* device, int* status)
{
	{
		std::ostringstream ss;
		for (int i = 0; i < argc; i++) ss << (i ? " " : "") << argv[i];
		o.commandLine = ss.str();
	}
	enum { OPT_HELP = 1, OPT_VERSION, OPT_DB, OPT_LIBRARY, OPT_STRAIN, OPT_SPECIES, OPT_GPU };
	static int format = ADJ, ss = 0;
	static const struct option longopts[] = {
		{ "kmer", required_argument, NULL, 'k' }, { "single-kmer", required_argument, NULL, 'K' },
		{ "min-overlap", required_argument, NULL, 'm' },
		{ "adj", no_argument, &format, ADJ }, { "asqg", no_argument, &format, ASQG }, { "dot", no_argument, &format, DOT },
		{ "gfa", no_argument, &format, GFA1 }, { "gfa1", no_argument, &format, GFA1 }, { "gfa2", no_argument, &format, GFA2 },
		{ "gv", no_argument, &format, DOT }, { "sam", no_argument, &format, SAM },
		{ "SS", no_argument, &ss, 1 }, { "no-SS", no_argument, &ss, 0 },
		{ "verbose", no_argument, NULL, 'v' }, { "help", no_argument, NULL, OPT_HELP }, { "version", no_argument, NULL, OPT_VERSION },
		{ "db", required_argument, NULL, OPT_DB }, { "library", required_argument, NULL, OPT_LIBRARY },
		{ "strain", required_argument, NULL, OPT_STRAIN }, { "species", required_argument, NULL, OPT_SPECIES },
		{ "gpu", required_argument, NULL, OPT_GPU },
		{ NULL, 0, NULL, 0 }
	};
	bool die = false;
	for (int c; (c = getopt_long(argc, argv, "k:K:m:v", longopts, NULL)) != -1;) {
		std::istringstream arg(optarg != NULL ? optarg : "");
		switch (c) {
		case '?': die = true; break;
		case 'k': arg >> o.k; break;
		case 'K': arg >> o.singleKmer; break;
		case 'm': arg >> o.minOverlap; break;
		case 'v': o.verbose++; break;
		case OPT_HELP: fputs(USAGE_MESSAGE, stdout); *status = EXIT_SUCCESS; return false;
		case OPT_VERSION: fputs(ABG_ADJ_PROGRAM " (ABySS, abyss_amd) " ABG_ADJ_VERSION "\n", stdout); *status = EXIT_SUCCESS; return false;
		case OPT_DB: case OPT_LIBRARY: case OPT_STRAIN: case OPT_SPECIES: {
			std::string s; arg >> s;
			fprintf(stderr, ABG_ADJ_PROGRAM ": warning: the database options are ignored (built without sqlite)\n");
			break;
		}
		case OPT_GPU: arg >> *device; break;
		}
		if (optarg != NULL && !arg.eof()) {
			fprintf(stderr, ABG_ADJ_PROGRAM ": invalid option: `-%c%s'\n", (char)c, optarg);
			*status = EXIT_FAILURE;
			return false;
		}
	}
	o.format = format;
	o.ss = ss;
	if (o.k <= 0) { fprintf(stderr, ABG_ADJ_PROGRAM ": missing -k,--kmer option\n"); die = true; }
	if (o.singleKmer > 0) { fprintf(stderr, ABG_ADJ_PROGRAM ": -K (paired de Bruijn graph) is not supported\n"); die = true; }
	if (!die && o.k < 2) { fprintf(stderr, ABG_ADJ_PROGRAM ": -k must be at least 2\n"); die = true; }
	if (!die && o.k - 1 > 256) { fprintf(stderr, ABG_ADJ_PROGRAM ": -k must be at most 257\n"); die = true; }
	if (die) {
		fprintf(stderr, "Try `" ABG_ADJ_PROGRAM " --help' for more information.\n");
		*status = EXIT_FAILURE;
		return false;
	}
	if (o.minOverlap == 0) o.minOverlap = o.k - 1;
	o.minOverlap = std::min(o.minOverlap, o.k - 1);
	for (; optind < argc; optind++) o.files.push_back(argv[optind]);
	if (o.files.empty()) o.files.push_back("-");
	return true;
}

inline int base_code(char c)
{
	switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}
// Kmer(seq.substr(at, len)): 2 bits per base, base j at bits 2(j%32) of word j/32 (baseToCode, Common/Sequence.cpp:96-105)
inline void pack_key(const std::string& s, size_t at, unsigned len, uint64_t* out, unsigned W)
{
	for (unsigned w = 0; w < W; w++) out[w] = 0;
	for (unsigned j = 0; j < len; j++) {
		const int c = base_code(s[at + j]);
		if (c < 0) {
			fprintf(stderr, "error: unexpected character: '%c'\n", s[at + j]);
			exit(EXIT_FAILURE);
		}
		out[j >> 5] |= (uint64_t)c << (2 * (j & 31));
	}
}
inline std::string key_string(const uint64_t* w, unsigned len, bool rc)
{
	std::string s(len, 'A');
	for (unsigned j = 0; j < len; j++) {
		const unsigned c = (unsigned)(w[j >> 5] >> (2 * (j & 31))) & 3u;
		if (rc) s[len - 1 - j] = "TGCA"[c]; else s[j] = "ACGT"[c];
	}
	return s;
}
inline std::string revcomp(const std::string& s)
{
	std::string r(s.rbegin(), s.rend());
	for (auto& c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
	return r;
}

// readContigs, AdjList.cpp:192-230
inline void read_contigs(const std::string& path, const Options& o, Graph& g, std::unordered_set<std::string>& seen)
{
	if (o.verbose > 0) fprintf(stderr, "Reading `%s'...\n", path.c_str());
	abghost::ReaderOptions ro;
	ro.trimMasked = 0; // AdjList.cpp:387
	const unsigned overlap = o.k - 1, W = g.W;
	const auto flatten = [](std::string& seq) {
		for (auto& c : seq) { // flattenAmbiguityCodes(seq), N untouched
			switch (c) {
			case 'M': case 'R': case 'W': case 'V': case 'H': case 'D': c = 'A'; break;
			case 'S': case 'Y': case 'B': c = 'C'; break;
			case 'K': c = 'G'; break;
			default: break;
			}
		}
	};
	// (a plain FASTA file of some size is parsed block-parallel -- read_fasta_blocks -- and its records taken in file order below)
	std::vector<std::vector<abghost::FastaRecord>> parts;
	const bool blocks = abghost::read_fasta_blocks(path, ro, std::max(1u, std::thread::hardware_concurrency()), parts, [&](abghost::FastaRecord& r) { if (!isdigit((unsigned char)r.seq[0])) flatten(r.seq); });
	std::unique_ptr<abghost::FastaReader> in;
	if (!blocks) in.reset(new abghost::FastaReader(path, ro));
	size_t pb = 0, pi = 0;
	std::string id, comment, seq;
	const auto next = [&]() -> bool {
		if (!blocks) { if (!in->read(id, comment, seq)) return false; if (!isdigit((unsigned char)seq[0])) flatten(seq); return true; }
		while (pb < parts.size() && pi >= parts[pb].size()) { std::vector<abghost::FastaRecord>().swap(parts[pb]); pb++; pi = 0; }
		if (pb >= parts.size()) return false;
		abghost::FastaRecord& r = parts[pb][pi++];
		id.swap(r.id); comment.swap(r.comment); seq.swap(r.seq);
		return true;
	};
	while (next()) {
		if (isdigit((unsigned char)seq[0])) {
			fprintf(stderr, ABG_ADJ_PROGRAM ": `%s': colour-space contigs are not supported\n", path.c_str());
			exit(EXIT_FAILURE);
		}
		if (seq.length() <= overlap) { // (an assertion of the reference)
			fprintf(stderr, ABG_ADJ_PROGRAM ": `%s': contig `%s' is %zu bases long: contigs must be longer than k-1 = %u\n",
			    path.c_str(), id.c_str(), seq.length(), overlap);
			exit(EXIT_FAILURE);
		}
		if (!seen.insert(id).second) { // (g_contigNames.insert asserts on a duplicate)
			fprintf(stderr, ABG_ADJ_PROGRAM ": `%s': duplicate contig ID `%s'\n", path.c_str(), id.c_str());
			exit(EXIT_FAILURE);
		}
		const size_t at = g.head.size();
		g.head.resize(at + W);
		g.tail.resize(at + W);
		pack_key(seq, 0, overlap, &g.head[at], W);
		pack_key(seq, seq.length() - overlap, overlap, &g.tail[at], W);
		unsigned length = 0, coverage = 0; // getCoverage, AdjList.cpp:127-133
		{
			std::istringstream ss(comment);
			ss >> length >> coverage;
		}
		g.name.push_back(id);
		g.length.push_back((unsigned)seq.length());
		g.coverage.push_back(coverage);
		if (g.n() >= (1ull << 30)) { fprintf(stderr, ABG_ADJ_PROGRAM ": too many contigs\n"); exit(EXIT_FAILURE); }
	}
}

// printGraphStats (Graph/GraphUtil.h:43-64) over the out-degrees
inline void print_graph_stats(FILE* out, const Graph& g)
{
	std::map<int, uint64_t> h;
	for (uint64_t u = 0; u < g.nv(); u++) h[(int)g.degree(u)]++;
	abgio::print_graph_stats(out, (unsigned)g.nv(), (unsigned)g.edges(), h);
}

// addOverlapsSA, AdjList.cpp:137-189: overlaps of [m, k-1) bases between the vertices without a k-1 overlap
inline void add_short_overlaps(const Options& o, Graph& g)
{
	const unsigned len = o.k - 1, W = g.W;
	std::vector<uint32_t> blunt;
	for (uint64_t u = 0; u < g.nv(); u++) if (g.off[u + 1] == g.off[u]) blunt.push_back((uint32_t)u);
	// the last k-1 bases of every blunt vertex: reverseComplement(prefixes[u^1])
	std::vector<std::string> suffix(blunt.size());
	for (size_t b = 0; b < blunt.size(); b++) {
		const uint32_t u = blunt[b];
		suffix[b] = (u & 1) ? key_string(&g.head[(size_t)(u >> 1) * W], len, true) : key_string(&g.tail[(size_t)(u >> 1) * W], len, false);
	}
	// SuffixArray::insert + construct: the proper suffixes of length >= m, by (strcmp, vertex)
	typedef std::pair<const char*, uint32_t> Entry;
	std::vector<Entry> sa;
	for (size_t b = 0; b < blunt.size(); b++)
		for (unsigned at = 1; at + o.minOverlap <= len; at++) sa.push_back(Entry(suffix[b].c_str() + at, blunt[b]));
	std::sort(sa.begin(), sa.end(), [](const Entry& a, const Entry& b) {
		const int c = strcmp(a.first, b.first);
		return c < 0 || (c == 0 && a.second < b.second);
	});
	g.extra.assign(g.nv(), std::vector<Edge>());
	struct Cmp {
		bool operator()(const Entry& a, const char* b) const { return strcmp(a.first, b) < 0; }
		bool operator()(const char* a, const Entry& b) const { return strcmp(a, b.first) < 0; }
	};
	for (size_t b = 0; b < blunt.size(); b++) {
		const uint32_t v = blunt[b] ^ 1u; // the complement: its FIRST k-1 bases are rc(suffix)
		const std::string vseq = revcomp(suffix[b]);
		std::set<uint32_t> seen;
		for (std::string q(vseq, 0, vseq.size() - 1); q.size() >= o.minOverlap && !q.empty(); q.erase(q.size() - 1)) {
			auto range = std::equal_range(sa.begin(), sa.end(), q.c_str(), Cmp());
			for (auto it = range.first; it != range.second; ++it) {
				const uint32_t u = it->second;
				if (o.ss && ((u ^ v) & 1u)) continue;
				if (seen.insert(u).second) g.extra[u].push_back(Edge{ v, -(int)q.size() }); // the longest overlap of a pair
			}
			if (q.size() == 1) break; // (chop asserts length > 1)
		}
	}
}

// main(), AdjList.cpp:307-420, after option parsing
inline int run(const Options& o, const Join& join, FILE* fout)
{
	Graph g;
	g.k = o.k;
	g.W = (o.k - 1 + 31) / 32;
	{
		std::unordered_set<std::string> seen;
		for (auto& f : o.files) read_contigs(f, o, g, seen);
	}
	if (o.verbose > 0) fprintf(stderr, "Finding overlaps of exactly k-1 bp...\n");
	join(o.k - 1, g.n(), g.head.data(), g.tail.data(), o.ss != 0, g.off, g.tgt);
	if (g.off.size() != g.nv() + 1 || g.off.back() != g.tgt.size()) {
		fprintf(stderr, ABG_ADJ_PROGRAM ": the overlap join returned an inconsistent result\n");
		return EXIT_FAILURE;
	}
	if (o.verbose > 0) print_graph_stats(stderr, g);
	if (o.minOverlap < o.k - 1) {
		if (o.verbose > 0) fprintf(stderr, "Finding overlaps of fewer than k-1 bp...\n");
		add_short_overlaps(o, g);
		if (o.verbose > 0) print_graph_stats(stderr, g);
	}
	{
		abgio::Out out(fout);
		abgio::write_graph(out, g, o.format, ABG_ADJ_PROGRAM, o.commandLine);
	}
	if (fflush(fout) != 0 || ferror(fout)) { fprintf(stderr, ABG_ADJ_PROGRAM ": error writing the output\n"); return EXIT_FAILURE; }
	return EXIT_SUCCESS;
}

} // namespace abgadj
