// rresolver_core.h -- host side of the drop-in `abyss-rresolver-short` (the rule abyss-pe runs after AdjList in Bloom mode,
// bin/abyss-pe:581-585): options, the contig graph and its surgery, read statistics, the path-support logic and the writers.
// The two things that touch every read and every candidate sequence -- filling the Bloom filter of the reads' r-mers and
// counting the r-mers of candidate sequences it holds -- are the caller's `ReadFilter` (abg_rr_* on the GPU in the product
// binary, include/abyss_amd.h; tests/hostcheck substitutes the same device logic run serially).
//
// Reference behaviour restated here (ABySS 2.3.10):
//   RResolver/RResolverShort.cpp:26-402      options, usage, main, writeResults
//   RResolver/RAlgorithmsShort.cpp:71-96     window / margin arithmetic
//   RResolver/RAlgorithmsShort.cpp:98-308    determineShortReadStats (read sizes, their merging, r values)
//   RResolver/RAlgorithmsShort.cpp:369-605   testCombination, expectedSpacingBetweenReads, determinePathSupport
//   RResolver/RAlgorithmsShort.cpp:607-834   buildRepeatSupportMap, updateStats, isSmallRepeat, resolveRepeats
//   RResolver/RAlgorithmsShort.cpp:871-1229  processGraph (repeat instances, graph modification)
//   RResolver/RAlgorithmsShort.cpp:1231-1323 writeHistograms, resolveShort
//   RResolver/BloomFilters.cpp:139-209,211-297   loadReads, buildFilters
//   RResolver/Contigs.cpp                    sequences, comments, loadContigs / storeContigs, assembleContigs
//   RResolver/SequenceTree.cpp               getTreeSequences
//   Graph/DirectedGraph.h, ContigGraph.h     adjacency lists in insertion order; (u,v) implies (~v,~u)
//   Graph/ContigGraphAlgorithms.h:40-240     contiguous_out/in, assemble_if, merge, copy_in/out_edges
//   Graph/DotIO.h:150-309, AdjIO.h:99-190    readers;  Common/ContigID.h, Dictionary.h  contig names
// How it differs in structure: the reference tests one candidate sequence at a time against the filter as it walks the
// repeats (OpenMP tasks).  Whether a combination can be tested at all depends on lengths only, so here a pass first lists
// every sequence to test (in the reference's -j1 order, random_shuffle calls included), the filter answers them all in ONE
// batch, and a second pass folds the counts into the supports exactly as the reference's loop does.  The results are those
// of the reference at -j1.  -e (error correction with btllib's SeedBloomFilter) is not offered.
#pragma once

#include "fasta_reader.h"
#include "graph_writers.h"
#include "si_bytes.h"

#include <algorithm>
#include <chrono>
#include <future>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <getopt.h>
#include <iterator>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace abgrr {

#define ABG_RR_PROGRAM "abyss-rresolver-short"

using abgio::ADJ; using abgio::ASQG; using abgio::DOT; using abgio::GFA1; using abgio::GFA2; using abgio::SAM;

// RAlgorithmsShort.h:15-28, BloomFilters.h:12
const int MIN_MARGIN = 2;
const int R_HEURISTIC = 60;
const double R_HEURISTIC_A = 1.0, R_HEURISTIC_B = 0.0;
const int MAX_SUBITERATIONS = 2;
const long READ_STATS_SAMPLE_SIZE = 100000;
const double READ_BATCH_FRACTION_THRESHOLD = 0.1;
const double SUPPORTED_PATHS_MIN = 0.15;
const double COV_APPROX_FORMULA_FACTOR = 4.00;
const int HASH_NUM = 7;

struct Options { // namespace opt, RResolverShort.cpp:75-135
	size_t bloomSize = 0;
	int threads = 1;
	std::string histPrefix, outputGraphPath, outputContigsPath;
	int threshold = 4, extract = 4, minTests = 18, maxTests = 40, branching = 75;
	std::vector<int> rValues;
	std::vector<double> covApproxFactors;
	int readQualityThreshold = 35;
	int errorCorrection = 0;
	unsigned maxReadSize = 350;
	double bfMemFactor = 1.0;
	std::string outputSupportedPathsPath, outputUnsupportedPathsPath;
	unsigned k = 0;
	int format = ADJ;
	int verbose = 0;
	int device = 0;
	std::string commandLine, contigsPath, graphPath;
	std::vector<std::string> readFiles;
};

// What the host asks of the read filter (btllib::KmerBloomFilter in the reference).  Errors end the program.
struct ReadFilter {
	virtual ~ReadFilter() {}
	virtual void create(uint64_t bytes, unsigned hash_num, unsigned r) = 0; // a new, empty filter (the previous one is dropped)
	// insert(seq.substr(0, max_bases)) for the sequences of one of the lengths given (all when nlen is 0) that hold an r-mer
	virtual void insert(const char* seqs, const uint64_t* offsets, uint64_t n, uint32_t max_bases, const uint32_t* lengths, uint32_t nlen) = 0;
	virtual void contains(const char* seqs, const uint64_t* offsets, uint64_t n, uint32_t* found) = 0;
	virtual uint64_t popcount() = 0;
	virtual uint64_t bytes() = 0;
};

static const char USAGE_MESSAGE[] =
    "Usage: " ABG_RR_PROGRAM " [OPTION]... <contigs> <graph> [<reads1> <reads2> ...]\n"
    "Resolve unitig repeats using a sliding window and\n"
    "and short read information.\n"
    "\n"
    " Arguments:\n"
    "\n"
    "  <contigs>  contigs in FASTA format\n"
    "  <graph>    contig adjacency graph\n"
    "  <reads>    reads in FASTA format\n"
    "\n"
    " Options:\n"
    "\n"
    "  -b, --bloom-size=N          read Bloom filter size. Unit suffixes 'K' (kilobytes), 'M' (megabytes), or 'G' (gigabytes) may be used. [required]\n"
    "  -g, --graph=FILE            write the contig adjacency graph to FILE. [required]\n"
    "  -c, --contigs=FILE          write the contigs to FILE. [required]\n"
    "  -j, --threads=N             use N parallel threads [1]\n"
    "  -k, --kmer=N                assembly k-mer size\n"
    "  -h, --hist=PREFIX           write the algorithm histograms with the given prefix. Histograms are omitted if no prefix is given.\n"
    "  -t, --threshold=N           set path support threshold to N. [4]\n"
    "  -x, --extract=N             extract N r-mers per read. [4]\n"
    "  -m, --min-tests=N           set minimum number of sliding window moves to N. Cannot be higher than 127. [18]\n"
    "  -M, --max-tests=N           set maximum number of sliding window moves to N. Cannot be higher than 127. [40]\n"
    "  -n, --branching=N           set maximum number of branching paths to N. [75]\n"
    "  -r, --rmer=N                explicitly set r value (k value used by rresolver). The number of set r values should be equal to the number of read sizes.\n"
    "  -a, --approx-factor         explicitly set coverage approximation factor.\n"
    "  -q, --quality--threshold=N  minimum quality all bases in rmers should have, on average. [35] (UNUSED)\n"
    "  -R, --max-read-size         upper limit on read size to consider for use with RResolver. [350]\n"
    "  -f, --bf-mem-factor         factor to multiply Bloom filter memory budget with in order to stay within similar memory usage as the rest of the pipeline. [1.0]\n"
    "  -S, --supported=FILE        write supported paths to FILE.\n"
    "  -U, --unsupported=FILE      write unsupported paths to FILE.\n"
    "                              Used for path sequence quality check.\n"
    "      --adj                   output the graph in ADJ format [default]\n"
    "      --asqg                  output the graph in ASQG format\n"
    "      --dot                   output the graph in GraphViz format\n"
    "      --gfa                   output the graph in GFA1 format\n"
    "      --gfa1                  output the graph in GFA1 format\n"
    "      --gfa2                  output the graph in GFA2 format\n"
    "      --gv                    output the graph in GraphViz format\n"
    "      --sam                   output the graph in SAM format\n"
    "  -v, --verbose               display verbose output\n"
    "      --help                  display this help and exit\n"
    "      --version               output version information and exit\n"
    "      --gpu=N                 HIP device ordinal [0]\n"
    "\n"
    "The read Bloom filter is built and queried on the GPU.  -e (error correction) is not supported by this build.\n";

// main's option loop and check_options, RResolverShort.cpp:170-376.  Returns false when the caller should exit with `*status`.
inline bool parse_options(int argc, char** argv, Options& o, int* status)
{
	{
		std::ostringstream ss;
		for (int i = 0; i < argc; i++) ss << (i ? " " : "") << argv[i];
		o.commandLine = ss.str();
	}
	enum { OPT_HELP = 1, OPT_VERSION, OPT_GPU };
	static int format = ADJ, errorCorrection = 0;
	static const struct option longopts[] = {
		{ "bloom-size", required_argument, NULL, 'b' }, { "threads", required_argument, NULL, 'j' },
		{ "graph", required_argument, NULL, 'g' }, { "contigs", required_argument, NULL, 'c' },
		{ "kmer", required_argument, NULL, 'k' }, { "hist", required_argument, NULL, 'h' },
		{ "threshold", required_argument, NULL, 't' }, { "extract", required_argument, NULL, 'x' },
		{ "min-tests", required_argument, NULL, 'm' }, { "max-tests", required_argument, NULL, 'M' },
		{ "branching", required_argument, NULL, 'n' }, { "rmer", required_argument, NULL, 'r' },
		{ "approx-factor", required_argument, NULL, 'a' }, { "quality-threshold", required_argument, NULL, 'q' },
		{ "error-correction", no_argument, &errorCorrection, 1 }, { "max-read-size", required_argument, NULL, 'R' },
		{ "bf-mem-factor", required_argument, NULL, 'f' }, { "supported", required_argument, NULL, 'S' },
		{ "unsupported", required_argument, NULL, 'U' },
		{ "adj", no_argument, &format, ADJ }, { "asqg", no_argument, &format, ASQG }, { "dot", no_argument, &format, DOT },
		{ "gfa", no_argument, &format, GFA1 }, { "gfa1", no_argument, &format, GFA1 }, { "gfa2", no_argument, &format, GFA2 },
		{ "gv", no_argument, &format, DOT }, { "sam", no_argument, &format, SAM },
		{ "verbose", no_argument, NULL, 'v' }, { "help", no_argument, NULL, OPT_HELP }, { "version", no_argument, NULL, OPT_VERSION },
		{ "gpu", required_argument, NULL, OPT_GPU },
		{ NULL, 0, NULL, 0 }
	};
	bool die = false;
	for (int c; (c = getopt_long(argc, argv, "b:j:g:c:k:h:t:x:m:M:n:r:a:q:eR:f:S:U:v", longopts, NULL)) != -1;) {
		std::istringstream arg(optarg != NULL ? optarg : "");
		switch (c) {
		case '?': die = true; break;
		case 'b': {
			uint64_t b = 0;
			if (!si_to_bytes(optarg, &b)) { // (SIToBytes leaves the stream failed: the check below)
				fprintf(stderr, ABG_RR_PROGRAM ": invalid option: `-%c%s'\n", (char)c, optarg);
				*status = EXIT_FAILURE;
				return false;
			}
			o.bloomSize = (size_t)b;
			break;
		}
		case 'j': arg >> o.threads; break;
		case 'k': arg >> o.k; break;
		case 'h': arg >> o.histPrefix; break;
		case 'g': arg >> o.outputGraphPath; break;
		case 'c': arg >> o.outputContigsPath; break;
		case 't': arg >> o.threshold; break;
		case 'x': arg >> o.extract; break;
		case 'm': arg >> o.minTests; break;
		case 'M': arg >> o.maxTests; break;
		case 'n': arg >> o.branching; break;
		case 'r': { int r = 0; arg >> r; o.rValues.push_back(r); break; }
		case 'R': arg >> o.maxReadSize; break;
		case 'f': arg >> o.bfMemFactor; break;
		case 'a': { double a = 0; arg >> a; o.covApproxFactors.push_back(a); break; }
		case 'q': arg >> o.readQualityThreshold; break;
		case 'e': errorCorrection = 1; break;
		case 'S': arg >> o.outputSupportedPathsPath; break;
		case 'U': arg >> o.outputUnsupportedPathsPath; break;
		case 'v': ++o.verbose; break;
		case OPT_HELP: fputs(USAGE_MESSAGE, stdout); *status = EXIT_SUCCESS; return false;
		case OPT_VERSION:
			fputs(ABG_RR_PROGRAM " (ABySS, abyss_amd) " ABG_IO_VERSION "\n", stdout);
			*status = EXIT_SUCCESS;
			return false;
		case OPT_GPU: arg >> o.device; break;
		}
		if (optarg != NULL && c != 'b' && (!arg.eof() || arg.fail())) {
			fprintf(stderr, ABG_RR_PROGRAM ": invalid option: `-%c%s'\n", (char)c, optarg);
			*status = EXIT_FAILURE;
			return false;
		}
	}
	o.format = format;
	o.errorCorrection = errorCorrection;
	if (o.bloomSize == 0) { fprintf(stderr, ABG_RR_PROGRAM ": missing or invalid value for mandatory option `-b'\n"); die = true; }
	if (argc - optind < 3) { fprintf(stderr, ABG_RR_PROGRAM ": missing input file arguments\n"); die = true; }
	if (o.k <= 0) { fprintf(stderr, ABG_RR_PROGRAM ": missing or invalid value for mandatory option `-k'\n"); die = true; }
	if (o.readQualityThreshold <= 0) { fprintf(stderr, ABG_RR_PROGRAM ": invalid value for option `-q'\n"); die = true; }
	if (o.outputGraphPath.empty()) { fprintf(stderr, ABG_RR_PROGRAM ": missing or invalid value for mandatory option `-g`\n"); die = true; }
	if (o.outputContigsPath.empty()) { fprintf(stderr, ABG_RR_PROGRAM ": missing or invalid value for mandatory option `-c`\n"); die = true; }
	if (o.threads <= 0) { fprintf(stderr, ABG_RR_PROGRAM ": invalid number of threads `-j`\n"); die = true; }
	if (o.minTests > o.maxTests) { fprintf(stderr, ABG_RR_PROGRAM ": --min-tests cannot be higher than --max-tests\n"); die = true; }
	if (o.maxTests > 127) { fprintf(stderr, ABG_RR_PROGRAM ": --max-tests cannot be higher than 127\n"); die = true; } // (int8_t counts: the usage says so, the reference overflows)
	if (o.errorCorrection) { fprintf(stderr, ABG_RR_PROGRAM ": -e (error correction with spaced seeds) is not supported by this build\n"); die = true; }
	if (die) {
		fprintf(stderr, "Try `" ABG_RR_PROGRAM " --help' for more information.\n");
		*status = EXIT_FAILURE;
		return false;
	}
	o.contigsPath = argv[optind++];
	o.graphPath = argv[optind++];
	for (int i = optind; i < argc; i++) o.readFiles.push_back(argv[i]);
	return true;
}

[[noreturn]] inline void die(const std::string& msg)
{
	fprintf(stderr, "%s\n", msg.c_str());
	exit(EXIT_FAILURE);
}

// ---- the contig graph: ContigGraph<DirectedGraph<ContigProperties, Distance>> + g_contigNames -------------------------
typedef uint32_t V; // vertex = 2 * contig + sense (ContigNode)
struct Edge { V v; int d; };

struct Graph {
	unsigned k = 0;
	std::vector<std::vector<Edge>> adj; // out-edges in insertion order
	std::vector<unsigned> length, coverage; // per VERTEX (the reference keeps a copy of the properties on both)
	std::vector<bool> rem;
	// g_contigNames + g_nextContigName (Common/ContigID.h, Dictionary.h)
	std::vector<std::string> names;
	std::unordered_map<std::string, unsigned> index;
	unsigned nextName = 0;

	uint64_t nv() const { return adj.size(); }
	bool removed(uint64_t u) const { return u < rem.size() && rem[u]; }
	const std::string& cname(uint64_t u) const { return names[u >> 1]; }
	unsigned len(uint64_t u) const { return length[u]; }
	unsigned cov(uint64_t u) const { return coverage[u]; }
	template <class F> void for_out(uint64_t u, F f) const { for (const Edge& e : adj[u]) f(e.v, e.d); }
	std::string vname(V u) const { return names[u >> 1] + ((u & 1) ? '-' : '+'); }

	// Dictionary::put (Dictionary.h:48-60) through put(vertex_name, ...) (ContigNode.h:236-247)
	void put_name(V u, std::string name)
	{
		const char c = name.empty() ? 0 : name[name.size() - 1];
		if (c == '+' || c == '-') name.erase(name.size() - 1);
		const unsigned id = u >> 1;
		if (id < names.size()) {
			if (names[id] != name) die("error: the names of vertex " + std::to_string(u) + " do not agree: `" + names[id] + "', `" + name + "'");
			return;
		}
		if (id != names.size()) die("error: vertices out of order near `" + name + "'");
		if (!index.emplace(name, id).second) { fprintf(stderr, "error: duplicate ID: `%s'\n", name.c_str()); exit(EXIT_FAILURE); }
		names.push_back(name);
	}
	// find_vertex (ContigNode.h:283-303)
	V find_vertex(std::string name) const
	{
		if (name.size() < 2) die("error: unexpected ID: `" + name + "'");
		const char c = name[name.size() - 1];
		name.erase(name.size() - 1);
		if (c != '+' && c != '-') die("error: unexpected ID: `" + name + c + "'");
		return find_contig(name) * 2 + (c == '-');
	}
	unsigned find_contig(const std::string& name) const
	{
		auto it = index.find(name);
		if (it == index.end()) { fprintf(stderr, "error: unexpected ID: `%s'\n", name.c_str()); exit(EXIT_FAILURE); }
		return it->second;
	}
	// createContigName / setNextContigName (ContigID.h:31-61)
	std::string create_name()
	{
		if (nextName == 0) {
			unsigned mx = 0;
			for (const std::string& s : names) {
				std::istringstream iss(s);
				unsigned x;
				if (iss >> x && iss.eof() && x > mx) mx = x;
			}
			nextName = names.empty() ? 0 : 1 + mx;
		}
		return std::to_string(nextName++);
	}

	// DirectedGraph
	V add_vertex1(unsigned l, unsigned c) { adj.emplace_back(); length.push_back(l); coverage.push_back(c); return (V)(adj.size() - 1); }
	void add_edge1(V u, V v, int d) { adj[u].push_back(Edge{ v, d }); }
	void remove_edge1(V u, V v)
	{
		auto& e = adj[u];
		e.erase(std::remove_if(e.begin(), e.end(), [v](const Edge& x) { return x.v == v; }), e.end());
	}
	const Edge* find_edge(V u, V v) const
	{
		for (const Edge& e : adj[u]) if (e.v == v) return &e;
		return nullptr;
	}
	bool has_edge(V u, V v) const { return find_edge(u, v) != nullptr; }
	// get(edge_bundle, g, u, v) (ContigProperties.h:187-205)
	int dist(V u, V v) const
	{
		const Edge* e = find_edge(u, v);
		if (!e) die("error: no edge " + vname(u) + " -> " + vname(v));
		return e->d;
	}
	unsigned out_degree(V u) const { return (unsigned)adj[u].size(); }
	uint64_t num_edges() const { uint64_t n = 0; for (auto& a : adj) n += a.size(); return n; }
	// ContigGraph (ContigGraph.h:124-228)
	unsigned in_degree(V u) const { return (unsigned)adj[u ^ 1].size(); }
	V add_vertex(unsigned l, unsigned c) { const V v = add_vertex1(l, c); add_vertex1(l, c); return v; }
	void add_edge(V u, V v, int d) { add_edge1(u, v, d); if (u != (v ^ 1)) add_edge1(v ^ 1, u ^ 1, d); }
	void remove_edge(V u, V v) { remove_edge1(u, v); if (u != (v ^ 1)) remove_edge1(v ^ 1, u ^ 1); }
	void clear_out_edges(V u)
	{
		for (const Edge& e : adj[u]) if ((e.v ^ 1) != u) remove_edge1(e.v ^ 1, u ^ 1);
		adj[u].clear();
	}
	void clear_vertex(V v) { clear_out_edges(v); clear_out_edges(v ^ 1); }
	void remove_vertex(V v)
	{
		if (rem.size() < adj.size()) rem.resize(adj.size(), false);
		rem[v] = true;
		rem[v ^ 1] = true;
	}
	uint64_t num_removed() const { uint64_t n = 0; for (uint64_t u = 0; u < nv(); u++) n += removed(u); return n; }
};

inline void print_graph_stats(FILE* out, const Graph& g)
{
	std::map<int, uint64_t> h;
	for (uint64_t u = 0; u < g.nv(); u++) if (!g.removed(u)) h[(int)g.out_degree((V)u)]++;
	abgio::print_graph_stats(out, (unsigned)(g.nv() - g.num_removed()), (unsigned)g.num_edges(), h);
}

// ---- readers ---------------------------------------------------------------------------------------------------
// A cursor over the whole file with the stream manipulators the reference parses with (Common/IOUtil.h:38-84)
struct Cursor {
	const std::string& s;
	size_t p = 0;
	explicit Cursor(const std::string& s) : s(s) {}
	bool eof() const { return p >= s.size(); }
	int peek() const { return eof() ? EOF : (unsigned char)s[p]; }
	void ws() { while (!eof() && isspace((unsigned char)s[p])) p++; }
	void expect(const char* pat) // operator>>(istream&, expect)
	{
		for (const char* q = pat; *q; ++q) {
			if (*q == ' ') { ws(); continue; }
			if (eof() || s[p] != *q) {
				fprintf(stderr, "error: Expected `%s' and saw ", q);
				if (eof()) fprintf(stderr, "end-of-file\n");
				else {
					const size_t e = s.find('\n', p);
					fprintf(stderr, "`%c'\nnear: %s\n", s[p], s.substr(p, e == std::string::npos ? e : e - p).c_str());
				}
				exit(EXIT_FAILURE);
			}
			p++;
		}
	}
	void ignore(char delim) { const size_t e = s.find(delim, p); p = e == std::string::npos ? s.size() : e + 1; }
	bool number(long long& x) // operator>>(int): leading whitespace, sign, digits
	{
		ws();
		size_t q = p;
		if (q < s.size() && (s[q] == '-' || s[q] == '+')) q++;
		if (q >= s.size() || !isdigit((unsigned char)s[q])) return false;
		x = strtoll(s.c_str() + p, nullptr, 10);
		while (q < s.size() && isdigit((unsigned char)s[q])) q++;
		p = q;
		return true;
	}
	unsigned uns(const char* what) { long long x; if (!number(x) || x < 0) die(std::string("error: expected a number (") + what + ")"); return (unsigned)x; }
	int integer(const char* what) { long long x; if (!number(x)) die(std::string("error: expected a number (") + what + ")"); return (int)x; }
	bool quoted(std::string& out) // read_dot_name
	{
		ws();
		if (peek() != '"') return false;
		p++;
		const size_t e = s.find('"', p);
		out = s.substr(p, e == std::string::npos ? e : e - p);
		p = e == std::string::npos ? s.size() : e + 1;
		return true;
	}
};
// operator>>(istream&, ContigProperties&) (ContigProperties.h:112-126) and Distance (:160-168)
inline void read_props(Cursor& in, unsigned& l, unsigned& c)
{
	in.ws();
	if (in.peek() == 'l') {
		in.expect("l =");
		l = in.uns("l");
		in.ws();
		if (in.peek() == 'C') { in.expect("C ="); c = in.uns("C"); }
	} else if (in.peek() == 'C') {
		in.expect("C=");
		c = in.uns("C");
		in.expect(", l=");
		l = in.uns("l");
	} else {
		l = in.uns("length");
		c = in.uns("coverage");
	}
}
inline int read_distance(Cursor& in)
{
	in.expect(" d = ");
	if (in.peek() == '"') { in.expect("\""); const int d = in.integer("d"); in.expect("\""); return d; }
	return in.integer("d");
}

// read_dot (Graph/DotIO.h:150-309) on the plain directed graph: every line of the file is one vertex or one edge
inline void read_dot(const std::string& text, Graph& g, Options& o)
{
	Cursor in(text);
	in.ws();
	in.expect("digraph");
	in.ignore('{');
	int defd = -(int)o.k + 1;
	for (bool done = false; !done;) {
		in.ws();
		if (in.eof()) break;
		switch (in.peek()) {
		case 'g':
			in.expect("graph [ ");
			if (in.peek() == 'k') {
				in.expect("k =");
				const unsigned k = in.uns("k");
				if (o.k > 0 && k != o.k) die("error: the graph was built with k=" + std::to_string(k) + ", not " + std::to_string(o.k));
				o.k = k;
				defd = -(int)o.k + 1;
			}
			in.ignore(']');
			break;
		case 'e':
			in.expect("edge [");
			defd = read_distance(in);
			in.ignore(']');
			break;
		default: done = true; break;
		}
		in.ws();
		if (in.peek() == ';') in.p++;
	}
	const bool addVertices = g.nv() == 0;
	for (std::string uname; in.quoted(uname);) {
		in.ws();
		if (in.eof()) die("error: unexpected end of the graph file");
		const char c = in.s[in.p++];
		if (c == ';' || c == '[') {
			unsigned l = 0, cv = 0;
			if (c == '[') { read_props(in, l, cv); in.ignore(']'); }
			if (addVertices) { const V u = g.add_vertex1(l, cv); g.put_name(u, uname); }
			else {
				const V u = g.find_vertex(uname);
				if (c == '[' && (g.length[u] != l || g.coverage[u] != cv)) die("error: vertex properties do not agree: \"" + uname + "\"");
			}
		} else if (c == '-') {
			in.expect(">");
			const V u = g.find_vertex(uname);
			in.ws();
			if (in.peek() == '{') {
				in.expect("{");
				for (std::string vn; in.quoted(vn);) g.add_edge1(u, g.find_vertex(vn), defd);
				in.expect(" }");
			} else {
				std::string vn;
				if (!in.quoted(vn)) { fprintf(stderr, "error: Expected `\"' and saw `%c'.\n", (char)in.peek()); exit(EXIT_FAILURE); }
				const V v = g.find_vertex(vn);
				int d = defd;
				in.ws();
				if (in.peek() == '[') { in.expect("["); d = read_distance(in); in.ignore(']'); }
				if (g.has_edge(u, v)) { // DisallowParallelEdges, GraphIO.h:84-92
					fprintf(stderr, "error: parallel edges: [d=%d], [d=%d]\n", g.dist(u, v), d);
					exit(EXIT_FAILURE);
				}
				g.add_edge1(u, v, d);
			}
		} else {
			fprintf(stderr, "error: Expected `[' or `->' and saw `%c'.\n", c);
			exit(EXIT_FAILURE);
		}
		in.ws();
		if (in.peek() == ';') in.p++;
	}
	in.expect("}");
	in.ws();
	if (!in.eof()) die("error: Expected end-of-file after the graph");
	if (g.nv() == 0) die("error: the graph has no vertices");
}

// read_adj (Graph/AdjIO.h:99-190), ADJ format: "name length coverage\t; out-edges\t; in-edges"
inline void read_adj(const std::string& text, Graph& g, const Options& o)
{
	const int defd = -(int)o.k + 1;
	std::vector<std::pair<size_t, size_t>> lines; // [begin, end)
	for (size_t a = 0; a < text.size();) {
		size_t e = text.find('\n', a);
		if (e == std::string::npos) e = text.size();
		if (e > a) lines.push_back({ a, e });
		a = e + 1;
	}
	if (lines.empty()) die("error: the graph file is empty");
	{
		const std::string first = text.substr(lines[0].first, lines[0].second - lines[0].first);
		if (std::count(first.begin(), first.end(), ';') != 2) die("error: the graph is neither GraphViz nor ADJ format (the formats this build reads)");
	}
	for (auto& ln : lines) {
		std::istringstream ss(text.substr(ln.first, ln.second - ln.first));
		std::string name;
		unsigned l = 0, c = 0;
		if (!(ss >> name >> l >> c)) die("error: malformed ADJ line");
		const V u = g.add_vertex(l, c);
		g.put_name(u, name);
	}
	for (auto& ln : lines) {
		const std::string line = text.substr(ln.first, ln.second - ln.first);
		const size_t s1 = line.find(';'), s2 = line.find(';', s1 + 1);
		std::istringstream head(line.substr(0, s1));
		std::string name;
		head >> name;
		const V u = 2 * g.find_contig(name);
		for (unsigned sense = 0; sense < 2; sense++) {
			std::string part = sense == 0 ? line.substr(s1 + 1, s2 - s1 - 1) : line.substr(s2 + 1);
			Cursor in(part);
			for (;;) {
				in.ws();
				if (in.eof()) break;
				size_t e = in.p;
				while (e < part.size() && !isspace((unsigned char)part[e])) e++;
				const std::string vn = part.substr(in.p, e - in.p);
				in.p = e;
				in.ws();
				const V v = g.find_vertex(vn);
				int d = defd;
				if (in.peek() == '[') { in.p++; d = read_distance(in); in.ignore(']'); }
				if (g.has_edge(u ^ sense, v ^ sense)) die("error: parallel edges in the ADJ file near `" + vn + "'");
				g.add_edge1(u ^ sense, v ^ sense, d);
			}
		}
	}
}

inline std::string slurp(const std::string& path)
{
	std::ifstream f(path, std::ios::binary);
	if (!f.good()) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
	std::ostringstream ss;
	ss << f.rdbuf();
	return ss.str();
}

inline std::string reverse_complement(const std::string& s)
{
	std::string r(s.rbegin(), s.rend());
	for (auto& c : r) {
		switch (c) { // complementBaseChar, Common/Sequence.cpp:21-47
		case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break;
		case 'a': c = 't'; break; case 'c': c = 'g'; break; case 'g': c = 'c'; break; case 't': c = 'a'; break;
		case 'N': case 'n': case '.': break;
		case 'M': c = 'K'; break; case 'R': c = 'Y'; break; case 'W': case 'S': break; case 'Y': c = 'R'; break; case 'K': c = 'M'; break;
		case 'V': c = 'B'; break; case 'H': c = 'D'; break; case 'D': c = 'H'; break; case 'B': c = 'V'; break;
		default: fprintf(stderr, "error: unexpected character: `%c'\n", c); exit(EXIT_FAILURE);
		}
	}
	return r;
}

// ---- supports and read sizes (RAlgorithmsShort.h:85-187) -----------------------------------------------------
enum UnknownReason : uint8_t {
	UNDETERMINED = 0, TOO_MANY_COMBINATIONS, OVER_MAX_TESTS, POSSIBLE_TESTS_LT_PLANNED, WINDOW_NOT_LONG_ENOUGH,
	HEAD_SHORTER_THAN_MARGIN, TAIL_SHORTER_THAN_MARGIN, DIFFERENT_CULPRIT
};
struct Support {
	int8_t found = -1, tests = -1, calculatedTests = -1;
	UnknownReason reason = UNDETERMINED;
	static int8_t clamp(long c) { return (int8_t)(c > 127 ? 127 : c); }
	static Support unknown_of(UnknownReason r) { Support s; s.reason = r; return s; }
	static Support unknown_of(long calculated, UnknownReason r) { Support s; s.calculatedTests = clamp(calculated); s.reason = r; return s; }
	static Support known(int found, int tests) { Support s; s.found = (int8_t)found; s.tests = (int8_t)tests; return s; }
	bool unknown() const { return tests == -1; }
	void reset() { found = -1; tests = -1; }
};
struct ReadSize {
	int size = 0;
	std::set<int> sizeAndMergedSizes;
	std::vector<int> rValues;
	long sampleCount = 0;
	double covApproxFactor = COV_APPROX_FORMULA_FACTOR;
};

typedef std::map<unsigned, std::map<unsigned, Support>> SupportMap;
typedef std::map<unsigned, SupportMap> RepeatSupportMap;
typedef std::map<int, size_t> Histogram; // Common/Histogram.h: value -> count, printed "value\tcount\n"
typedef std::vector<std::pair<V, int>> ImaginaryContigPath;
struct PathLess { // std::set<ImaginaryContigPath>: ContigNode compares as a signed index (ContigNode.h:57-60)
	bool operator()(const ImaginaryContigPath& a, const ImaginaryContigPath& b) const
	{
		return std::lexicographical_compare(a.begin(), a.end(), b.begin(), b.end(), [](const std::pair<V, int>& x, const std::pair<V, int>& y) {
			if ((int)x.first != (int)y.first) return (int)x.first < (int)y.first;
			return x.second < y.second;
		});
	}
};
typedef std::set<ImaginaryContigPath, PathLess> ImaginaryContigPaths;

struct Resolution {
	RepeatSupportMap repeatSupportMap;
	int r = 0;
	Histogram findsHistogram, fractionFindsHistogram, calculatedTestsHistogram;
	bool failed = false;
};

class Resolver {
  public:
	Resolver(Options& o, ReadFilter& f) : opt(o), filter(f) {}

	Options& opt;
	ReadFilter& filter;
	Graph g;
	std::vector<std::string> seqs;     // g_contigSequences: per vertex, both orientations
	std::vector<std::string> comments; // g_contigComments: per contig
	std::vector<ReadSize> readSizes;
	ReadSize current;
	long readsSampleSize = 0;
	unsigned r = 0;                    // g_vanillaBloom->get_k()
	ImaginaryContigPaths supportedPaths, unsupportedPaths;
	// progress (RUtils.cpp:10-64)
	std::string progressName;
	unsigned progressNumber = 0, progressTotal = 0, progressUpdates = 0, progressLastPrinted = 0;

	// ---- Contigs.cpp ----
	void load_graph()
	{
		if (opt.verbose) fprintf(stderr, "Loading contig graph from `%s'...\n", opt.graphPath.c_str());
		const std::string text = slurp(opt.graphPath);
		size_t p = 0;
		while (p < text.size() && isspace((unsigned char)text[p])) p++;
		const int c = p < text.size() ? text[p] : EOF;
		g.k = opt.k;
		if (c == 'd') read_dot(text, g, opt);
		else if (c == '@' || c == 'H' || c == '>' || c == 'g')
			die("error: `" + opt.graphPath + "': this build reads the contig graph in GraphViz (--dot) or ADJ format only");
		else read_adj(text, g, opt);
		g.k = opt.k;
		if (g.nv() & 1) die("error: `" + opt.graphPath + "': a contig is missing one of its two vertices");
		if (opt.verbose) { fprintf(stderr, "Contig graph loaded.\n"); print_graph_stats(stderr, g); }
	}
	void load_contigs()
	{
		if (opt.verbose) fprintf(stderr, "Loading contigs from `%s'...\n", opt.contigsPath.c_str());
		abghost::ReaderOptions ro;
		ro.foldCase = 0; // FastaReader::NO_FOLD_CASE, Contigs.cpp:128
		auto take = [&](const std::string& id, std::string& comment, std::string& s, std::string* rc) {
			auto it = g.index.find(id);
			if (it == g.index.end()) return;
			if (seqs.size() / 2 != it->second) die("error: `" + opt.contigsPath + "': contig `" + id + "' is out of the graph's order");
			comments.push_back(std::move(comment));
			if (rc) { seqs.push_back(std::move(s)); seqs.push_back(std::move(*rc)); }
			else { std::string r = reverse_complement(s); seqs.push_back(std::move(s)); seqs.push_back(std::move(r)); }
		};
		// (a plain FASTA file of some size: parsed block-parallel, the reverse complements made where the records are parsed --
		// one thread reads ~200 MB/s, and the unitigs of a genome are tens of megabytes the reads' filter waits for)
		bool done = false;
		if (!(getenv("ABG_RR_SERIAL_CONTIGS") && atoi(getenv("ABG_RR_SERIAL_CONTIGS")))) {
			std::vector<std::vector<abghost::FastaRecord>> parts;
			if (contigs_ahead_.valid()) { done = contigs_ahead_.get(); parts.swap(contigs_parts_); } // (parsed while the graph was read: run())
			else done = abghost::read_fasta_blocks(opt.contigsPath, ro, (unsigned)std::max(1, opt.threads), parts,
			    [&](abghost::FastaRecord& r) { if (g.index.count(r.id)) r.aux = reverse_complement(r.seq); });
			if (done) for (auto& part : parts) for (abghost::FastaRecord& r : part) take(r.id, r.comment, r.seq, &r.aux);
		}
		if (!done) {
			abghost::FastaReader in(opt.contigsPath, ro);
			std::string id, comment, s;
			while (in.read(id, comment, s)) take(id, comment, s, nullptr);
		}
		if (seqs.empty()) die("error: `" + opt.contigsPath + "': no contig of the graph found");
		if (seqs.size() != g.nv()) die("error: `" + opt.contigsPath + "': " + std::to_string(g.nv() / 2 - seqs.size() / 2) + " contigs of the graph are missing");
		if (isdigit((unsigned char)seqs.front()[0])) die("error: colour-space contigs are not supported");
		if (opt.verbose) fprintf(stderr, "Contigs loaded.\n");
	}
	int contig_size(V u) const { return (int)seqs[u].size(); }
	std::string path_sequence(const std::vector<V>& path) const // getPathSequence, Contigs.cpp:44-62
	{
		std::string s = seqs[path[0]];
		for (size_t i = 1; i < path.size(); i++) {
			const int overlap = -g.dist(path[i - 1], path[i]);
			const std::string& t = seqs[path[i]];
			if (overlap < 0 || (int)s.size() < overlap || (int)t.size() < overlap || s.compare(s.size() - overlap, overlap, t, 0, overlap) != 0)
				die("error: contigs " + g.vname(path[i - 1]) + " and " + g.vname(path[i]) + " do not overlap as the graph says");
			s.append(t, overlap, std::string::npos);
		}
		return s;
	}
	std::string path_sequence(const ImaginaryContigPath& path) const // Contigs.cpp:64-82
	{
		std::string s = seqs[path[0].first];
		for (size_t i = 1; i < path.size(); i++) {
			const int overlap = -path[i].second;
			const std::string& t = seqs[path[i].first];
			if (overlap < 0 || (int)s.size() < overlap || (int)t.size() < overlap) die("error: a stored path does not overlap as recorded");
			s.append(t, overlap, std::string::npos);
		}
		return s;
	}
	double contig_base_coverage(V u) const // Contigs.cpp:90-95
	{
		return double((long)g.coverage[u]) * opt.k / double(seqs[u].size() - opt.k + 1);
	}
	void store_contigs(const std::string& path)
	{
		if (opt.verbose) fprintf(stderr, "Storing contigs to `%s'...\n", path.c_str());
		FILE* f = fopen(path.c_str(), "w");
		if (!f) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		{
			abgio::Out out(f);
			for (uint64_t u = 0; u < g.nv(); u += 2) {
				if (g.removed(u)) continue;
				out << '>' << g.cname(u);
				if (!comments[u >> 1].empty()) out << ' ' << comments[u >> 1];
				out << '\n' << seqs[u] << '\n';
			}
		}
		if (fclose(f) != 0) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		if (opt.verbose) fprintf(stderr, "Contigs stored.\n");
	}
	void store_graph(const std::string& path)
	{
		if (opt.verbose) fprintf(stderr, "Storing contig graph to `%s'...\n", path.c_str());
		FILE* f = fopen(path.c_str(), "w");
		if (!f) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		{
			abgio::Out out(f);
			abgio::write_graph(out, g, opt.format, ABG_RR_PROGRAM, opt.commandLine);
		}
		if (fclose(f) != 0) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		if (opt.verbose) fprintf(stderr, "Contig graph stored.\n");
	}

	// assemble_if + merge (ContigGraphAlgorithms.h:118-240) on a copy, then assembleContigs (Contigs.cpp:199-258)
	static bool contiguous_out(const Graph& h, V u) { return h.out_degree(u) == 1 && h.in_degree(h.adj[u][0].v) == 1; }
	static bool contiguous_in(const Graph& h, V u) { return contiguous_out(h, u ^ 1); }
	static void copy_out_edges(Graph& h, V u, V uout)
	{
		bool palindrome = false;
		int pd = 0;
		const size_t n = h.adj[u].size(); // (u's own list does not change in here)
		for (size_t i = 0; i < n; i++) {
			const Edge e = h.adj[u][i];
			if ((e.v ^ 1) == u) { palindrome = true; pd = e.d; }
			else h.add_edge(uout, e.v, e.d);
		}
		if (palindrome) { h.add_edge(uout, u ^ 1, pd); h.add_edge(uout, uout ^ 1, pd); }
	}
	void assemble_contigs()
	{
		if (opt.verbose) fprintf(stderr, "Assembling contigs... ");
		std::vector<std::vector<V>> paths;
		Graph h(g);
		const uint64_t nv0 = h.nv();
		for (uint64_t ui = 0; ui < nv0; ui++) {
			V u = (V)ui;
			if (!contiguous_out(h, u) || contiguous_in(h, u) || u == (h.adj[u][0].v ^ 1)) continue;
			std::vector<V> path;
			while (contiguous_out(h, u)) {
				const V v = h.adj[u][0].v;
				if (u == (v ^ 1)) break; // IsPalindrome
				path.push_back(u);
				u = v;
			}
			path.push_back(u);
			// merge: a new vertex with the summed properties, the path's in- and out-edges
			unsigned l = h.length[path[0]], c = h.coverage[path[0]];
			for (size_t i = 1; i < path.size(); i++) {
				l += h.dist(path[i - 1], path[i]);
				l += h.length[path[i]];
				c += h.coverage[path[i]];
			}
			const V nu = h.add_vertex(l, c);
			copy_out_edges(h, path.front() ^ 1, nu ^ 1); // copy_in_edges
			copy_out_edges(h, path.back(), nu);
			for (V x : path) h.clear_vertex(x);
			for (V x : path) h.remove_vertex(x);
			paths.push_back(path);
		}
		// (the names are one dictionary for both graphs: kept on g, handed to the new graph at the end)
		for (size_t i = 0; i < paths.size(); i++) g.put_name((V)(2 * (g.nv() / 2 + i)) + (paths[i][0] & 1), g.create_name());
		for (const auto& path : paths) {
			const std::string s = path_sequence(path);
			unsigned coverage = 0;
			for (V x : path) coverage += g.coverage[x];
			std::string comment = std::to_string(s.size()) + ' ' + std::to_string(coverage) + ' ' + g.vname(path.front());
			if (path.size() == 3) comment += ',' + g.vname(path[1]);
			else if (path.size() > 3) comment += ",...";
			comment += ',' + g.vname(path.back());
			seqs.push_back(s);
			seqs.push_back(reverse_complement(s));
			comments.push_back(comment);
		}
		h.names = g.names; h.index = g.index; h.nextName = g.nextName;
		g = std::move(h);
		if (opt.verbose) fprintf(stderr, "Done!\n");
	}

	// ---- SequenceTree.cpp ----
	struct TreeNode { V v; int start, length, maxLength; };
	TreeNode tree_node(V v, int overlap, int maxLength, bool forward) const
	{
		const int size = (int)seqs[v].size();
		const int end = std::min(overlap + maxLength, size);
		TreeNode n{ v, overlap, end - overlap, maxLength };
		if (n.length <= 0) die("error: contig " + g.vname(v) + " is no longer than its overlap");
		if (!forward) n.start = size - end;
		return n;
	}
	std::vector<std::string> tree_sequences(V start, int overlap, int maxLength, bool forward, int maxPaths) const
	{
		std::vector<std::vector<TreeNode>> traces;
		std::deque<size_t> queue;
		traces.push_back({ tree_node(start, overlap, maxLength, forward) });
		queue.push_back(0);
		int leaves = 1;
		while (!queue.empty()) {
			const size_t ti = queue.front();
			queue.pop_front();
			const TreeNode node = traces[ti].back();
			std::vector<TreeNode> children;
			if (node.maxLength > node.length) {
				if (forward) {
					for (const Edge& e : g.adj[node.v]) children.push_back(tree_node(e.v, -e.d, node.maxLength - node.length, forward));
				} else {
					for (const Edge& e : g.adj[node.v ^ 1]) {
						const V intig = e.v ^ 1;
						children.push_back(tree_node(intig, -g.dist(intig, node.v), node.maxLength - node.length, forward));
					}
				}
			}
			if (!children.empty() && leaves + (int)children.size() - 1 <= maxPaths) {
				for (size_t i = 0; i < children.size(); i++) {
					if (i + 1 < children.size()) {
						std::vector<TreeNode> copy = traces[ti];
						copy.push_back(children[i]);
						traces.push_back(std::move(copy));
						queue.push_back(traces.size() - 1);
					} else {
						traces[ti].push_back(children[i]);
						queue.push_back(ti);
					}
				}
				leaves += (int)children.size() - 1;
			}
		}
		std::vector<std::string> out;
		out.reserve(traces.size());
		for (const auto& trace : traces) {
			std::string s;
			if (forward) for (auto it = trace.begin(); it != trace.end(); ++it) s.append(seqs[it->v], it->start, it->length);
			else for (auto it = trace.rbegin(); it != trace.rend(); ++it) s.append(seqs[it->v], it->start, it->length);
			out.push_back(std::move(s));
		}
		return out;
	}

	// ---- RUtils.cpp ----
	void progress_start(const std::string& name, unsigned total)
	{
		if (!opt.verbose) return;
		progressName = name;
		std::string t = name;
		t[0] = (char)tolower((unsigned char)t[0]);
		fprintf(stderr, "\n%u. Starting %s...\nProgress: 0%%", ++progressNumber, t.c_str());
		fflush(stderr);
		progressTotal = total; progressUpdates = 0; progressLastPrinted = 0;
	}
	void progress_update()
	{
		if (!opt.verbose) return;
		progressUpdates++;
		const double fraction = double(progressUpdates) / progressTotal;
		if (progressUpdates == progressTotal) fprintf(stderr, "\rProgress: 100%%\n%s done.\n", progressName.c_str());
		else if (double(progressUpdates - progressLastPrinted) / progressTotal >= 0.01 && progressUpdates < progressTotal) {
			fprintf(stderr, "\rProgress: %d%%", int(fraction * 100.0));
			fflush(stderr);
			progressLastPrinted = progressUpdates;
		}
	}

	// ---- read statistics: determineShortReadStats, RAlgorithmsShort.cpp:98-308 ----
	static double fraction_of_total(const ReadSize& b, long total) { return double(b.sampleCount) / double(total); }
	abghost::ReaderOptions raw_reader() const
	{
		// btllib::SeqReader hands the records over as they are, upper-cased: no chastity filter, no trimming
		abghost::ReaderOptions ro;
		ro.chastityFilter = 0;
		ro.trimMasked = 0;
		return ro;
	}
	bool determine_read_stats()
	{
		if (opt.verbose) fprintf(stderr, "Determining read stats...\n");
		readSizes.clear();
		// (the files' samples are taken side by side when all of them can be opened -- a file that cannot is the sequential loop's to
		// report -- and merged in file order)
		std::vector<Histogram> hists(opt.readFiles.size());
		const auto sample = [&](size_t f, bool worker) {
			abghost::FastaReader reader(opt.readFiles[f], raw_reader());
			if (worker) reader.on_worker_thread();
			std::string id, comment, s;
			for (long num = 0; num < READ_STATS_SAMPLE_SIZE && reader.read(id, comment, s); num++) {
				if (s.size() > opt.maxReadSize) continue;
				hists[f][(int)s.size()]++;
			}
		};
		bool side_by_side = opt.readFiles.size() > 1 && opt.threads > 1;
		for (const std::string& filename : opt.readFiles) if (filename == "-" || access(filename.c_str(), R_OK) != 0) side_by_side = false;
		if (side_by_side) {
			std::vector<std::thread> pool;
			for (size_t f = 0; f < opt.readFiles.size(); f++) pool.emplace_back([&, f]() { sample(f, true); });
			for (auto& t : pool) t.join();
		} else
			for (size_t f = 0; f < opt.readFiles.size(); f++) sample(f, false);
		for (size_t f = 0; f < opt.readFiles.size(); f++) {
			const Histogram& hist = hists[f];
			for (const auto& kv : hist) {
				ReadSize* batch = nullptr;
				for (auto& b : readSizes) if (b.size == kv.first) { batch = &b; break; }
				if (!batch) { readSizes.push_back(ReadSize()); batch = &readSizes.back(); batch->size = kv.first; }
				batch->sampleCount += (long)kv.second;
			}
		}
		readsSampleSize = 0;
		for (const auto& b : readSizes) readsSampleSize += b.sampleCount;
		if (readSizes.empty()) { fprintf(stderr, "Insufficient number of short reads. Finishing...\n"); return false; }
		std::sort(readSizes.begin(), readSizes.end(), [](ReadSize a, ReadSize b) { return a.size < b.size; });
		// sizes within 2 of one another are one read size (up to three merged into a fourth), named after the most frequent
		std::vector<ReadSize> merged;
		std::set<size_t> skip;
		for (size_t i = 0; i + 1 < readSizes.size(); i++) {
			if (skip.count(i)) continue;
			int mergeCount = 0;
			readSizes[i].sizeAndMergedSizes.insert(readSizes[i].size);
			for (size_t j = i + 1; j < readSizes.size(); j++) {
				if (readSizes[j].size - readSizes[i].size <= 2) {
					readSizes[i].sizeAndMergedSizes.insert(readSizes[j].size);
					if (readSizes[i].sampleCount <= readSizes[j].sampleCount) readSizes[i].size = readSizes[j].size;
					readSizes[i].sampleCount += readSizes[j].sampleCount;
					skip.insert(j);
					if (++mergeCount >= 3) break;
				}
			}
			merged.push_back(readSizes[i]);
		}
		if (!skip.count(readSizes.size() - 1)) {
			readSizes.back().sizeAndMergedSizes.insert(readSizes.back().size);
			merged.push_back(readSizes.back());
		}
		readSizes = merged;
		std::sort(readSizes.begin(), readSizes.end(), [](ReadSize a, ReadSize b) { return a.sampleCount > b.sampleCount; });
		if (fraction_of_total(readSizes[0], readsSampleSize) < READ_BATCH_FRACTION_THRESHOLD) {
			fprintf(stderr, "Insufficient reads of same size. Finishing...\n");
			return false;
		}
		{
			std::vector<ReadSize> kept;
			for (const auto& b : readSizes) if (fraction_of_total(b, readsSampleSize) >= READ_BATCH_FRACTION_THRESHOLD) kept.push_back(b);
			readSizes = kept;
		}
		std::sort(readSizes.begin(), readSizes.end(), [](ReadSize a, ReadSize b) { return a.size < b.size; });
		if (opt.verbose) {
			fprintf(stderr, "Read lengths determined to be: ");
			for (size_t i = 0; i < readSizes.size(); i++)
				fprintf(stderr, "%s%d (%f%%)", i ? ", " : "", readSizes[i].size, fraction_of_total(readSizes[i], readsSampleSize) * 100.0);
			fprintf(stderr, "\n");
		}
		if (!opt.rValues.empty() && opt.rValues.size() < readSizes.size()) {
			fprintf(stderr, "%zu r values provided, %zu needed.\n", opt.rValues.size(), readSizes.size());
			exit(-1);
		}
		std::sort(opt.rValues.begin(), opt.rValues.end());
		for (size_t i = 0; i < readSizes.size(); i++) {
			ReadSize& batch = readSizes[i];
			if (!opt.rValues.empty()) {
				const int rv = opt.rValues[i + (opt.rValues.size() - readSizes.size())];
				if (rv <= (int)opt.k) { fprintf(stderr, "r size (%d) must be larger than assembly k (%u).\n", rv, opt.k); exit(-1); }
				if (rv > batch.size - opt.extract + 1) {
					fprintf(stderr, "r size (%d) must be smaller than or equal to read size - extract + 1 (%d).\n", rv, batch.size - opt.extract + 1);
					exit(-1);
				}
				batch.rValues.push_back(rv);
			} else {
				const int rv = std::min({ int(opt.k + R_HEURISTIC), int(batch.size * R_HEURISTIC_A + R_HEURISTIC_B), int(batch.size - opt.extract + 1) });
				if (rv > (int)opt.k) batch.rValues.push_back(rv);
			}
		}
		if (opt.verbose) {
			fprintf(stderr, "Using r values: ");
			for (size_t i = 0; i < readSizes.size(); i++)
				for (size_t j = 0; j < readSizes[i].rValues.size(); j++) {
					fprintf(stderr, "%d (%d)", readSizes[i].rValues[j], readSizes[i].size);
					if (i + 1 < readSizes.size() || j + 1 < readSizes[i].rValues.size()) fprintf(stderr, ", ");
				}
			fprintf(stderr, "\n");
		}
		std::sort(opt.covApproxFactors.begin(), opt.covApproxFactors.end());
		for (size_t i = 0; i < readSizes.size(); i++) if (i < opt.covApproxFactors.size()) readSizes[i].covApproxFactor = opt.covApproxFactors[i];
		if (opt.verbose) {
			fprintf(stderr, "Using coverage approximation factors: ");
			for (size_t i = 0; i < readSizes.size(); i++) {
				std::ostringstream ss;
				ss << readSizes[i].covApproxFactor;
				fprintf(stderr, "%s%s (%d)", i ? ", " : "", ss.str().c_str(), readSizes[i].size);
			}
			fprintf(stderr, "\n");
		}
		return true;
	}

	// ---- buildFilters + loadReads, BloomFilters.cpp:139-297 ----
	void build_filters(int rv, size_t bytes)
	{
		if (opt.verbose) fprintf(stderr, "Building Bloom filter(s) for r value %d\n", rv);
		if (bytes == 0) die(ABG_RR_PROGRAM ": the Bloom filter would have no bytes (-b x -f)");
		if (opt.verbose > 1) fprintf(stderr, "Vanilla Bloom filter memory = %s\n", bytes_to_si(bytes).c_str());
		filter.create(bytes, HASH_NUM, (unsigned)rv);
		r = (unsigned)rv;
		// loadReads: every read once; the filter takes the first r + extract - 1 bases of the reads of the current size, and the
		// sizes' shares of the WHOLE read set replace those of the sample
		std::vector<uint64_t> lenHist; // reads by length, over all files
		const uint32_t span = (uint32_t)(rv + opt.extract - 1);
		std::vector<uint32_t> wanted(current.sizeAndMergedSizes.begin(), current.sizeAndMergedSizes.end());
		uint64_t total = 0;
		for (const std::string& path : opt.readFiles) {
			if (opt.verbose) fprintf(stderr, "Loading reads from `%s'...\n", path.c_str());
			abghost::SequenceReader reader(path, raw_reader(), (unsigned)std::min(32, std::max(1, opt.threads))); // (as in abyss-bloom-dbg: beyond 32 parser threads the run gets slower)
			std::vector<uint64_t> off;
			auto count = [&](uint64_t len) { if (len >= lenHist.size()) lenHist.resize(len + 1, 0); lenHist[len]++; total++; };
			if (reader.has_blocks()) {
				abghost::SequenceReader::Block b;
				double t_wait = 0, t_off = 0, t_ins = 0, t0 = tnow();
				while (reader.next_block(b)) {
					const double t1 = tnow();
					const uint64_t n = b.seq_end.size();
					off.resize(n + 1);
					off[0] = 0;
					for (uint64_t i = 0; i < n; i++) { off[i + 1] = b.seq_end[i]; count(off[i + 1] - off[i]); }
					const double t2 = tnow();
					filter.insert(b.seqs.data(), off.data(), n, span, wanted.data(), (uint32_t)wanted.size());
					const double t3 = tnow();
					t_wait += t1 - t0; t_off += t2 - t1; t_ins += t3 - t2; t0 = t3;
				}
				if (timing_) fprintf(stderr, "[host]   `%s': %.3f s waiting for the reader, %.3f s over the lengths, %.3f s in the filter's insert calls\n", path.c_str(), t_wait, t_off, t_ins);
			} else {
				std::string id, comment, s, buf;
				off.assign(1, 0);
				auto flush = [&]() {
					if (off.size() > 1) filter.insert(buf.data(), off.data(), off.size() - 1, span, wanted.data(), (uint32_t)wanted.size());
					buf.clear();
					off.assign(1, 0);
				};
				while (reader.read(id, comment, s)) {
					count(s.size());
					buf += s;
					off.push_back(buf.size());
					if (buf.size() >= (64u << 20)) flush();
				}
				flush();
			}
		}
		readsSampleSize = (long)total;
		auto reads_of = [&](const std::set<int>& sizes) {
			long n = 0;
			for (int s : sizes) if (s >= 0 && (size_t)s < lenHist.size()) n += (long)lenHist[s];
			return n;
		};
		current.sampleCount = reads_of(current.sizeAndMergedSizes);
		{
			// (a read counts for the FIRST read size whose set holds its length, BloomFilters.cpp:175-181)
			std::set<int> taken;
			for (auto& b : readSizes) {
				std::set<int> mine;
				for (int s : b.sizeAndMergedSizes) if (taken.insert(s).second) mine.insert(s);
				b.sampleCount = reads_of(mine);
			}
		}
		if (opt.verbose) {
			fprintf(stderr, "\nUpdated read lengths' fractions determined to be: ");
			for (size_t i = 0; i < readSizes.size(); i++)
				fprintf(stderr, "%s%d (%f%%)", i ? ", " : "", readSizes[i].size, fraction_of_total(readSizes[i], readsSampleSize) * 100.0);
			fprintf(stderr, "\n");
		}
		if (opt.verbose > 1) {
			const double occ = double(filter.popcount()) / double(filter.bytes() * 8);
			fprintf(stderr, "Vanilla Bloom filter (k = %u) occupancy = %.3g%%, FPR = %.3g%%\n", r, occ * 100.0, std::pow(occ, double(HASH_NUM)) * 100.0);
		}
	}
	static std::string bytes_to_si(size_t n) // Common/StringUtil.h:50-63
	{
		std::ostringstream s;
		s.precision(3);
		if (n < 1024) s << n;
		else if (n < (1ULL << 20)) s << (double)n / (1ULL << 10) << "k";
		else if (n < (1ULL << 30)) s << (double)n / (1ULL << 20) << "M";
		else s << (double)n / (1ULL << 30) << "G";
		return s.str();
	}

	// ---- path support, RAlgorithmsShort.cpp:71-96,369-605 ----
	static int min_window_length(int tests, int repeatSize, int minMargin) { return tests - 1 + minMargin + repeatSize + minMargin; }
	static bool window_long_enough(int windowSize, int tests, int repeatSize, int minMargin) { return windowSize >= min_window_length(tests, repeatSize, minMargin); }
	static int margin_of(int windowSize, int tests, int repeatSize) { return (windowSize + tests - 1 - repeatSize + 1) / 2; }

	double expected_spacing(V left, V repeat, V right) const
	{
		const long pathLength = 1000000;
		const double pathBaseCoverage = std::min({ contig_base_coverage(left), contig_base_coverage(repeat), contig_base_coverage(right) });
		const double pathBases = pathBaseCoverage * pathLength;
		double meanReadKmerContribution = 0;
		for (const auto& b : readSizes) meanReadKmerContribution += fraction_of_total(b, readsSampleSize) * (b.size - (int)opt.k + 1);
		const double baseContributionRatio = fraction_of_total(current, readsSampleSize) * (current.size - (int)opt.k + 1) / meanReadKmerContribution;
		const double approxNumOfReads = double(pathBases * baseContributionRatio) / double(opt.k * (current.size - opt.k + 1));
		return std::max(double(1.0), double(pathLength - current.size + 1) / double(approxNumOfReads));
	}

	// One path intig -> repeat -> outig: its support if lengths alone decide it, else the sequences the filter is to be asked about
	struct Pending {
		unsigned repeat, intig, outig; // vertex indices
		long calculatedTests = 0;
		bool decided = false;
		Support support;        // decided: the answer; else: what the loop over the combinations ends with if it meets an
		bool endsUnknown = false; // unknown combination (after the queries [q0, q1))
		size_t q0 = 0, q1 = 0;
	};
	struct Queries {
		std::string seqs;
		std::vector<uint64_t> off{ 0 };
		std::vector<uint32_t> found;
		size_t size() const { return off.size() - 1; }
	};
	// libstdc++'s std::random_shuffle (bits/stl_algo.h: for i = 1 .. n-1 swap v[i] with v[rand() % (i + 1)]): the reference calls it
	// on the heads and tails of a path with more than branching^2 combinations, and its rand() stream is part of the -j1
	// behaviour.  The reference never calls srand and nothing else in it draws from rand(), so the stream is glibc's from seed 1 --
	// kept here as a generator of our own (glibc's TYPE_3 additive feedback: r[i] = r[i-3] + r[i-31], output r[i] >> 1), because
	// in this process other libraries (the HIP runtime) may draw from rand() as well.
	struct GlibcRand {
		uint32_t r[34];
		int at = 0;
		GlibcRand()
		{
			int32_t x[344 + 34];
			x[0] = 1;
			for (int i = 1; i < 31; i++) {
				const int64_t hi = x[i - 1] / 127773, lo = x[i - 1] % 127773;
				int64_t w = 16807 * lo - 2836 * hi;
				if (w < 0) w += 2147483647;
				x[i] = (int32_t)w;
			}
			for (int i = 31; i < 34; i++) x[i] = x[i - 31];
			for (int i = 34; i < 344; i++) x[i] = (int32_t)((uint32_t)x[i - 31] + (uint32_t)x[i - 3]);
			for (int i = 0; i < 34; i++) r[i] = (uint32_t)x[344 - 34 + i];
		}
		int next()
		{
			// r holds the last 34 values as a ring; the new one is r[-31] + r[-3]
			const uint32_t v = r[(at + 34 - 31) % 34] + r[(at + 34 - 3) % 34];
			r[at] = v;
			at = (at + 1) % 34;
			return (int)(v >> 1);
		}
	};
	mutable GlibcRand rng;
	template <class T> void random_shuffle(std::vector<T>& v) const
	{
		for (size_t i = 1; i < v.size(); i++) {
			const size_t j = (size_t)(rng.next() % (long)(i + 1));
			if (i != j) std::swap(v[i], v[j]);
		}
	}
	Pending plan_path(V left, V repeat, V right, Queries& q) const
	{
		Pending p;
		p.repeat = repeat; p.intig = left; p.outig = right;
		const std::string& rep = seqs[repeat];
		const int repeatSize = (int)rep.size();
		const long calculatedTests = std::lround(expected_spacing(left, repeat, right) * current.covApproxFactor + opt.threshold);
		p.calculatedTests = calculatedTests;
		auto decide = [&](UnknownReason why) { p.decided = true; p.support = Support::unknown_of(calculatedTests, why); return p; };
		long requiredTests = std::max<long>(calculatedTests, opt.minTests);
		if (requiredTests > opt.maxTests) return decide(OVER_MAX_TESTS);
		const int windowSize = (int)r;
		if (!window_long_enough(windowSize, (int)requiredTests, repeatSize, MIN_MARGIN)) return decide(WINDOW_NOT_LONG_ENOUGH);
		const int leftDistance = g.dist(left, repeat), rightDistance = g.dist(repeat, right);
		const int margin = margin_of(windowSize, (int)requiredTests, repeatSize);
		std::vector<std::string> heads = tree_sequences(left, -leftDistance, margin, false, 2 * opt.branching);
		std::vector<std::string> tails = tree_sequences(right, -rightDistance, margin, true, 2 * opt.branching);
		long combinations = (long)heads.size() * (long)tails.size();
		if (combinations > (long)opt.branching * opt.branching) {
			random_shuffle(heads);
			random_shuffle(tails);
			const size_t br = (size_t)opt.branching;
			if (heads.size() > br && tails.size() > br) { heads.resize(br); tails.resize(br); }
			else if (tails.size() <= br) { const size_t n = br * br / tails.size(); if (n < heads.size()) heads.resize(n); }
			else { const size_t n = br * br / heads.size(); if (n < tails.size()) tails.resize(n); }
		}
		for (const auto& h : heads) if ((long)h.size() < margin) return decide(HEAD_SHORTER_THAN_MARGIN);
		for (const auto& t : tails) if ((long)t.size() < margin) return decide(TAIL_SHORTER_THAN_MARGIN);
		p.q0 = q.size();
		for (const auto& head : heads) {
			for (const auto& tail : tails) {
				// testCombination, :369-417
				const int planned = (int)std::max<long>(requiredTests, opt.minTests);
				const int possible = (int)(head.size() + rep.size() + tail.size()) - windowSize + 1;
				UnknownReason why = UNDETERMINED;
				bool unk = true;
				int m2 = 0;
				if (possible < planned) why = POSSIBLE_TESTS_LT_PLANNED;
				else if (planned > opt.maxTests) why = OVER_MAX_TESTS;
				else {
					m2 = margin_of(windowSize, planned, repeatSize);
					if ((long)head.size() < m2) why = HEAD_SHORTER_THAN_MARGIN;
					else if ((long)tail.size() < m2) why = TAIL_SHORTER_THAN_MARGIN;
					else unk = false;
				}
				if (unk) {
					p.endsUnknown = true;
					p.support = Support::unknown_of(why);
					p.q1 = q.size();
					return p;
				}
				if (possible > planned + 1) { q.seqs.append(head, head.size() - m2, m2); q.seqs += rep; q.seqs.append(tail, 0, m2); }
				else { q.seqs += head; q.seqs += rep; q.seqs += tail; }
				q.off.push_back(q.seqs.size());
			}
		}
		p.q1 = q.size();
		return p;
	}
	Support fold(const Pending& p, const Queries& q) const // the loop of determinePathSupport, :574-604
	{
		if (p.decided) return p.support;
		Support best = Support::unknown_of(p.calculatedTests, UNDETERMINED);
		for (size_t i = p.q0; i < p.q1; i++) {
			const int len = (int)(q.off[i + 1] - q.off[i]);
			const Support s = len >= (int)r ? Support::known((int)q.found[i], len - (int)r + 1) : Support::known(0, 0);
			if (s.found > best.found) best = s;
			else if (best.found == 0 && s.tests > best.tests) best.tests = s.tests;
		}
		if (p.endsUnknown) best = p.support;
		best.calculatedTests = (int8_t)p.calculatedTests;
		return best;
	}

	bool is_small_repeat(V node) const // :678-687
	{
		return !g.removed(node) && !(node & 1) && window_long_enough((int)r, opt.minTests, contig_size(node), MIN_MARGIN) &&
		       (g.in_degree(node) > 0 && g.out_degree(node) > 0) && (g.in_degree(node) > 1 || g.out_degree(node) > 1);
	}

	Resolution resolve_repeats() // :689-834
	{
		const long total = (long)(g.nv() - g.num_removed()) / 2;
		long repeats = 0;
		Resolution res;
		res.r = (int)r;
		progress_start("Path resolution (r = " + std::to_string(r) + ")", (unsigned)(total * 2));
		// pass 1: the paths of every small repeat, their combinations listed.  The order only matters to the rand() stream of
		// random_shuffle, and there the reference's -j1 order is what libgomp makes of iteratorMultithreading (RUtils.h:19-63):
		// per region of MAX_SIMULTANEOUS_TASKS = 60000 vertices the first 65 tasks are queued (GOMP_task defers a task while
		// the team holds at most 64 per thread) and run at the region's end in creation order; every later one runs on the spot.
		std::vector<Pending> pend;
		Queries q;
		const uint64_t REGION = 60000, DEFERRED = 65;
		for (uint64_t base = 0; base < g.nv(); base += REGION) {
			std::vector<V> small;
			for (uint64_t ui = base; ui < std::min<uint64_t>(g.nv(), base + REGION); ui++) {
				const V node = (V)ui;
				if (g.removed(node)) continue;
				if (is_small_repeat(node)) small.push_back(node); else progress_update();
			}
			repeats += (long)small.size();
			std::rotate(small.begin(), small.begin() + std::min<size_t>(DEFERRED, small.size()), small.end());
			for (const V node : small)
				for (const Edge& in : g.adj[node ^ 1]) {
					const V intig = in.v ^ 1;
					for (const Edge& out : g.adj[node]) pend.push_back(plan_path(intig, node, out.v, q));
				}
		}
		// the filter's answers, in one batch
		q.found.assign(q.size(), 0);
		if (q.size()) filter.contains(q.seqs.data(), q.off.data(), q.size(), q.found.data());
		// pass 2: supports per repeat (buildRepeatSupportMap, :607-645) and the statistics (updateStats, :647-676)
		std::vector<Support> supports;
		for (size_t i = 0; i < pend.size();) {
			const unsigned repeat = pend[i].repeat;
			SupportMap sm;
			bool unknown = false;
			size_t j = i;
			for (; j < pend.size() && pend[j].repeat == repeat; j++) {
				const Support s = fold(pend[j], q);
				sm[pend[j].intig][pend[j].outig] = s;
				unknown |= s.unknown();
			}
			if (unknown)
				for (auto& a : sm) for (auto& b : a.second) if (!b.second.unknown()) { b.second.reset(); b.second.reason = DIFFERENT_CULPRIT; }
			for (const auto& a : sm) for (const auto& b : a.second) {
				const Support& s = b.second;
				supports.push_back(s);
				if (!s.unknown()) {
					res.findsHistogram[s.found]++;
					res.fractionFindsHistogram[int(double(s.found) / double(s.tests) * 100)]++;
				}
				res.calculatedTestsHistogram[s.calculatedTests]++;
			}
			res.repeatSupportMap[repeat] = std::move(sm);
			progress_update();
			i = j;
		}
		long pathsKnown = 0, pathsUnknown = 0, pathsSupported = 0, pathsUnsupported = 0;
		static const char* labels[] = { "Undetermined", "Too many combinations", "Over max tests", "Possible tests < planned tests",
			"Window not long enough", "Head shorter than margin", "Tail shorter than margin", "Different culprit" };
		int reasonCounts[8] = { 0 };
		for (const auto& s : supports) { if (s.unknown()) { pathsUnknown++; reasonCounts[s.reason]++; } else pathsKnown++; }
		const long pathsTotal = pathsKnown + pathsUnknown;
		auto pct = [](long num, long denom) { return denom == 0 ? 0.0 : 100.0 * double(num) / double(denom); };
		auto print_counts = [&]() {
			fprintf(stderr, "Small repeats = %ld/%ld (%f%%)\n", repeats, total, pct(repeats, total));
			fprintf(stderr, "Known support paths = %ld / %ld (%f%%)\n", pathsKnown, pathsTotal, pct(pathsKnown, pathsTotal));
			fprintf(stderr, "Unknown support paths = %ld / %ld (%f%%)\n", pathsUnknown, pathsTotal, pct(pathsUnknown, pathsTotal));
			for (int i = 0; i < 8; i++) fprintf(stderr, "%s%s: %f%%", i ? ", " : "", labels[i], pct(reasonCounts[i], pathsUnknown));
			fprintf(stderr, "\n");
		};
		if (repeats > 0 && pathsKnown > 0) {
			for (const auto& kv : res.findsHistogram) { if (kv.first >= opt.threshold) pathsSupported += (long)kv.second; else pathsUnsupported += (long)kv.second; }
			const double sampleFactor = double(pathsKnown) / double(pathsSupported + pathsUnsupported);
			pathsSupported = (long)(pathsSupported * sampleFactor);
			pathsUnsupported = (long)(pathsUnsupported * sampleFactor);
			if (opt.verbose) {
				print_counts();
				fprintf(stderr, "Supported paths ~= %ld/%ld (%f%%)\n", pathsSupported, pathsKnown, pct(pathsSupported, pathsKnown));
				fprintf(stderr, "Unsupported paths ~= %ld/%ld (%f%%)\n", pathsUnsupported, pathsKnown, pct(pathsUnsupported, pathsKnown));
			}
			if (double(pathsSupported) / double(pathsKnown) < SUPPORTED_PATHS_MIN) {
				fprintf(stderr, "Insufficient support found. Is something wrong with the data?\n");
				res.failed = true;
			}
		} else {
			fprintf(stderr, "No small resolveable junctions were found!\n");
			if (opt.verbose) print_counts();
			res.failed = true;
		}
		return res;
	}

	// ---- processGraph, :871-1229 ----
	struct RepeatInstance {
		V instance, original;
		std::vector<V> originalIntigs, originalOutigs;
		bool in_intigs(V n) const { return std::find(originalIntigs.begin(), originalIntigs.end(), n) != originalIntigs.end(); }
		bool in_outigs(V n) const { return std::find(originalOutigs.begin(), originalOutigs.end(), n) != originalOutigs.end(); }
		RepeatInstance reverse() const
		{
			RepeatInstance x{ instance ^ 1, original ^ 1, {}, {} };
			for (V o : originalOutigs) x.originalIntigs.push_back(o ^ 1);
			for (V i : originalIntigs) x.originalOutigs.push_back(i ^ 1);
			return x;
		}
	};
	static bool good(const Support& s, int threshold) { return s.unknown() || s.found >= threshold; }
	void process_graph(const Resolution& res)
	{
		progress_start("New paths and vertices setup", (unsigned)(res.repeatSupportMap.size() * 3));
		struct OldEdge { V u, v; };
		struct NewEdge { V u, v; int d; };
		struct NewVertex { V original, node; };
		std::vector<OldEdge> edges2remove;
		std::vector<NewEdge> edges2add;
		std::vector<NewVertex> vertices2add;
		std::map<int, std::vector<RepeatInstance>> instances;
		size_t lastId = g.nv() / 2;
		// 1: the paths by their support
		for (const auto& rs : res.repeatSupportMap) {
			const V repeat = rs.first;
			instances.emplace((int)repeat, std::vector<RepeatInstance>());
			instances.emplace((int)(repeat ^ 1), std::vector<RepeatInstance>());
			for (const auto& a : rs.second) {
				const V intig = a.first;
				for (const auto& b : a.second) {
					const V outig = b.first;
					const ImaginaryContigPath path = { { intig, 0 }, { repeat, g.dist(intig, repeat) }, { outig, g.dist(repeat, outig) } };
					if (good(b.second, opt.threshold)) supportedPaths.insert(path);
					else { unsupportedPaths.insert(path); supportedPaths.erase(path); }
				}
			}
			progress_update();
		}
		// 2: the in-neighbours of a repeat grouped by their set of supported out-neighbours: one instance of the repeat per group
		for (const auto& rs : res.repeatSupportMap) {
			const V repeat = rs.first;
			auto& fwd = instances.at((int)repeat);
			auto& rev = instances.at((int)(repeat ^ 1));
			for (const auto& a : rs.second) {
				const V intig = a.first;
				std::vector<V> supportedOutigs;
				for (const auto& b : a.second) if (good(b.second, opt.threshold)) supportedOutigs.push_back(b.first);
				bool matched = false;
				for (auto& inst : fwd) {
					if (inst.originalOutigs.size() == supportedOutigs.size()) {
						matched = true;
						for (V o : supportedOutigs) if (!inst.in_outigs(o)) { matched = false; break; }
					}
					if (matched) { inst.originalIntigs.push_back(intig); break; }
				}
				if (!matched && !supportedOutigs.empty()) {
					if (fwd.empty()) fwd.push_back(RepeatInstance{ repeat, repeat, { intig }, supportedOutigs });
					else fwd.push_back(RepeatInstance{ (V)(2 * lastId++) + (repeat & 1), repeat, { intig }, supportedOutigs });
				}
			}
			if (!fwd.empty()) for (const auto& inst : fwd) rev.push_back(inst.reverse());
			else {
				fwd.push_back(RepeatInstance{ repeat, repeat, {}, {} });
				rev.push_back(fwd.back().reverse());
			}
			progress_update();
		}
		// 3: what to remove and what to add
		for (const auto& rs : res.repeatSupportMap) {
			const V repeat = rs.first;
			for (const auto& inst : instances.at((int)repeat)) {
				std::vector<std::pair<V, V>> intigInst, outigInst; // (instance, original)
				for (V intig : inst.originalIntigs) {
					auto it = instances.find((int)intig);
					if (it != instances.end()) { for (const auto& x : it->second) if (x.in_outigs(repeat)) intigInst.push_back({ x.instance, x.original }); }
					else intigInst.push_back({ intig, intig });
				}
				for (V outig : inst.originalOutigs) {
					auto it = instances.find((int)outig);
					if (it != instances.end()) { for (const auto& x : it->second) if (x.in_intigs(repeat)) outigInst.push_back({ x.instance, x.original }); }
					else outigInst.push_back({ outig, outig });
				}
				if (inst.instance == inst.original) {
					for (const Edge& e : g.adj[inst.original ^ 1]) edges2remove.push_back({ e.v ^ 1, inst.original });
					for (const Edge& e : g.adj[inst.original]) edges2remove.push_back({ inst.original, e.v });
				} else vertices2add.push_back({ inst.original, inst.instance });
				for (const auto& x : intigInst) edges2add.push_back({ x.first, inst.instance, g.dist(x.second, inst.original) });
				for (const auto& x : outigInst) edges2add.push_back({ inst.instance, x.first, g.dist(inst.original, x.second) });
			}
			progress_update();
		}
		std::sort(vertices2add.begin(), vertices2add.end(), [](const NewVertex& a, const NewVertex& b) { return a.node < b.node; });
		std::sort(edges2add.begin(), edges2add.end(), [](const NewEdge& a, const NewEdge& b) { return a.u < b.u || (a.u == b.u && a.v < b.v); });
		progress_start("Graph modification", (unsigned)(edges2remove.size() + vertices2add.size() + edges2add.size()));
		for (const auto& e : edges2remove) { if (g.has_edge(e.u, e.v)) g.remove_edge(e.u, e.v); progress_update(); }
		for (const auto& nvx : vertices2add) {
			if (seqs.size() != nvx.node || comments.size() != (nvx.node >> 1)) die("error: internal: new vertices out of order");
			seqs.push_back(seqs[nvx.original]);
			seqs.push_back(seqs[nvx.original ^ 1]);
			g.put_name(nvx.node, g.create_name());
			g.add_vertex(g.length[nvx.original], g.coverage[nvx.original]);
			comments.push_back(comments[nvx.original >> 1]);
			progress_update();
		}
		for (const auto& e : edges2add) { if (!g.has_edge(e.u, e.v)) g.add_edge(e.u, e.v, e.d); progress_update(); }
	}

	void write_histograms(const Resolution& res, int subiteration) // :1231-1257
	{
		if (opt.verbose) { fprintf(stderr, "Writing algorithm histograms..."); fflush(stderr); }
		const std::string stem = opt.histPrefix + "-r" + std::to_string(res.r) + "-" + std::to_string(subiteration + 1);
		auto write = [&](const std::string& path, const Histogram& h, bool fraction) {
			std::ofstream f(path.c_str());
			for (const auto& kv : h) f << kv.first << '\t' << kv.second << '\n';
			if (fraction && (h.empty() || h.rbegin()->first != 100)) f << 100 << "\t0\n"; // FractionHistogram, :42-49
		};
		write(stem + "-finds.tsv", res.findsHistogram, false);
		write(stem + "-percent-finds.tsv", res.fractionFindsHistogram, true);
		write(stem + "-calculated-tests.tsv", res.calculatedTestsHistogram, false);
		if (opt.verbose) fprintf(stderr, " Done!\n");
	}

	void resolve_short() // :1259-1323
	{
		if (!determine_read_stats()) return;
		tmark("read statistics");
		if (opt.verbose) fprintf(stderr, "\nRunning resolution algorithm...\n");
		for (size_t bi = 0; bi < readSizes.size(); bi++) {
			const ReadSize batch = readSizes[bi];
			current = batch;
			for (int rv : batch.rValues) {
				if (rv < (int)opt.k) { fprintf(stderr, "r value %d(%d) is too short - skipping.\n", rv, current.size); continue; }
				if (opt.verbose) fprintf(stderr, "\nRead size = %d, r = %d ...\n\n", batch.size, rv);
				build_filters(rv, (size_t)(opt.bfMemFactor * double(opt.bloomSize)));
				tmark(" filter built");
				for (int j = 0; j < MAX_SUBITERATIONS; j++) {
					if (opt.verbose) fprintf(stderr, "\nSubiteration %d...\n", j + 1);
					const size_t before = unsupportedPaths.size();
					const Resolution res = resolve_repeats();
					if (!res.failed) {
						process_graph(res);
						assemble_contigs();
						if (!opt.histPrefix.empty()) write_histograms(res, j);
					}
					tmark(" subiteration");
					if (unsupportedPaths.size() == before) break;
				}
			}
		}
		if (opt.verbose) fprintf(stderr, "Resolution algorithm done.\n\n");
	}

	void write_paths(const std::string& path, const ImaginaryContigPaths& paths, const char* what)
	{
		if (opt.verbose) fprintf(stderr, "Writing %s paths to `%s'...\n", what, path.c_str());
		std::ofstream f(path);
		int counter = 0;
		for (const auto& p : paths) f << '>' << counter++ << '\n' << path_sequence(p) << '\n';
		if (!f.good()) { fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno)); exit(EXIT_FAILURE); }
		if (opt.verbose) fprintf(stderr, "%c%s paths written.\n", toupper(what[0]), what + 1);
	}

	std::future<bool> contigs_ahead_; std::vector<std::vector<abghost::FastaRecord>> contigs_parts_;
	bool timing_ = false; double tl_ = 0;
	static double tnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
	void tmark(const char* what) { if (timing_) { const double t = tnow(); fprintf(stderr, "[host] %-28s %.3f s\n", what, t - tl_); tl_ = t; } }
	int run() // main after the options, RResolverShort.cpp:378-402
	{
		// (ABG_RR_TIMING: where the run's time goes, on stderr)
		timing_ = getenv("ABG_RR_TIMING") != nullptr;
		tl_ = tnow();
		const auto mark = [&](const char* what) { tmark(what); };
		// (the contigs' file is parsed -- nothing more: which records the graph knows is found out afterwards, in file order -- while
		// the graph is read; a graph file that cannot be opened is load_graph's to report first)
		if (opt.threads > 1 && opt.graphPath != "-" && access(opt.graphPath.c_str(), R_OK) == 0 && !(getenv("ABG_RR_SERIAL_CONTIGS") && atoi(getenv("ABG_RR_SERIAL_CONTIGS")))) {
			// (exit() -- a graph file with something wrong in it -- waits for that thread: no static destructors beside it)
			static std::future<bool>* ahead = nullptr;
			if (!ahead) { ahead = &contigs_ahead_; atexit([]() { if (ahead && ahead->valid()) ahead->wait(); }); }
			contigs_ahead_ = std::async(std::launch::async, [this]() {
				abghost::ReaderOptions ro;
				ro.foldCase = 0; // (as load_contigs)
				return abghost::read_fasta_blocks(opt.contigsPath, ro, (unsigned)std::max(1, opt.threads), contigs_parts_,
				    [](abghost::FastaRecord& r) { r.aux = reverse_complement(r.seq); });
			});
		}
		load_graph();
		mark("graph read");
		load_contigs();
		mark("contigs read");
		resolve_short();
		mark("resolved");
		if (opt.verbose) { fprintf(stderr, "Stats after resolution:\n"); print_graph_stats(stderr, g); }
		store_contigs(opt.outputContigsPath);
		store_graph(opt.outputGraphPath);
		mark("outputs written");
		if (!opt.outputSupportedPathsPath.empty()) write_paths(opt.outputSupportedPathsPath, supportedPaths, "supported");
		if (!opt.outputUnsupportedPathsPath.empty()) write_paths(opt.outputUnsupportedPathsPath, unsupportedPaths, "unsupported");
		return EXIT_SUCCESS;
	}
};

} // namespace abgrr
