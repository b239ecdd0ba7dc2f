/*
 * abg_oracle.h -- CPU restatement of the abyss-bloom-dbg hot path (ABySS 2.3.10).
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (abyss_amd/, include/abyss_amd.h) never links or calls it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against (a) the only known-answer hash vector in the reference tree
 * (vendor/nthash/unittest/UnitTests.cpp:39-53), (b) hash streams and counter
 * arrays produced by the reference's own nthash.hpp / CountingBloomFilter.hpp
 * compiled unmodified into oracle/_ref/tier1, and (c) the unitig FASTA,
 * --read-log and -T trace written by the unmodified reference binary
 * (oracle/_ref/abyss-bloom-dbg -j1) on seeded synthetic read sets, with the
 * resulting fixtures committed under tests/golden/.
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef ABG_ORACLE_H
#define ABG_ORACLE_H 1

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_KMER 192 /* configure.ac:151 */
#define ORC_MAX_HASHES 32 /* configure.ac:156 */

/* ReadResult, BloomDBG/bloom-dbg.h:256-266 */
enum orc_read_result {
	ORC_RR_UNINITIALIZED = 0,
	ORC_RR_SHORTER_THAN_K,
	ORC_RR_NON_ACGT,
	ORC_RR_BLUNT_END,
	ORC_RR_NOT_SOLID,
	ORC_RR_ALL_KMERS_VISITED,
	ORC_RR_ALL_BRANCH_KMERS_VISITED,
	ORC_RR_GENERATED_CONTIGS
};

/* PathExtensionResultCode, Graph/ExtendPath.h:46-57 */
enum orc_ext_code {
	ORC_ER_AMBI_IN = 0,
	ORC_ER_AMBI_OUT,
	ORC_ER_DEAD_END,
	ORC_ER_CYCLE,
	ORC_ER_LENGTH_LIMIT
};

typedef struct orc_ctx orc_ctx;

/* One output contig (what outputContig hands to printContig and the -T trace,
 * bloom-dbg.h:538-620). */
typedef struct {
	uint64_t contig_id; /* UINT64_MAX when redundant */
	uint64_t read_index;
	const char* seq;
	uint32_t length;
	uint32_t coverage;
	int redundant;
	uint32_t left_ext, right_ext; /* extension lengths */
	int left_code, right_code;    /* orc_ext_code */
	const char* seed;             /* seed k-mer */
} orc_contig;

typedef void (*orc_contig_cb)(void* user, const orc_contig* c);

/* counters = roundUp64(round(B / 1.125)), bloom-dbg.cc:365-367 */
uint64_t orc_counters_for_budget(uint64_t bloom_bytes);

/* `counters` is the number of uint8 counters (CountingBloomFilter ctor rounds it
 * up to a multiple of 8, CountingBloomFilter.hpp:40-50); the visited filter has
 * the same number of bits (bloom-dbg.h:910).  mask may be NULL/"" (no spaced seed). */
orc_ctx* orc_create(unsigned k, unsigned num_hashes, unsigned min_cov, unsigned trim,
    uint64_t counters, const char* mask);
void orc_destroy(orc_ctx*);

uint64_t orc_size(const orc_ctx*);           /* number of counters == visited bits */
uint8_t* orc_counters(orc_ctx*);             /* raw counter array (size bytes)     */
uint8_t* orc_visited(orc_ctx*);              /* raw visited bits (size/8 bytes)    */
uint64_t orc_popcount(const orc_ctx*);       /* CountingBloomFilter.hpp:219-229 */
uint64_t orc_filtered_popcount(const orc_ctx*); /* :233-242 */

/* ntHash of one sequence: RollingHashIterator semantics (RollingHashIterator.h:35-97).
 * Writes k-mer start positions and num_hashes values per valid k-mer; returns the
 * number of valid k-mers (may exceed cap; only cap entries are written). */
uint64_t orc_hash_seq(const orc_ctx*, const char* seq, size_t len, uint32_t* pos_out,
    uint64_t* hashes_out, uint64_t cap);

/* PASS 1: loadSeq over each sequence in order (BloomIO.h:32-41). */
void orc_load_seqs(orc_ctx*, const char* seqs, const uint64_t* offsets, uint64_t n);

/* bloom concept on precomputed hashes (n x num_hashes) */
void orc_insert_hashes(orc_ctx*, const uint64_t* hashes, uint64_t n);
void orc_min_count(const orc_ctx*, const uint64_t* hashes, uint64_t n, uint8_t* out);
void orc_visited_insert_hashes(orc_ctx*, const uint64_t* hashes, uint64_t n);
void orc_visited_contains(const orc_ctx*, const uint64_t* hashes, uint64_t n, uint8_t* out);

/* PASS 2: processRead over each read in order (bloom-dbg.h:781-882, 972-1089).
 * results_out (may be NULL) receives one orc_read_result per read. Returns the
 * number of contigs output so far (next contig id). */
uint64_t orc_assemble(orc_ctx*, const char* seqs, const uint64_t* offsets, uint64_t n,
    uint8_t* results_out, orc_contig_cb cb, void* user);

/* AssemblyCounters, AssemblyCounters.h:15-31 */
void orc_counters_get(const orc_ctx*, uint64_t* solid_reads, uint64_t* visited_reads,
    uint64_t* reads_processed, uint64_t* bases_assembled, uint64_t* next_contig_id);

/* graph probes for unit parity (ExtendPath.h): kmer is k ACGT chars.
 * dir: 0 = FORWARD, 1 = REVERSE (Graph/Path.h:37). */
int orc_look_ahead(const orc_ctx*, const char* kmer, int dir, unsigned depth);
/* returns orc_ext_code; succ_out (k+1 bytes) receives the vertex returned by successor() */
int orc_successor(const orc_ctx*, const char* kmer, int dir, unsigned trim, unsigned fp_trim,
    char* succ_out);
/* neighbours present in the solid filter, bit i = BASE_CHARS[i] (RollingBloomDBG.h:302-427) */
unsigned orc_out_mask(const orc_ctx*, const char* kmer);
/* -g: outputGraph (bloom-dbg.h:1171-1242): the lines between "digraph g {" and "}", handed over in chunks */
typedef void (*orc_text_cb)(void* user, const char* text, size_t len);
void orc_output_graph(const orc_ctx*, const char* seqs, const uint64_t* offsets, uint64_t n,
    orc_text_cb cb, void* user, uint64_t* nodes_out, uint64_t* edges_out);
unsigned orc_in_mask(const orc_ctx*, const char* kmer);

/* HashAgnosticCascadingBloom (Bloom/HashAgnosticCascadingBloom.h:26-182) as built by
 * `abyss-bloom build -t rolling-hash -l LEVELS` (Bloom/bloom.cc:585-602): `levels` bit filters
 * of `level_bits` bits (multiple of 64); insert = first level that does not contain the
 * k-mer (:124-133); the last level is what is serialised (:143-150). */
typedef struct orc_cascade orc_cascade;
orc_cascade* orc_cascade_create(unsigned k, unsigned num_hashes, unsigned levels, uint64_t level_bits);
void orc_cascade_destroy(orc_cascade*);
void orc_cascade_load_seqs(orc_cascade*, const char* seqs, const uint64_t* offsets, uint64_t n);
uint8_t* orc_cascade_level(orc_cascade*, unsigned level); /* level_bits / 8 bytes */

/* spaced seeds, BloomDBG/SpacedSeed.h:18-75 (out must hold k+1 bytes) */
void orc_seed_kmer_pair(unsigned k, unsigned K, char* out);
void orc_seed_qr(unsigned len, char* out);
void orc_seed_qr_pair(unsigned k, unsigned qr_len, char* out);

#ifdef __cplusplus
}
#endif
#endif
