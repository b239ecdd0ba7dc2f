/*
 * abyss_amd.h -- C ABI of the MI355X-native Bloom-filter de Bruijn graph unitig stage.
 *
 * The reference (bcgsc/abyss 2.3.10) has no FFI seam for this path: abyss-bloom-dbg is
 * header-template C++ instantiated into one binary.  The seam it
 * does have is (1) the process boundary -- the abyss-bloom-dbg command line, FASTA and
 * Bloom file formats -- served by abyss_amd/bin/abyss-bloom-dbg, and (2) the in-process
 * "bloom concept" + assembly entry points that BloomDBG/bloom-dbg.h is templated over.
 * This header is the batch-granular replacement of (2): each entry point names the
 * reference interface it stands in for.  Plain pointers and sizes only; the library owns
 * all device memory; the caller owns every host buffer.  One abg_ctx is not thread-safe;
 * distinct contexts may be used from distinct threads.
 *
 * Every function returns ABG_OK (0) or a negative ABG_E* code; abg_last_error() gives the
 * message.  The reference reports errors by printing and exit(EXIT_FAILURE)
 * (Common/IOUtil.h:14-22); the host binary maps non-zero codes to that behaviour.
 * The library itself never exits or aborts: whatever fails inside it -- a device allocation, a
 * capacity the data exceeds, a collective, an internal invariant -- comes back as ABG_ENOMEM /
 * ABG_EINTERNAL.  After one of those two a context is only good for abg_destroy() (the one exception is
 * spelt out at abg_keep_reads: a store of kept reads that cannot grow is dropped without failing anything).
 * (Environment, for tests: ABG_MEM_LIMIT_MB gives every context created afterwards a device
 * memory budget; a request beyond it fails like a hipMalloc that found no room.)
 *
 * The implementation is HIP for gfx950 only: abg_create() fails with ABG_ENODEV when no
 * GPU is present.  There is no CPU fallback.
 */
#ifndef ABYSS_AMD_H
#define ABYSS_AMD_H 1

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABG_OK 0
#define ABG_EINVAL (-1)   /* bad argument / unsupported parameter combination */
#define ABG_ENODEV (-2)   /* no usable HIP device */
#define ABG_ENOMEM (-3)   /* device or host allocation failed */
#define ABG_EINTERNAL (-4)
#define ABG_EAGAIN (-5)   /* abg_assemble_kept: the store of kept reads was dropped; nothing was assembled, the context is intact */

#define ABG_MAX_KMER 192  /* configure.ac:151 (MAX_KMER) */
#define ABG_MAX_HASHES 32 /* configure.ac:156 (MAX_HASHES) */

/* ReadResult, BloomDBG/bloom-dbg.h:256-266 */
enum abg_read_result {
	ABG_RR_UNINITIALIZED = 0,
	ABG_RR_SHORTER_THAN_K,
	ABG_RR_NON_ACGT,
	ABG_RR_BLUNT_END,
	ABG_RR_NOT_SOLID,
	ABG_RR_ALL_KMERS_VISITED,
	ABG_RR_ALL_BRANCH_KMERS_VISITED,
	ABG_RR_GENERATED_CONTIGS
};

/* PathExtensionResultCode, Graph/ExtendPath.h:46-57 */
enum abg_ext_code {
	ABG_ER_AMBI_IN = 0,
	ABG_ER_AMBI_OUT,
	ABG_ER_DEAD_END,
	ABG_ER_CYCLE,
	ABG_ER_LENGTH_LIMIT
};

/* AssemblyParams, BloomDBG/AssemblyParams.h:13-85 (the fields the path uses) */
typedef struct abg_params {
	uint32_t k;          /* -k  k-mer size, 2..ABG_MAX_KMER */
	uint32_t num_hashes; /* -H  Bloom hash functions [4] */
	uint32_t min_cov;    /* --kc minimum k-mer count [2] */
	uint32_t trim;       /* -t  max branch length to trim; UINT32_MAX = k (bloom-dbg.cc:507-509) */
	uint64_t bloom_bytes;/* -b  memory budget; counters = roundUp64(round(B/1.125)) (bloom-dbg.cc:365-367) */
	uint64_t counters;   /* if non-zero, use exactly this many counters instead of bloom_bytes (-i, tests) */
	const char* spaced_seed; /* -s / -K / --qr-seed pattern (MaskedKmer::mask(), BloomDBG/MaskedKmer.h:25-48): k
	                          * characters '0'/'1', beginning and ending with '1', symmetric; NULL or "" = none.
	                          * Only read during abg_create. */
	int32_t device;      /* HIP device ordinal */
	int32_t verbose;
	/* tuning; 0 = default */
	uint64_t insert_batch_kmers;
	uint32_t claim_log2;
	uint32_t walk_slots;
	uint32_t wtab_log2;
	/* > 0: build a HashAgnosticCascadingBloom of this many levels instead of the counting filter
	 * (`abyss-bloom build -t rolling-hash -l N`, Bloom/bloom.cc:585-602); `counters` is then the
	 * number of BITS per level.  PASS 2 is not available in this mode. */
	uint32_t cascade_levels;
	/* Partitioned run (abg_attach_comm) with a filter beyond one device: each rank keeps its own range of the counters only, PASS 2
	 * probes the all-gathered bit plane "counter >= min_cov" (B/8 bytes) and sums coverage through an all-reduce.  0 = when the whole
	 * filter would not fit this device, 1 = always, 2 = never.  A context created this way holds no counters until a communicator
	 * is attached; -g, abg_contains_seq and a cascading filter are not available on it. */
	uint32_t slice_filter;
	uint32_t reserved_[5];
} abg_params;

/* AssemblyCounters, BloomDBG/AssemblyCounters.h:15-31 */
typedef struct abg_counters {
	uint64_t solid_reads, visited_reads, reads_processed, bases_assembled, next_contig_id;
} abg_counters;

/* What outputContig hands to printContig and to the -T trace (bloom-dbg.h:186-254,538-620).
 * Pointers are valid only during the callback. */
typedef struct abg_contig {
	uint64_t contig_id;   /* UINT64_MAX when redundant (not printed) */
	uint64_t read_index;  /* index, within this abg_assemble_* call, of the seeding read */
	const char* seq;      /* ACGT, NUL-terminated */
	uint32_t length;
	uint32_t coverage;    /* getSeqAbsoluteKmerCoverage, bloom-dbg.h:92-109 */
	int32_t redundant;
	uint32_t left_ext, right_ext;
	int32_t left_code, right_code; /* abg_ext_code */
	uint32_t seed_pos;    /* read k-mer index of the seed k-mer */
} abg_contig;
typedef void (*abg_contig_cb)(void* user, const abg_contig* contig);

typedef struct abg_ctx abg_ctx;

/* defaults: num_hashes 4, min_cov 2, trim UINT32_MAX (AssemblyParams.h:78-85) */
void abg_params_init(abg_params* p);
/* CountingBloomFilter<uint8_t>(counters, H, k, kc) + BloomFilter(size, H, k)
 * (bloom-dbg.cc:349-369, bloom-dbg.h:909-911) */
int abg_create(const abg_params* p, abg_ctx** out);
void abg_destroy(abg_ctx* ctx);
const char* abg_last_error(const abg_ctx* ctx); /* ctx may be NULL: error of the last failed abg_create */

/* Empty filters, zero counters, empty contigEndKmers: the state right after abg_create, keeping the
 * device memory (what destroying and re-creating the context would do, minus ~26 GB of hipFree /
 * hipMalloc for a 2G filter).  Tuning and parameters are unchanged. */
int abg_reset(abg_ctx* ctx);

/* bloom.size() / sizeInBytes(): number of uint8 counters == number of visited bits */
int abg_filter_size(const abg_ctx* ctx, uint64_t* counters);

/* PASS 1 -- BloomDBG::loadSeq for each sequence in order (BloomIO.h:32-41; loadFile
 * :50-94 at -j1).  Sequences are ASCII, offsets has n + 1 entries; characters are
 * upper-cased and k-mers overlapping a non-ACGT character are skipped exactly like
 * RollingHashIterator (RollingHashIterator.h:35-97). */
int abg_load_seqs(abg_ctx* ctx, const char* seqs, const uint64_t* offsets, uint64_t n);

/* the same over a read set held in `nchunks` buffers, taken one after the other (a reader's blocks as
 * they lie) */
int abg_load_seqs_v(abg_ctx* ctx, uint32_t nchunks, const char* const* seqs, const uint64_t* const* offsets,
    const uint64_t* n);

/* The reference reads its input twice (loadBloomFilter, then assemble: bloom-dbg.cc:519-547).  With
 * abg_keep_reads(ctx, 1, expected_bases) the reads of the abg_load_seqs / abg_load_seqs_v calls that
 * follow stay on the device as PASS 1 packed them (2 bits a base), and abg_assemble_kept runs PASS 2 over
 * all of them as ONE read stream -- what abg_assemble_seqs_v over the same buffers would do, without
 * packing and uploading them a second time.  results (may be NULL) holds one byte per read loaded since
 * abg_keep_reads, abg_contig.read_index counts through them.  expected_bases (0: unknown) sizes the
 * store.  abg_keep_reads fails with ABG_ENOMEM when the reads would take more than an eighth of the
 * device's memory (the context stays usable: nothing is kept, that is all); if the store cannot grow later,
 * loading goes on without it -- PASS 1 is complete and correct -- and abg_assemble_kept returns ABG_EAGAIN
 * before touching anything: the caller reads its input again, as the reference does.  Any OTHER failure of
 * abg_assemble_kept (ABG_ENOMEM inside PASS 2 included) is final as everywhere else.  abg_assemble_kept and
 * abg_keep_reads(ctx, 0, 0) release the store.
 * While reads are kept, a load call returns once its sequences are packed (the buffers may then be
 * reused); the upload and the ordered insert run beside the caller's work on the next chunk, one call's
 * at a time and in call order.  Every other entry point waits for it first, and a failure of that
 * deferred part is returned by the next call on the context, whichever it is. */
int abg_keep_reads(abg_ctx* ctx, int on, uint64_t expected_bases);
int abg_assemble_kept(abg_ctx* ctx, uint8_t* results, abg_contig_cb cb, void* user);

/* PASS 1 on sequences already resident in device memory in the packed layout:
 * 2 bits per base (A,C,G,T = 0..3), 16 bases per uint32 word, every sequence starting on
 * a word boundary.  d_words/d_woff/d_len are device pointers (woff has n + 1 entries);
 * every sequence must be pure ACGT with len >= k. */
int abg_load_packed(abg_ctx* ctx, const uint32_t* d_words, const uint64_t* d_woff,
    const uint32_t* d_len, uint64_t n);

/* popCount() / filtered_popcount() of the counting filter (CountingBloomFilter.hpp:219-242) */
int abg_counting_stats(abg_ctx* ctx, uint64_t* popcount, uint64_t* filtered_popcount);

/* raw arrays, for operator<< / loadFilter (CountingBloomFilter.hpp:262-281,370-379;
 * BloomFilter.hpp:105-114,283-294), checkpoints (Checkpoint.h) and parity tests.
 * counters: abg_filter_size() bytes; visited: abg_filter_size()/8 bytes. */
int abg_counters_export(abg_ctx* ctx, uint8_t* host_out);
int abg_counters_import(abg_ctx* ctx, const uint8_t* host_in);
int abg_visited_export(abg_ctx* ctx, uint8_t* host_out);
int abg_visited_import(abg_ctx* ctx, const uint8_t* host_in);

/* PASS 2 -- processRead for each read in order (bloom-dbg.h:781-882; the batch loop of
 * assemble() :1012-1066 at -j1).  results (may be NULL) receives one abg_read_result per
 * read; cb is invoked once per outputContig call, in the reference's order, redundant
 * contigs included (they appear in the -T trace).  State (visited filter, contigEndKmers,
 * counters) carries over between calls, so a read stream may be fed in chunks.  cb may run on a
 * thread the library made (a batch's contigs are handed over while the device works on the next
 * batch): one call at a time, in order, and none after the assemble call has returned. */
int abg_assemble_seqs(abg_ctx* ctx, const char* seqs, const uint64_t* offsets, uint64_t n,
    uint8_t* results, abg_contig_cb cb, void* user);
/* the same over a read set held in `nchunks` buffers (seqs[c], offsets[c], n[c] as above): ONE pass
 * over all of them -- the reference's assemble() reads its files as one stream (bloom-dbg.h:1012-1066),
 * and one call keeps one walk schedule instead of starting a new one per chunk.  Read indices
 * (results, abg_contig.read_index) count through the chunks in order; results holds sum(n) bytes. */
int abg_assemble_seqs_v(abg_ctx* ctx, uint32_t nchunks, const char* const* seqs, const uint64_t* const* offsets,
    const uint64_t* n, uint8_t* results, abg_contig_cb cb, void* user);
/* the same on device-resident packed reads (pure ACGT, len >= k) */
int abg_assemble_packed(abg_ctx* ctx, const uint32_t* d_words, const uint64_t* d_woff,
    const uint32_t* d_len, uint64_t n, uint8_t* results, abg_contig_cb cb, void* user);

/* one level of the cascading filter (abg_filter_size()/8 bytes); the reference serialises the
 * last one (HashAgnosticCascadingBloom.h:143-150) */
int abg_cascade_export(abg_ctx* ctx, uint32_t level, uint8_t* host_out);

int abg_get_counters(const abg_ctx* ctx, abg_counters* out);
int abg_set_counters(abg_ctx* ctx, const abg_counters* in); /* resume, Checkpoint.h:159-228 */

/* ---- probes used by parity tests and tools ------------------------------------- */
/* ntHash of every valid k-mer of one sequence: positions and num_hashes values each
 * (RollingHashIterator + RollingHash::getHashes, RollingHash.h:141-146), computed on the
 * device.  Returns the number of valid k-mers in *n_out; writes at most cap entries. */
int abg_hash_seq(abg_ctx* ctx, const char* seq, uint64_t len, uint32_t* pos_out,
    uint64_t* hashes_out, uint64_t cap, uint64_t* n_out);

/* goodKmerSet.contains() for every valid k-mer of one sequence, in RollingHashIterator order: the
 * loop of writeCovTrack (-C / -R, bloom-dbg.h:1282-1334).  Writes at most cap entries. */
int abg_contains_seq(abg_ctx* ctx, const char* seq, uint64_t len, uint32_t* pos_out,
    uint8_t* contains_out, uint64_t cap, uint64_t* n_out);

/* ---- multi-GPU: one process per GPU, the filter range-partitioned over the ranks -----------
 * Stands in for what the reference does across machines with MPI messages
 * (Parallel/NetworkSequenceCollection.cpp: k-mers routed to the rank that owns them) and for
 * its filter windows (`abyss-bloom build -w M/N`, Bloom/BloomFilterWindow.h:30-40: a filter
 * split by position).  With a communicator attached, PASS 1 keeps the counting filter
 * range-partitioned by position -- rank q owns counters [q*chunk, (q+1)*chunk), chunk =
 * roundUp64(ceil(size / world)).  The ops of a batch are hashed once (each rank a slice, the
 * slices all-gathered), every rank bins the (op, counter) pairs on the counters it owns into the
 * LDS-sized tiles of its own range and settles them there, two bytes per op through one
 * all_reduce(MAX) carry what an op needs to know from the other ranks, and the few ops left
 * go through reservation rounds with one all_reduce(MIN) of a byte per op and round.  The
 * shards are all-gathered before PASS 2, whose classification and walks are split over the
 * ranks, whose contigs are gathered, and whose ordered commit tests and stamps each rank's own
 * bits (a byte per candidate and per contig through all_reduce).  Results are bit-identical to
 * a single-GPU run over the concatenated read set.
 *
 * The communicator is a table of two collectives over buffers in the library's memory space
 * (device memory): abg_rccl_comm_create() fills it with RCCL (stream-ordered, over xGMI);
 * tests supply their own (e.g. gloo through host staging, abg_dev_copy).  Every rank must make
 * the same sequence of abg_* calls with the same arguments (abg_share_reads excepted, which
 * takes each rank's own reads). */
enum abg_dtype { ABG_U8 = 0, ABG_U32 = 1, ABG_U64 = 2 };
enum abg_redop { ABG_SUM = 0, ABG_MAX = 1, ABG_MIN = 2 };
typedef struct abg_comm {
	int32_t rank, world;
	/* non-zero: the functions enqueue on `stream` (a hipStream_t) and return at once; zero: they
	 * are called with that stream idle and return when the data is in place */
	int32_t stream_ordered;
	/* sizeof(abg_comm) as the caller was compiled with it.  0 (a zeroed field: what callers of the round-3 header pass, whose
	 * struct ended after all_reduce) or anything short of the member: all_to_all_v is not read and counts as NULL. */
	int32_t struct_size;
	void* user;
	/* in place: rank q's part is counts[q] bytes at buf + displs[q]; the caller's own part is
	 * there already; on completion every rank holds every part.  Returns 0 on success. */
	int (*all_gather_v)(void* user, void* buf, const uint64_t* counts, const uint64_t* displs, void* stream);
	/* in place, element-wise over `count` elements of abg_dtype with abg_redop */
	int (*all_reduce)(void* user, void* buf, uint64_t count, int32_t dtype, int32_t op, void* stream);
	/* personalised exchange (may be NULL: the engine then keeps to the two collectives above): send_counts[q]
	 * bytes at send + send_displs[q] go to rank q, recv_counts[q] bytes from rank q arrive at recv +
	 * recv_displs[q]; the two buffers do not overlap.  This is what routes k-mer probes to the ranks that own
	 * their counters (the reference's precedent: Parallel/NetworkSequenceCollection.cpp:1499-1506 computeNodeID
	 * + one MPI message per k-mer operation; here one exchange per batch of operations). */
	int (*all_to_all_v)(void* user, const void* send, const uint64_t* send_counts, const uint64_t* send_displs,
	    void* recv, const uint64_t* recv_counts, const uint64_t* recv_displs, void* stream);
} abg_comm;
#define ABG_MAX_RANKS 16

/* Switch the context to the partitioned run (before any load).  The table is copied; `user`
 * must outlive the context.  world == 1 is allowed (and is the plain single-GPU path).
 * On a sliced filter (abg_params.slice_filter: a rank holds its own range of the counters only) the calls that look at the
 * whole counting filter are COLLECTIVES -- every rank must make them, in the same order, or the run hangs:
 * abg_counters_export, abg_counters_import, abg_counting_stats (and abg_load_* / abg_assemble_* as in every partitioned run).
 * A sliced context that holds counters refuses a communicator with another rank or world (ABG_EINVAL). */
int abg_attach_comm(abg_ctx* ctx, const abg_comm* comm);

/* All-gather of the ranks' packed read sets, in rank order, into device buffers the context
 * owns (valid until the next call or abg_destroy): every rank then passes the returned
 * pointers to abg_load_packed / abg_assemble_packed.  Input layout as abg_load_packed. */
int abg_share_reads(abg_ctx* ctx, const uint32_t* d_words, const uint64_t* d_woff, const uint32_t* d_len,
    uint64_t n_local, const uint32_t** g_words, const uint64_t** g_woff, const uint32_t** g_len,
    uint64_t* n_total);

/* RCCL communicator (librccl.so.1 is opened at run time).  Rank 0 makes the 128-byte id and
 * hands it to the others by whatever channel launched the ranks (bench.py: the
 * torch.distributed store); every rank then creates its communicator on its own device. */
int abg_rccl_unique_id(uint8_t id[128]);
int abg_rccl_comm_create(const uint8_t id[128], int32_t rank, int32_t world, int32_t device, abg_comm* out);
int abg_rccl_comm_destroy(abg_comm* comm);

/* hipMemcpy on the context's stream, synchronous: kind 0 = host to device, 1 = device to host,
 * 2 = device to device (for communicators that stage through the host) */
int abg_dev_copy(abg_ctx* ctx, void* dst, const void* src, uint64_t n, int32_t kind);
/* device memory on the context's GPU for callers that stage packed reads themselves (the *_packed
 * entry points take device pointers); release with abg_dev_free before abg_destroy */
int abg_dev_alloc(abg_ctx* ctx, uint64_t bytes, void** out);
int abg_dev_free(abg_ctx* ctx, void* ptr);

/* -g -- outputGraph (BloomDBG/bloom-dbg.h:1171-1242): for every sequence, trimSeq (:399-451) and a
 * breadth-first search over the solid filter's de Bruijn graph from its first k-mer and from the
 * reverse complement of its last one (Graph/BreadthFirstSearch.h:93-167), printing
 * GraphvizBFSVisitor's lines (:1097-1159): "\tKMER;\n" when a vertex is discovered, "\tU -> V;\n"
 * for every out-edge of a visited vertex.  cb receives that text in chunks, WITHOUT the
 * "digraph g {" / "}" frame; nodes / edges (may be NULL) receive the visitor's counters for this
 * call.  The set of visited vertices carries over between calls (abg_reset clears it), so a read
 * stream may be fed in chunks, as with abg_assemble_seqs. */
typedef void (*abg_text_cb)(void* user, const char* text, uint64_t len);
int abg_output_graph_seqs(abg_ctx* ctx, const char* seqs, const uint64_t* offsets, uint64_t n,
    abg_text_cb cb, void* user, uint64_t* nodes, uint64_t* edges);

/* kernel timing: when enabled, every launch is bracketed by HIP events on the stream it
 * runs on; abg_profile_get reports total milliseconds and launch count of one kernel
 * family ("hash_claim", "insert_round", "insert_retry", "insert_drain", "classify", "read_prep",
 * "walk", "rewalk", "contig_prep", "predict", "precommit", "commit", "pc_count", "pc_stamp",
 * "pc_short", "pc_timemin", "pc_decide", "pc_break", "pc_apply", "pc_write", "popcount"; partitioned run:
 * "insert_apply", "drain_vals", "drain_load", "merge_fix", "comm_all_reduce", "comm_all_gather"). */
int abg_profile_enable(abg_ctx* ctx, int on);
int abg_profile_reset(abg_ctx* ctx);
int abg_profile_get(abg_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);
/* work statistics of the engine since creation */
typedef struct abg_stats {
	uint64_t insert_rounds, walk_rounds, candidates, walked, rewalked, commit_breaks;
	uint64_t commit_rounds; /* passes of the parallel commit (0: the ordered kernel ran) */
	uint64_t generated;     /* candidates whose contigs were committed (parallel commit only) */
	uint64_t bulk_calls, bulk_steps; /* read-guided bulk steps of the walkers: calls that advanced, vertices taken */
	uint64_t lin_steps;     /* unbranched steps taken one at a time */
	uint64_t guide_slots;   /* slots of the guide table of the last abg_assemble_* call (0: none) */
	uint64_t chain_steps;   /* branch-chain vertices settled a read at a time (successor()'s trueBranch chains) */
	uint64_t batch_cuts;    /* PASS-2 batches cut short at the candidate cap */
	uint64_t overflows;     /* rounds restarted because a walker ran out of some capacity */
	uint64_t memo_hits, memo_adds; /* successor() answers taken from / added to the shared memo */
	uint64_t tiled_ops;     /* PASS 1: k-mer ops of batches that went through the LDS tiles ... */
	uint64_t tiled_pending; /* ... of which this many shared a counter with another k-mer and took the reservation rounds */
	uint64_t tile_overflows; /* ... batches whose bins overflowed (judged through a sort of their pairs instead of the tiles; without memory for that, by the reservation rounds as a whole) */
	uint64_t cls_covered_reads; /* PASS 2: reads whose classification took k-mers from the archive of committed contigs instead of probing the filters (until round 6: pre_requests, always 0) */
	uint64_t archive_bases; /* ... bytes of that archive in use (until round 6: pre_adds, always 0) */
	uint64_t cls_decided_reads; /* ... of which this many got their whole verdict there, both look-aheads included (until round 6: cancelled, always 0) */
	uint64_t counter_bytes_held; /* bytes of the counting filter this context holds: all of it, or its own range of a sliced filter (abg_params.slice_filter) */
} abg_stats;
int abg_get_stats(const abg_ctx* ctx, abg_stats* out);

/* ---- the stage after the unitigs: AdjList's k-1 overlap join (AdjList/AdjList.cpp) ------------
 * The reference keeps, per contig i, the vertices i+ (2i) and i- (2i+1), the first k-1 bases of
 * every vertex in `prefixes` and the vertices by their last k-1 bases in an unordered_map
 * (readContigs, AdjList.cpp:192-230); buildOverlapGraph (:233-263) then adds, for v = 0..2n-1 and
 * every u in suffixMap[prefixes[v]] in insertion order, the edge (v^1) -> (u^1) with distance
 * -(k-1), skipping pairs of different sense under --SS.  abg_overlap_join computes exactly those
 * edges on the GPU, as a CSR over the source vertex with every adjacency list in the reference's
 * order; a maintainer replaces the loop of :245-257 by one call and an add_edge per target.
 *
 * head_keys / tail_keys: HOST arrays of n_contigs keys of W = ceil(overlap / 32) uint64 words, the
 * first and the last `overlap` (= k-1) bases of every contig as read (Kmer(seq.substr(...)),
 * :210-211), 2 bits per base (A C G T = 0 1 2 3), base j at bits 2*(j%32) of word j/32.
 * A context owns a HIP stream and its scratch; the result of the last join stays in device memory
 * until the next join or abg_overlap_destroy. */
typedef struct abg_overlap abg_overlap;
int abg_overlap_create(int device, abg_overlap** out);
void abg_overlap_destroy(abg_overlap* o);
const char* abg_overlap_last_error(const abg_overlap* o); /* o may be NULL: the last failed abg_overlap_create */
int abg_overlap_join(abg_overlap* o, uint32_t overlap, uint64_t n_contigs, const uint64_t* head_keys,
    const uint64_t* tail_keys, int strand_specific, uint64_t* n_edges);
/* offsets: 2 * n_contigs + 1 entries (out-edges of vertex s are targets[offsets[s] .. offsets[s+1]));
 * targets: n_edges entries.  Either may be NULL. */
int abg_overlap_edges(abg_overlap* o, uint64_t* offsets, uint32_t* targets);
/* kernel timing as abg_profile_enable / abg_profile_get: "overlap_keys", "sort_pairs",
 * "overlap_count", "scan", "overlap_fill" */
int abg_overlap_profile(abg_overlap* o, int on);
int abg_overlap_profile_get(abg_overlap* o, const char* name, double* total_ms, uint64_t* launches);

/* ---- the stage after AdjList: the read filter of abyss-rresolver-short (RResolver/BloomFilters.cpp) ------------
 * RResolver keeps ONE plain Bloom filter of the reads' r-mers per r value: g_vanillaBloom = btllib::KmerBloomFilter(bytes,
 * HASH_NUM = 7, r) (BloomFilters.h:12, BloomFilters.cpp:246), filled by loadReads (:139-209: for every read of the current
 * read size, insert(seq.substr(0, r + extract - 1))) and asked by testSequence (RAlgorithmsShort.cpp:310-366: found =
 * contains(sequence)).  btllib is not part of the reference tree (configure.ac:268-276 wants it installed); the hash is its
 * published ntHash2 -- forward + reverse strand value, multiply-shift extras, r-mers with a character other than ACGT (either
 * case) skipped -- and the array is btllib's BloomFilter: `bytes` rounded up to a multiple of 8, bit (h % bits) % 8 of byte
 * (h % bits) / 8.  An abg_rr owns a HIP stream, the filter in device memory and its staging buffers; one abg_rr is not
 * thread-safe.  Sequences are ASCII with n + 1 offsets as in abg_load_seqs.  A maintainer replaces the three btllib calls by
 * these (INTEGRATION.md shows the stub). */
typedef struct abg_rr abg_rr;
int abg_rr_create(int device, uint64_t bytes, uint32_t hash_num, uint32_t r, abg_rr** out);
void abg_rr_destroy(abg_rr* f);
const char* abg_rr_last_error(const abg_rr* f); /* f may be NULL: the last failed abg_rr_create */
int abg_rr_bytes(const abg_rr* f, uint64_t* bytes); /* KmerBloomFilter::get_bytes(): the size after rounding */
int abg_rr_clear(abg_rr* f);
/* KmerBloomFilter::insert(seq.substr(0, max_bases)) for every sequence whose length is one of lengths[0 .. n_lengths) (every
 * sequence when n_lengths is 0: the test of BloomFilters.cpp:182) and at least r.  max_bases: 1..4096.  *n_inserted (may be
 * NULL) is incremented by the number of sequences of a wanted length (currentReadCount, :185).  Returns once the caller's buffers
 * have been read; the kernels may still be running (every other call orders itself behind them). */
int abg_rr_insert_seqs(abg_rr* f, const char* seqs, const uint64_t* offsets, uint64_t n, uint32_t max_bases,
    const uint32_t* lengths, uint32_t n_lengths, uint64_t* n_inserted);
/* found[i] = KmerBloomFilter::contains(sequence i): how many of its r-mers the filter holds */
int abg_rr_contains_seqs(abg_rr* f, const char* seqs, const uint64_t* offsets, uint64_t n, uint32_t* found);
int abg_rr_popcount(abg_rr* f, uint64_t* bits_set);  /* get_pop_cnt(): occupancy = bits_set / (8 * bytes), FPR = occupancy ^ hash_num */
int abg_rr_export(abg_rr* f, uint8_t* host_out);      /* the array, abg_rr_bytes() bytes (parity tests) */
int abg_rr_sync(abg_rr* f);                           /* waits for everything queued */
/* kernel timing as abg_profile_enable / abg_profile_get: "rr_insert", "rr_contains", "rr_popcount" */
int abg_rr_profile(abg_rr* f, int on);
int abg_rr_profile_get(abg_rr* f, const char* name, double* total_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif
