// abg_engine.h -- orchestration of the two passes of abyss-bloom-dbg over device
// memory: PASS 1 (BloomIO.h:32-41 loadSeq: ntHash every k-mer, conservative-update
// insert into the uint8 counting filter, in exact read order) and PASS 2
// (bloom-dbg.h:781-882,972-1089: classify reads, walk unitigs, commit contigs in read
// order).  The engine is a template over a Backend that allocates memory and runs
// "for each item" functors; the product instantiates it with the HIP backend only
// (abg_kernels.hip).  tests/hostcheck instantiates it with a serial backend to check
// the logic against the oracle on machines without a GPU.
#pragma once
#include "abg_walk.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <string>
#include <type_traits>
#include <vector>

namespace abg {

// A failure inside the engine or its backend -- out of device memory, a capacity the data exceeds,
// a collective that failed, an invariant that does not hold -- is thrown as a Failure and turned
// into the C ABI's return code and abg_last_error() text at the boundary (abg_kernels.hip: guarded):
// the library never exits the host process (the reference's binaries print and exit,
// Common/IOUtil.h:14-22; that is the host binary's decision here too).  After ABG_ENOMEM /
// ABG_EINTERNAL a context is only good for abg_destroy.
struct Failure { int code; std::string msg; };
constexpr int FAIL_INVAL = -1, FAIL_NOMEM = -3, FAIL_INTERNAL = -4; // == ABG_EINVAL, ABG_ENOMEM, ABG_EINTERNAL (checked in abg_host.h)
[[noreturn]] inline void fail_now(int code, const std::string& msg) { throw Failure{ code, msg }; }
inline std::string strf(const char* fmt, unsigned long long a = 0, unsigned long long b = 0)
{
	char buf[256];
	snprintf(buf, sizeof buf, fmt, a, b);
	return buf;
}

struct Config {
	uint32_t k = 0, nh = 4, kc = 2, trim = 0;
	uint64_t counters = 0;            // number of uint8 counters == visited bits (cascade mode: bits per level)
	uint32_t cascade_levels = 0;      // > 0: HashAgnosticCascadingBloom of that many levels instead of counters
	std::string spaced_seed;          // k characters of '0'/'1', or empty (MaskedKmer::mask())
	// tuning (defaults sized for one MI355X; overridable through abg_params / env)
	uint64_t insert_batch_kmers = 1ull << 24; // k-mer ops per ordered-insert batch (a genome k-mer then recurs ~once per batch at 50x)
	uint32_t claim_log2 = 30;         // PASS 1 claim slots per table (x2 tables, 8 B each: 16 GiB; false conflicts fall with the load)
	uint32_t drain_threshold = 1u << 12; // pending ops at or below which the retry tail runs in one workgroup
	uint64_t compact_threshold = 1u << 20; // rounds with at least this many ops flag their losers and compact them (FInsertRound)
	bool async_load = true;           // abg_load_seqs*: the device's share of a call runs beside the caller's next packing (Session::load_seqs_v)
	bool tiled_insert = true;         // PASS 1 through LDS-sized tiles of the counter array (see TileEnv); else reservation rounds only
	bool benign_sharers = true;       // ... and k-mers that share a counter they cannot write are settled by the tiles as well (op_verdict)
	uint32_t cls_debug_skip = 0;      // FClassify::dbg_skip
	bool cls_archive = true;          // PASS 2: the classification takes the k-mers of a read that lie on a contig committed by an earlier batch from an archive of those contigs instead of probing the filters (ContigArchive)
	uint32_t cls_archive_max_mb = 16384; // ... whose bases and table together stay below this
	bool sorted_overflow = true;      // a batch whose pairs run a bin over is judged and applied through a sort of its pairs (Engine::sorted_judge) instead of taking the reservation rounds as a whole
	bool cosettle = true;             // ... and k-mers that may write shared counters, when every k-mer on those is settled too (op_verdict, FCoSettle)
	uint32_t cosettle_passes = 6;     // passes of that fixed point before the candidates left over go to the rounds after all (1 .. CO_MAX_PASSES)
	uint32_t cosettle_log2 = 24;      // bits of the table of counters the rounds' k-mers touch
	uint32_t walk_slots = 4096;       // concurrent walkers (one per wavefront; 2048 are resident)
	uint32_t tb_cap = 512;            // trueBranch frames per walker beyond the ones that fit in LDS
	uint32_t buf_cap = 1u << 16;      // extension bases per side per walker
	uint64_t pool_cap = 1ull << 28;   // contig pool bytes
	uint32_t rec_cap = 1u << 22;      // contig records per round
	uint32_t wtab_log2 = 26;          // walker vertex table entries (grown per launch to fit its walkers, up to:)
	uint32_t wtab_log2_max = 31;
	uint32_t cend_log2 = 20;          // contigEndKmers table entries
	uint64_t p2_first_batch = 65536;  // PASS 2 read batches grow geometrically from here (a launch is bound by its slowest walker: 8 launches of configs[1] instead of 9, 886 vs 898 ms per step)
	uint64_t p2_max_batch = 1ull << 22;
	uint32_t p2_growth = 2;           // batch i + 1 holds p2_growth times the reads of batch i
	uint32_t p2_crowded = 1u << 18;   // more candidates than this in a batch: halve the next one
	uint32_t p2_starved_growth = 2;   // growth factor after such a batch
	uint32_t p2_starved = 6144;       // fewer candidates than this: the batch was latency-bound, double the next one
	uint32_t slice_filter = 0;        // partitioned run: a rank keeps only its own range of the counters and PASS 2 probes the all-gathered
	                                  // bit plane (B beyond one GPU): 0 = when the whole filter would not fit the device, 1 = always, 2 = never
	bool solid_plane = true;          // PASS 2 probes the bit plane "counter >= kc" instead of the counters (Engine::ensure_plane)
	uint32_t classify_slots = 65536;  // lanes of the classification kernel in flight (each owns 22 KB of lookAhead scratch)
	uint32_t cls_both_max_mb = 1024;  // ... for filters whose two-bit array is at most this large
	bool cls_both = true;             // the classification probes the solid plane and the visited filter in one array of two bits a position (FBothBuild)
	bool prefetch_classify = true;    // classify batch i + 1 on a side stream while batch i's walkers thin out
	uint32_t memo_log2 = 0;           // entries of the successor() memo (0: sized to the filter; see SuccMemo)
	bool memo = true;
	uint32_t p2_max_candidates = 1u << 18; // a batch is cut after this many candidates
	uint32_t t_tags = 1024;           // passes of the parallel commit between two clearings of its time stamps
	uint32_t guide_stride = 4;        // every guide_stride-th read guides the walkers' bulk steps (0: no guide, see Guide).  (8 halves guide_build, 27 -> 14 ms, and gives it back: 5x the unguided steps, rewalk +13 ms; 16: +60 ms)
	uint32_t guide_log2_max = 31;     // at most this many guide slots (8 bytes each)
	bool guide_seen = true;           // the bulk steps keep what they found out about a read's k-mers for the next walker (Guide::seen)
	bool link_duplicates = true;      // the commit decides a contig's copies among a batch's records by their original (Engine::link_duplicates)
	bool par_commit = true;           // parallel fixed-point commit (4 bytes of time stamp per filter bit) ...
	bool overlap_bins = true;              // PASS 1: the next batch is hashed and binned on the side stream while this one is applied
	bool overlap_purity = true;            // ... and its tiles judged there too (tile_purity reads nothing but the bins)
	uint32_t dist_hash_all_ranks = 2;      // partitioned tiles: up to this many ranks, every rank hashes every op itself
	uint32_t dist_route_min_ranks = 4;     // ... from this many ranks on, the (op, counter) pairs are routed to their owners (Engine::insert_tiles_routed); 0: never
	uint64_t keep_insert_scratch_bytes = 16ull << 30; // PASS 1's scratch is given back before PASS 2 when larger than this
	uint64_t par_commit_max_bytes = 16ull << 30; // ... unless that would take more than this: then a stamp per bit the commit touches (hashed)
	int verbose = 0;
};

struct Counters { // AssemblyCounters.h:15-31
	uint64_t solid_reads = 0, visited_reads = 0, reads_processed = 0, bases_assembled = 0,
	         contig_id = 0;
};

// ASCII -> base code 0..3 (A, C, G, T in either case), 0xFF + 3 for anything else: the low two
// bits are what gets packed, bit 7 says "not a base"
struct BaseCodes {
	uint8_t t[256];
	BaseCodes()
	{
		for (int i = 0; i < 256; i++) t[i] = 0xFF;
		t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
	}
};
// Host-side staging of pure-ACGT sequences in the device layout (see Batch).
// (a vector whose resize() leaves new elements uninitialised: the packed words are written right after)
template <class T>
struct NoInitAlloc : std::allocator<T> {
	template <class U> struct rebind { using other = NoInitAlloc<U>; };
	template <class U, class... A> void construct(U* p, A&&... a)
	{
		if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
	}
};
struct HostBatch {
	std::vector<uint32_t, NoInitAlloc<uint32_t>> words;
	std::vector<uint64_t> woff{ 0 };
	std::vector<uint32_t> len;
	std::vector<uint64_t> koff{ 0 };
	void clear() { words.clear(); woff.assign(1, 0); len.clear(); koff.assign(1, 0); }
	uint64_t n() const { return len.size(); }
	// append one sequence given as ASCII; A/C/G/T in either case become 0..3, anything else
	// becomes some base (callers only pass such characters where a spaced seed ignores them);
	// the caller guarantees len >= k
	void add_ascii(const char* s, uint32_t L, uint32_t k)
	{
		static const BaseCodes codes;
		uint64_t w0 = words.size();
		words.resize(w0 + (L + 15) / 16);
		uint32_t* out = words.data() + w0;
		uint32_t i = 0;
		for (; i + 16 <= L; i += 16) {
			uint32_t w = 0;
			for (uint32_t j = 0; j < 16; j++) w |= (uint32_t)(codes.t[(unsigned char)s[i + j]] & 3u) << (2 * j);
			*out++ = w;
		}
		if (i < L) {
			uint32_t w = 0;
			for (uint32_t j = 0; i + j < L; j++) w |= (uint32_t)(codes.t[(unsigned char)s[i + j]] & 3u) << (2 * j);
			*out = w;
		}
		woff.push_back(words.size());
		len.push_back(L);
		koff.push_back(koff.back() + (L - k + 1));
	}
};

struct ContigOut {
	uint64_t contig_id;   // UINT64_MAX if redundant
	uint64_t read_index;  // index of the seeding read within the assemble call
	std::string seq;
	uint32_t coverage;
	bool redundant;
	uint32_t left_ext, right_ext;
	int left_code, right_code;
	uint32_t seed_pos;
};

// ============================================================ PASS 1 functors
// claim value: newer epochs compare lower so stale entries never win
ABG_HD uint64_t claim_val(uint32_t epoch, uint32_t t) { return ((uint64_t)(0xFFFFFFFFu - epoch) << 32) | t; }

struct FHash { // one item per k-mer op: canonical ntHash computed from scratch
	Params p; Batch b; uint64_t* h0;
	ABG_HD void operator()(uint64_t t, uint32_t) const
	{
		uint64_t r = find_seq(b.koff, b.n, t);
		uint32_t j = (uint32_t)(t - b.koff[r]);
		h0[t] = scratch_hash(p, [&](unsigned i) { return batch_base(b, r, j + i); });
	}
};
// Fused PASS 1 front end: each lane owns HC_RUN consecutive k-mer ops, hashes the first from
// scratch (NTF64/NTR64 base forms) and rolls along the read for the rest (NTC64,
// nthash.hpp:242-257,275-279), stores the canonical hashes (64 contiguous bytes per lane) and
// issues the H claims of every op right away, so the hashing ALU work hides under the
// latency of the claim atomics.
#ifndef ABG_HC_RUN
#define ABG_HC_RUN 8
#endif
constexpr uint32_t HC_RUN = ABG_HC_RUN;
// DIST: the filter is range-partitioned over the ranks of a communicator (see Engine::Comm);
// every rank hashes every op but claims only the counters in its own range [lo, lo + span)
// (the partitioned run WITHOUT tiles, ABG_TILED=0; with tiles see Engine::insert_tiles_dist).
template <bool DIST>
struct FHashClaimT {
	Params p; Batch b; uint64_t* h0; uint64_t T; uint64_t* claim; uint64_t cmask; uint32_t epoch;
	uint64_t lo, span;
	uint64_t kbase; // (see FHashOps)
	ABG_HD void claim_all(uint64_t h, uint64_t v) const
	{
		for (unsigned i = 0; i < p.nh; i++) {
			const uint64_t q = pos_i(p, h, i);
			if (!DIST || q - lo < span) atomic_min_u64(&claim[q & cmask], v);
		}
	}
	ABG_HD void operator()(uint64_t g, uint32_t) const
	{
		uint64_t t0 = g * HC_RUN;
		if (t0 >= T) return;
		uint64_t t1 = t0 + HC_RUN < T ? t0 + HC_RUN : T;
		uint64_t r = find_seq(b.koff, b.n, t0 + kbase);
		uint64_t rend = b.koff[r + 1] - kbase;
		unsigned k = p.k;
		uint64_t fh = 0, rh = 0;
		bool fresh = true;
		for (uint64_t t = t0; t < t1; t++) {
			while (t >= rend) { r++; rend = b.koff[r + 1] - kbase; fresh = true; }
			uint32_t j = (uint32_t)(t + kbase - b.koff[r]);
			if (p.mask) {
				// spaced seed: from scratch over the '1' positions (not the headline configuration)
				uint64_t h = scratch_hash(p, [&](unsigned i) { return batch_base(b, r, j + i); });
				h0[t] = h;
				claim_all(h, claim_val(epoch, (uint32_t)t));
				continue;
			}
			if (fresh) {
				fh = 0; rh = 0;
				for (unsigned i = 0; i < k; i++) {
					fh = srol1(fh) ^ seed_of(batch_base(b, r, j + i));
					rh = srol1(rh) ^ seed_of(3u - batch_base(b, r, j + k - 1 - i));
				}
				fresh = false;
			} else {
				unsigned out = batch_base(b, r, j - 1), in = batch_base(b, r, j + k - 1);
				fh = srol1(fh) ^ seed_of(in) ^ p.seed_k[out];
				rh = sror1(rh ^ p.seedrc_k[in] ^ seed_of(3u - out));
			}
			uint64_t h = rh < fh ? rh : fh;
			h0[t] = h;
			claim_all(h, claim_val(epoch, (uint32_t)t));
		}
	}
};
typedef FHashClaimT<false> FHashClaim;
struct FClaim { // first round: every op claims its H counters
	Params p; const uint64_t* h0; uint64_t* claim; uint64_t cmask; uint32_t epoch;
	ABG_HD void operator()(uint64_t t, uint32_t) const
	{
		uint64_t h = h0[t];
		uint64_t v = claim_val(epoch, (uint32_t)t);
		for (unsigned i = 0; i < p.nh; i++)
			atomic_min_u64(&claim[pos_i(p, h, i) & cmask], v);
	}
};
// HashAgnosticCascadingBloom::insert (Bloom/HashAgnosticCascadingBloom.h:124-133): set the
// k-mer's bits in the first level that does not contain it yet.  Called by the one op that
// holds the claims on all of its positions, so no other op of the round reads or writes
// those bits (other bits of the same words are set with atomicOr by their owners).
struct Cascade {
	uint32_t* bits;       // levels x level_words uint32 words; NULL: counting mode
	uint64_t level_words;
	uint32_t levels;
};
ABG_HD void cascade_insert(const Params& p, const Cascade& c, uint64_t h, bool coherent)
{
	for (uint32_t l = 0; l < c.levels; l++) {
		uint32_t* lv = c.bits + (uint64_t)l * c.level_words;
		bool contains = true; // BloomFilter::contains, BloomFilter.hpp:249-259
		for (unsigned j = 0; j < p.nh; j++) {
			uint64_t q = pos_i(p, h, j);
			uint32_t w = coherent ? ld_coherent(&lv[q >> 5]) : lv[q >> 5];
			contains = contains & (((w >> (q & 31)) & 1u) != 0);
		}
		if (!contains) {
			for (unsigned j = 0; j < p.nh; j++) { // BloomFilter::insert, BloomFilter.hpp:182-191
				uint64_t q = pos_i(p, h, j);
				atomic_or_u32(&lv[q >> 5], 1u << (q & 31));
			}
			return;
		}
	}
}

// One round of the deterministic-reservation insert: an op that holds the claim on all
// of its counters is the earliest pending op touching them, so applying it now is what
// the sequential loop of the reference would do (CountingBloomFilter::incrementMin,
// CountingBloomFilter.hpp:135-162); the others re-claim in the other table for the next
// round.  pend == NULL means "all ops 0..n-1".
struct FInsertRound {
	Params p; const uint64_t* h0; uint8_t* cnt; Cascade casc;
	const uint32_t* pend; uint32_t* next; uint32_t* next_n;
	const uint64_t* claim_cur; uint64_t* claim_next; uint64_t cmask; uint32_t epoch;
	uint8_t* lost; // non-NULL: losers are flagged here (and compacted by the caller) instead of appended to `next`
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		uint32_t t = pend ? pend[i] : (uint32_t)i;
		uint64_t h = h0[t];
		uint64_t v = claim_val(epoch, t);
		bool win = true;
		uint64_t pq[4] = { 0, 0, 0, 0 };
		const bool four = p.nh <= 4; // (the usual case: positions computed once, the loads of a phase in flight together)
		if (four) {
			uint64_t cl[4];
#pragma unroll
			for (unsigned j = 0; j < 4; j++) { pq[j] = pos_i(p, h, j < p.nh ? j : 0u); cl[j] = claim_cur[pq[j] & cmask]; }
#pragma unroll
			for (unsigned j = 0; j < 4; j++) win = win & (cl[j] == v);
		} else
		for (unsigned j = 0; j < p.nh; j++)
			win = win & (claim_cur[pos_i(p, h, j) & cmask] == v);
		// losers queue for the next round: one counter bump per wavefront -- all on ONE address, which a
		// round with millions of ops feels; such rounds flag the losers and have them compacted instead
		uint32_t slot = 0;
		if (lost) lost[i] = win ? 0 : 1;
		else slot = wave_append_slot(next_n, !win);
		if (win && casc.bits) {
			cascade_insert(p, casc, h, false);
		} else if (win && four) {
			// (the holder of all its claims is the only op of this round on its counters: what it read is what is there)
			unsigned c[4], mn = 255;
#pragma unroll
			for (unsigned j = 0; j < 4; j++) c[j] = cnt[pq[j]];
#pragma unroll
			for (unsigned j = 0; j < 4; j++) mn = c[j] < mn ? c[j] : mn;
			if (mn < 255) {
#pragma unroll
				for (unsigned j = 0; j < 4; j++) if (j < p.nh && c[j] == mn) cnt[pq[j]] = (uint8_t)(mn + 1);
			}
		} else if (win) {
			unsigned mn = 255;
			for (unsigned j = 0; j < p.nh; j++) { unsigned c = cnt[pos_i(p, h, j)]; mn = c < mn ? c : mn; }
			if (mn < 255)
				for (unsigned j = 0; j < p.nh; j++) {
					uint64_t q = pos_i(p, h, j);
					if (cnt[q] == mn) cnt[q] = (uint8_t)(mn + 1);
				}
		} else {
			if (!lost) next[slot] = t;
			uint64_t v2 = claim_val(epoch + 1, t);
			for (unsigned j = 0; j < p.nh; j++)
				atomic_min_u64(&claim_next[pos_i(p, h, j) & cmask], v2);
		}
	}
};
// ---- the same round when the counters are range-partitioned over R ranks (Engine::Comm).
// Every rank runs every pending op but looks only at the counters it owns: FEvalDist folds
// "holds the claim on all of my counters" and "minimum of my counters" into one byte per op,
//   0 = some claim lost; 1..255 = min(local minimum + 1, 255); 255 also = "none of its counters is mine",
// an all_reduce(MIN) over the ranks turns that into "0 = loser, else min(global minimum + 1, 255)",
// and FApplyDist increments the owned counters equal to the global minimum (incrementMin,
// CountingBloomFilter.hpp:135-162) or re-claims for the next round.  A counter c is bumped iff
// c < 255 and c + 1 == r: for r <= 254 that is c == min; r == 255 leaves c == 254 == min (a
// saturated minimum has no counter below 255 to match).
struct FEvalDist {
	Params p; const uint64_t* h0; const uint8_t* cnt; const uint32_t* pend;
	const uint64_t* claim_cur; uint64_t cmask; uint32_t epoch; uint64_t lo, span; uint8_t* res;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t t = pend ? pend[i] : (uint32_t)i;
		const uint64_t h = h0[t];
		const uint64_t v = claim_val(epoch, t);
		bool held = true, any = false;
		unsigned mn = 255;
		for (unsigned j = 0; j < p.nh; j++) {
			const uint64_t q = pos_i(p, h, j);
			if (q - lo >= span) continue;
			any = true;
			held = held & (claim_cur[q & cmask] == v);
			const unsigned c = cnt[q];
			mn = c < mn ? c : mn;
		}
		res[i] = (uint8_t)(!any ? 255u : !held ? 0u : (mn + 1 > 255 ? 255u : mn + 1));
	}
};
struct FApplyDist {
	Params p; const uint64_t* h0; uint8_t* cnt; const uint32_t* pend;
	uint64_t* claim_next; uint64_t cmask; uint32_t epoch; uint64_t lo, span; const uint8_t* res; uint8_t* lost;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t t = pend ? pend[i] : (uint32_t)i;
		const uint64_t h = h0[t];
		const unsigned r = res[i];
		lost[i] = r == 0;
		if (r == 0) {
			const uint64_t v2 = claim_val(epoch + 1, t);
			for (unsigned j = 0; j < p.nh; j++) {
				const uint64_t q = pos_i(p, h, j);
				if (q - lo < span) atomic_min_u64(&claim_next[q & cmask], v2);
			}
			return;
		}
		for (unsigned j = 0; j < p.nh; j++) {
			const uint64_t q = pos_i(p, h, j);
			if (q - lo >= span) continue;
			const unsigned c = cnt[q];
			if (c < 255 && c + 1 == r) cnt[q] = (uint8_t)(c + 1);
		}
	}
};
// Hand-over to the single-workgroup tail (insert_drain): the few ops still pending touch
// counters on every rank, so their current values are exchanged once (FDrainVals, then an
// all_reduce(MAX): the owner contributes the value, everybody else 0), written over the local
// non-authoritative copies together with the missing claims (FDrainLoad), and every rank runs
// the same drain; only the counters a rank owns keep their meaning afterwards.
struct FDrainVals {
	Params p; const uint64_t* h0; const uint8_t* cnt; const uint32_t* pend; uint64_t lo, span; uint8_t* val;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint64_t h = h0[pend[i]];
		for (unsigned j = 0; j < p.nh; j++) {
			const uint64_t q = pos_i(p, h, j);
			val[i * p.nh + j] = q - lo < span ? cnt[q] : (uint8_t)0;
		}
	}
};
struct FDrainLoad {
	Params p; const uint64_t* h0; uint8_t* cnt; const uint32_t* pend; uint64_t lo, span; const uint8_t* val;
	uint64_t* claim_cur; uint64_t cmask; uint32_t epoch;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t t = pend[i];
		const uint64_t h = h0[t];
		const uint64_t v = claim_val(epoch, t);
		for (unsigned j = 0; j < p.nh; j++) {
			const uint64_t q = pos_i(p, h, j);
			if (q - lo < span) continue;
			cnt[q] = val[i * p.nh + j]; // same value from every op that shares q
			atomic_min_u64(&claim_cur[q & cmask], v);
		}
	}
};
// Tail of the reservation rounds inside ONE workgroup: when few ops are still pending, a
// launch per round is all latency, so the remaining rounds run in a loop here with
// workgroup barriers between the claim and the apply phase.  Claims and counters are read
// through L2 (other lanes of the workgroup wrote them moments ago).
// Sync policy: tid(), nthreads(), barrier(), bcast(uint32_t).
#if defined(__HIP_DEVICE_COMPILE__)
ABG_HD uint8_t ld_coherent_u8(const uint8_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
ABG_HD uint8_t ld_coherent_u8(const uint8_t* p) { return *p; }
#endif
struct InsertDrainEnv {
	Params p; const uint64_t* h0; uint8_t* cnt; Cascade casc;
	uint32_t* list_a; uint32_t* list_b; uint32_t n; // pending ops in list_a
	uint64_t* claim_a; uint64_t* claim_b; uint64_t cmask; uint32_t epoch; // claims of list_a are in claim_a under `epoch`
	uint32_t* counter;   // scratch word; [1] receives the number of rounds run
};
template <class Sync>
ABG_HDN void insert_drain(InsertDrainEnv e, Sync& sy)
{
	const Params& p = e.p;
	const uint32_t tid = sy.tid(), T = sy.nthreads();
	uint32_t n = e.n, rounds = 0;
	uint32_t* cur = e.list_a; uint32_t* nxt = e.list_b;
	uint64_t* ccur = e.claim_a; uint64_t* cnext = e.claim_b;
	uint32_t epoch = e.epoch;
	while (n) {
		if (tid == 0) e.counter[0] = 0;
		sy.barrier();
		for (uint32_t i = tid; i < n; i += T) {
			uint32_t t = ld_coherent(&cur[i]); // written by another lane in the previous round
			uint64_t h = e.h0[t];
			uint64_t v = claim_val(epoch, t);
			bool win = true;
			for (unsigned j = 0; j < p.nh; j++)
				win = win & (ld_coherent(&ccur[pos_i(p, h, j) & e.cmask]) == v);
			if (win && e.casc.bits) {
				cascade_insert(p, e.casc, h, true);
			} else if (win) {
				unsigned mn = 255;
				for (unsigned j = 0; j < p.nh; j++) { unsigned c = ld_coherent_u8(&e.cnt[pos_i(p, h, j)]); mn = c < mn ? c : mn; }
				if (mn < 255)
					for (unsigned j = 0; j < p.nh; j++) {
						uint64_t q = pos_i(p, h, j);
						if (ld_coherent_u8(&e.cnt[q]) == mn) e.cnt[q] = (uint8_t)(mn + 1);
					}
			} else {
				uint32_t slot = atomic_add_u32(&e.counter[0], 1);
				nxt[slot] = t;
				uint64_t v2 = claim_val(epoch + 1, t);
				for (unsigned j = 0; j < p.nh; j++)
					atomic_min_u64(&cnext[pos_i(p, h, j) & e.cmask], v2);
			}
		}
		sy.barrier();
		n = sy.bcast(ld_coherent(&e.counter[0]));
		uint32_t* tl = cur; cur = nxt; nxt = tl;
		uint64_t* tc = ccur; ccur = cnext; cnext = tc;
		epoch++;
		rounds++;
	}
	if (tid == 0) e.counter[1] = rounds;
	sy.barrier();
}
struct FGatherU32 { // out[i] = in[idx[i]]
	const uint32_t* in; const uint32_t* idx; uint32_t* out;
	ABG_HD void operator()(uint64_t i, uint32_t) const { out[i] = in[idx[i]]; }
};
// k-mers per sequence (0 for one shorter than k, which is flagged), ahead of the prefix sum that makes koff
struct FKmerCounts {
	const uint32_t* len; uint32_t k; uint64_t* out; uint32_t* short_flag;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t L = len[i];
		out[i] = L >= k ? (uint64_t)(L - k + 1) : 0;
		if (L < k) *short_flag = 1;
	}
};
// The op ranges of PASS 1's batches, cut on the device (one thread: a greedy walk of binary searches
// over the prefix sums): range r holds the sequences [s, e) -- as many as fit in batch_ops ops, at
// least one -- and their ops [k0, k1).
struct OpRange { uint64_t s, e, k0, k1; };
struct FCutRanges {
	const uint64_t* koff; uint64_t n, batch_ops; OpRange* out; uint32_t cap; uint32_t* count;
	ABG_HD void operator()(uint64_t, uint32_t) const
	{
		uint32_t r = 0;
		for (uint64_t s = 0; s < n;) {
			// largest e in (s, n] with koff[e] - koff[s] <= batch_ops; at least s + 1
			uint64_t lo = s + 1, hi = n;
			const uint64_t lim = koff[s] + batch_ops;
			while (lo < hi) {
				const uint64_t mid = lo + (hi - lo + 1) / 2;
				if (koff[mid] <= lim) lo = mid; else hi = mid - 1;
			}
			const uint64_t e = lo;
			if (r < cap) out[r] = OpRange{ s, e, koff[s], koff[e] };
			r++;
			s = e;
		}
		*count = r;
	}
};
struct FSolidPlane { // bit i of the plane = counter i >= kc; one item per 64 counters (m is a multiple of 64)
	const uint64_t* cnt8; uint32_t kc; uint64_t* plane;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		uint64_t bits = 0;
		for (unsigned w = 0; w < 8; w++) {
			const uint64_t x = cnt8[i * 8 + w];
			for (unsigned b = 0; b < 8; b++) bits |= (uint64_t)(((x >> (8 * b)) & 0xFFu) >= kc ? 1u : 0u) << (8 * w + b);
		}
		plane[i] = bits;
	}
};
// The classification asks every k-mer's H positions two things, "solid?" and "visited?": two sectors a position while the answers live
// in two arrays.  `both` holds them side by side -- bit 2i the solid plane's, bit 2i + 1 the visited filter's -- so the pair is one load
// (FClassify); built from the two at the start of an assemble call, kept up by the commits (both_set) while it runs.
ABG_HD uint64_t spread_bits(uint32_t x) // bit i of x to bit 2i
{
	uint64_t v = x;
	v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
	v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
	v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
	v = (v | (v << 2)) & 0x3333333333333333ull;
	v = (v | (v << 1)) & 0x5555555555555555ull;
	return v;
}
struct FBothBuild { // one item per 32 positions
	const uint32_t* plane32; const uint32_t* vis32; uint64_t* both;
	ABG_HD void operator()(uint64_t i, uint32_t) const { both[i] = spread_bits(plane32[i]) | (spread_bits(vis32[i]) << 1); }
};
ABG_HD void both_set(uint32_t* both32, uint64_t pos) { if (both32) atomic_or_u32(&both32[pos >> 4], 2u << (2u * (uint32_t)(pos & 15u))); }
struct FPopcount { // CountingBloomFilter::popCount / filtered_popcount (hpp:219-242), 8 counters per item
	const uint64_t* cnt8; uint32_t kc; uint64_t* out; // out[0] non-zero, out[1] >= kc
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		uint64_t w = cnt8[i];
		uint32_t nz = 0, ge = 0;
		for (int b = 0; b < 8; b++) { unsigned c = (unsigned)(w >> (8 * b)) & 0xFF; nz += c != 0; ge += c >= kc; }
		if (nz) atomic_add_u64(&out[0], nz);
		if (ge) atomic_add_u64(&out[1], ge);
	}
};

// ============================================================ PASS 1, tiled
// The counter array cut into tiles of TILE_COUNTERS counters that a workgroup holds in LDS.  Per
// batch of ops the (op, counter) pairs are binned by tile; what happens to a counter is then
// decided and applied by the one workgroup that has its tile in LDS, reading and writing the
// filter as a stream of whole tiles instead of one random sector per probe:
//
//   hash      h0[t] for every op (rolling ntHash, 8 ops per lane)                            FHashOps
//   bin       every (op, hash) pair into its tile's bin: [h, t, offset within the tile]      FBinPairs
//   purity    per tile: sort the bin by offset (LDS); a counter all of whose pairs belong to
//             ONE k-mer is pure; the earliest op of the k-mer leads its n ops                tile_purity
//   target    per op: a leader whose k-mer has only pure counters computes what n successive
//             conservative updates leave behind: every counter max(c, min(m + n, 255))       FOpTarget
//   apply     per tile: counters into LDS, max() with the leaders' targets, tile back        tile_apply
//   the rest  ops of k-mers sharing a counter with another k-mer of the batch go through the
//             reservation rounds (FClaim / FInsertRound / insert_drain) in op order
//
// Why that is the sequential result (CountingBloomFilter::incrementMin in read order,
// CountingBloomFilter.hpp:135-162): the ops of one k-mer touch the same H counters.  If no op of
// any OTHER k-mer of the batch touches one of them, those n ops commute with every other op of the
// batch, and n conservative updates in a row turn minimum m into min(m + n, 255) and raise every
// counter below that to it -- one update raises the counters equal to the minimum by one and
// leaves the others, which are above it, alone.  All remaining ops touch counters no such k-mer
// touches, and keep their order among themselves in the reservation rounds.
constexpr int MAX_RANKS = 16;
constexpr uint32_t TILE_BITS = 16, TILE_COUNTERS = 1u << TILE_BITS; // 64 KB of counters in LDS
// The pairs are binned, and judged by tile_purity, in units of a tile or of a half / quarter of one (BIN_COUNTERS); tile_apply takes the
// TILE_SPLIT bins of its tile together.  A batch is sized by the pairs a BIN sees (what tile_purity's table in LDS holds): with two
// bins to the tile a batch is twice as large and the counter array is streamed through tile_apply half as often.
#ifndef ABG_TILE_SPLIT_LOG2
#define ABG_TILE_SPLIT_LOG2 1 // (measured on configs[1]: 0 -> 653 ms a step, 1 -> 634, 2 -> 675-681: notes/README.md)
#endif
constexpr uint32_t BIN_BITS = TILE_BITS - ABG_TILE_SPLIT_LOG2, BIN_COUNTERS = 1u << BIN_BITS, TILE_SPLIT = 1u << ABG_TILE_SPLIT_LOG2;
#ifndef ABG_TILE_PAIRS_LOG2
#define ABG_TILE_PAIRS_LOG2 11 // a batch is sized so that a tile sees 2^this pairs of it on average (12: half as many batches, twice the LDS table; see notes/README.md)
#endif
constexpr uint32_t TILE_PAIRS_MEAN = 1u << ABG_TILE_PAIRS_LOG2;
constexpr uint32_t TILE_SORT_MAX = 3u * TILE_PAIRS_MEAN / 2;        // pairs of one tile per batch, at most (3072)
// An (op, counter) pair as the bins hold it: the op's canonical hash and "op id | hash function << 28" -- 12 bytes.
// The counter is not stored: it is pos_i(h, j), some twenty integer instructions wherever a pair is looked at,
// against four bytes written twice and read three times per pair and batch (the bins are what PASS 1 moves most of).
struct TilePair { uint32_t hlo, hhi, tj; };
constexpr uint32_t TP_T_BITS = 28, TP_T_MASK = (1u << TP_T_BITS) - 1; // (a batch holds at most 2^28 ops, a k-mer at most 16 hash functions)
ABG_HD uint64_t tp_h(const TilePair& r) { return ((uint64_t)r.hhi << 32) | r.hlo; }
ABG_HD uint32_t tp_t(const TilePair& r) { return r.tj & TP_T_MASK; }
ABG_HD uint32_t tp_j(const TilePair& r) { return r.tj >> TP_T_BITS; }
constexpr uint32_t LEAD_BIT = 0x80000000u;                          // TileEnv::lead: this op is the earliest of its k-mer's ops in the batch
struct TileEnv {
	Params p; uint8_t* cnt;
	uint64_t lo, m;                   // the counters [lo, m) are tiled (a rank's own range; the whole filter: 0, m)
	const uint64_t* h0;
	TilePair* bins; uint32_t cap;     // [ntiles][cap]
	uint32_t* tcur;                   // [ntiles] pairs in each bin
	uint32_t* lead;                   // [T] ops n of this op's k-mer in the batch, from any counter only that k-mer touches (0: none known); | LEAD_BIT on the earliest of them
	uint8_t* opflag;                  // [T] bit j: counter j of the op's k-mer is shared with another k-mer of the batch (0xFF: some counter is, nh > 8)
	uint8_t* tgt;                     // [T] leaders: the value their counters are raised to (0: nothing to do)
	uint8_t* pendf;                   // [T] 1: the op goes through the reservation rounds (FOpTarget / FDistTarget)
	uint32_t* flags;                  // [0] a bin overflowed: the batch goes through the reservation rounds instead
	uint32_t benign = 1;              // op_verdict: k-mers that cannot write their shared counters are settled by the tiles too
	uint32_t lead_js = 2;             // tile_purity: pairs of the first lead_js hash functions write `lead` (a partitioned run: all -- every rank must know)
	uint32_t count_max = 0x7FFFFFFFu; // a counter with this many pairs or more is treated like a shared one (a partitioned run: 254 -- the ranks pass a leader's op count in a byte)
	const uint8_t* cval = nullptr;    // partitioned run: [T x nh] 255 - (counter j of op t), combined from the ranks that own them (FDistPack2); NULL: op_verdict reads e.cnt
	// the settling of k-mers that DO write shared counters (op_verdict, FCoSettle); all null / 0 when that is off
	uint32_t* bad = nullptr;          // [(bad_mask + 1) / 32] bit "some k-mer on this counter goes through the rounds", by hashed counter position
	uint32_t bad_mask = 0;
	uint8_t* wmask = nullptr;         // [T] candidates: bit j = the k-mer may raise its shared counter j
	uint32_t* chg = nullptr;          // [CO_MAX_PASSES + 1] chg[i] != 0: pass i took a candidate back
	uint32_t* colist = nullptr;       // [T] the candidates, those of a group of 2^glog2 consecutive ops packed at the start of the group's stretch (CO_* below)
	uint8_t* cocnt = nullptr;         // [T >> glog2] how many of a group's ops are candidates
	uint32_t glog2 = 0;               // 6 where a wavefront runs 64 consecutive ops (FOpTarget), 0 for a serial caller
};
constexpr uint32_t CO_MAX_PASSES = 15;
constexpr uint8_t PEND_NO = 0, PEND_ROUNDS = 1, PEND_CANDIDATE = 2; // TileEnv::pendf
// an entry of TileEnv::colist: the op's place in its group << 26 | CO_MORE | the `bad` slot of the one shared counter it may raise
// (CO_MORE: it may raise several -- TileEnv::wmask says which, the slot field is 0); CO_DEAD once it went to the rounds
constexpr uint32_t CO_LANE_SHIFT = 26, CO_MORE = 1u << 25, CO_SLOT_MASK = CO_MORE - 1, CO_DEAD = 0xFFFFFFFFu;
// canonical hashes of the k-mers [j0, j1) of one sequence: the first from scratch, the rest
// rolled (NTC64, nthash.hpp:242-257,275-279).  Under a spaced seed the rolled state is the UNMASKED
// pair plus the XOR of the masked positions' terms (what maskHash takes out again,
// nthash.hpp:537-547): rolling moves every run of masked positions by one base, so those terms
// are updated per run (masked_terms_shifted, abg_core.h), not per position.  A code outside 0..3
// (the 'N' of a contig column no '1' covers) only ever sits under a '0', where its term cancels.
template <class Get, class Put>
ABG_HD void kmer_hash_run(const Params& p, Get get, uint32_t j0, uint32_t j1, Put put)
{
	if (j0 >= j1) return;
	const unsigned k = p.k;
	if (p.mask) {
		const MaskTab& m = *p.mask;
		uint64_t fh = 0, rh = 0, df = 0, dr = 0;
		for (unsigned i = 0; i < k; i++) {
			fh = srol1(fh) ^ seed_of(get(j0 + i));
			rh = srol1(rh) ^ seed_of(3u - get(j0 + k - 1 - i));
		}
		for (unsigned q = 0; q < m.nmasked; q++) {
			const unsigned pos = m.pos[q], c = get(j0 + pos);
			df ^= srol_n(seed_of(c), k - 1 - pos);
			dr ^= srol_n(seed_of(3u - c), pos);
		}
		for (uint32_t j = j0;; j++) {
			const uint64_t fs = fh ^ df, rs = rh ^ dr;
			put(j, rs < fs ? rs : fs);
			if (j + 1 >= j1) break;
			// to k-mer j + 1: position a of every run [a, b) leaves it, position b enters
			for (unsigned r = 0; r < m.nruns; r++) {
				const unsigned ia = m.run_a[r], ib = m.run_b[r];
				const unsigned xa = get(j + ia), xb = get(j + ib);
				df ^= srol_n(seed_of(xa), k - 1 - ia) ^ srol_n(seed_of(xb), k - 1 - ib);
				dr ^= srol_n(seed_of(3u - xa), ia) ^ srol_n(seed_of(3u - xb), ib);
			}
			df = srol1(df); dr = sror1(dr);
			const unsigned out = get(j), in = get(j + k);
			fh = srol1(fh) ^ seed_of(in) ^ srol_n(seed_of(out), k);
			rh = sror1(rh ^ srol_n(seed_of(3u - in), k) ^ seed_of(3u - out));
		}
		return;
	}
	uint64_t fh, rh;
	scratch_hashes(p, [&](unsigned i) { return get(j0 + i); }, fh, rh);
	put(j0, rh < fh ? rh : fh);
	for (uint32_t j = j0 + 1; j < j1; j++) {
		const unsigned out = get(j - 1), in = get(j + k - 1);
		fh = srol1(fh) ^ seed_of(in) ^ sel4(p.seed_k, out);
		rh = sror1(rh ^ sel4(p.seedrc_k, in) ^ seed_of(3u - out));
		put(j, rh < fh ? rh : fh);
	}
}
template <int NW>
struct FHashOps { // FHashClaim without the claims; the first k-mer of a lane's run comes off the read's words at once
	Params p; Batch b; uint64_t* h0; uint64_t T;
	uint64_t kbase; // b.koff holds k-mer prefix sums of a longer batch: op t of this one is k-mer kbase + t there
	uint64_t tlo;   // the ops [tlo, T) are hashed (partitioned run: each rank hashes a slice of the batch)
	uint32_t run;   // consecutive ops per lane: one k-mer hashed from scratch, the rest rolled (Engine::hash_run_)
	ABG_HD void operator()(uint64_t g, uint32_t) const
	{
		uint64_t t0 = tlo + g * run;
		if (t0 >= T) return;
		uint64_t t1 = t0 + run < T ? t0 + run : T;
		uint64_t r = find_seq(b.koff, b.n, t0 + kbase);
		uint64_t rend = b.koff[r + 1] - kbase;
		unsigned k = p.k;
		uint64_t fh = 0, rh = 0;
		bool fresh = true;
		// (the lanes of a wave write hashes a run apart: every 8-byte store is a write transaction of its own, and 870 M of them are
		// what this kernel took its time for.  An even op's hash waits for its odd neighbour's: one 16-byte store for the two.)
		struct alignas(16) Pair { uint64_t a, b; };
		uint64_t held = 0; bool have = false;
		for (uint64_t t = t0; t < t1; t++) {
			while (t >= rend) { r++; rend = b.koff[r + 1] - kbase; fresh = true; }
			uint32_t j = (uint32_t)(t + kbase - b.koff[r]);
			if (p.mask) {
				// spaced seed: the k-mers of this read among the lane's ops in one rolled run
				const uint64_t tend = t1 < rend ? t1 : rend;
				kmer_hash_run(p, [&](unsigned i) { return batch_base(b, r, i); }, j, j + (uint32_t)(tend - t),
				    [&](uint32_t jj, uint64_t h) { h0[t + (jj - j)] = h; });
				t = tend - 1;
				continue;
			}
			if (fresh) {
				const Kmer<NW> s = window_kmer<NW>(b.words, b.woff[r], j, k);
				kmer_hashes(s, k, fh, rh);
				fresh = false;
			} else {
				unsigned out = batch_base(b, r, j - 1), in = batch_base(b, r, j + k - 1);
				fh = srol1(fh) ^ seed_of(in) ^ p.seed_k[out];
				rh = sror1(rh ^ p.seedrc_k[in] ^ seed_of(3u - out));
			}
			const uint64_t hv = rh < fh ? rh : fh;
			if (!(t & 1) && t + 1 < t1) { held = hv; have = true; }
			else if (have) { *(Pair*)&h0[t - 1] = Pair{ held, hv }; have = false; }
			else h0[t] = hv;
		}
	}
};
// Binning in two passes, each a workgroup-local counting sort, because one global atomic per pair
// on the bins' cursors is what such a kernel then spends its time on (67 M device-scope atomics
// on 29 k addresses: 6 ms per batch).  Pass 1: a workgroup takes BIN_CHUNK_OPS ops, counts its
// pairs per COARSE bin (a run of 2^cshift tiles) in LDS, reserves room in every coarse bin with one
// global atomic, and writes the pairs there in runs.  Pass 2: a workgroup takes BIN_CHUNK_PAIRS
// pairs of one coarse bin and does the same over that bin's tiles.
// (8192 ops a workgroup: its ~70 pairs per coarse bin leave as runs of ~850 bytes -- at 2048 ops the runs were 216 bytes and the
// bins' write traffic 2.2 times their size)
constexpr uint32_t BIN_CHUNK_OPS = 8192, BIN_CHUNK_PAIRS = 4096, BIN_MAX_COARSE = 2048, BIN_MAX_FINE = 4096;
struct BinEnv {
	TileEnv e; uint64_t T;
	TilePair* coarse; uint32_t ccap; uint32_t* ccur; // [ncoarse][ccap], [ncoarse]
	uint32_t cshift, ncoarse;
};
struct FBinCoarse { // item: a chunk of BIN_CHUNK_OPS ops; fast memory: 2 x ncoarse words
	static constexpr uint32_t FAST = 2 * BIN_MAX_COARSE * 4, THREADS = 1024, PER = BIN_CHUNK_OPS / THREADS, MAXH = 4;
	BinEnv b;
	template <class Sync> ABG_HDN void operator()(uint64_t c, void* fast, Sync& sy) const
	{
		uint32_t* hist = (uint32_t*)fast; uint32_t* cur = hist + b.ncoarse;
		const uint32_t tid = sy.tid(), nt = sy.nthreads();
		const uint64_t t0 = c * BIN_CHUNK_OPS, t1 = t0 + BIN_CHUNK_OPS < b.T ? t0 + BIN_CHUNK_OPS : b.T;
		const uint64_t span = b.e.m - b.e.lo;
		const unsigned nh = b.e.p.nh;
		// the coarse bin of counter j of hash h, or ~0 when the counter is another rank's
		auto coarse_of = [&](uint64_t h, unsigned j) -> uint32_t {
			const uint64_t pos = pos_i(b.e.p, h, j) - b.e.lo; // (a counter outside [lo, m) is another rank's)
			return pos < span ? (uint32_t)((pos >> BIN_BITS) >> b.cshift) : 0xFFFFFFFFu;
		};
		auto put = [&](uint64_t h, uint64_t t, unsigned j, uint32_t cb) {
			const uint32_t slot = atomic_add_u32(&cur[cb], 1);
			if (slot >= b.ccap) { b.e.flags[0] = 1; return; }
			TilePair& r = b.coarse[(uint64_t)cb * b.ccap + slot];
			r.hlo = (uint32_t)h; r.hhi = (uint32_t)(h >> 32); r.tj = (uint32_t)t | ((uint32_t)j << TP_T_BITS);
		};
		for (uint32_t i = tid; i < b.ncoarse; i += nt) hist[i] = 0;
		sy.barrier();
		// a thread keeps its ops' hashes and coarse bins in registers between the count and the scatter (up to MAXH hash
		// functions: the positions are some twenty instructions each); a serial caller works them out twice instead
		const bool keep = nt >= THREADS && nh <= MAXH;
		uint64_t hh[PER]; uint32_t cbv[PER][MAXH];
		if (keep) {
#pragma unroll
			for (uint32_t u = 0; u < PER; u++) { // (the loads first, none under a condition: PER round trips become one)
				const uint64_t t = t0 + tid + (uint64_t)u * nt;
				hh[u] = b.e.h0[t < t1 ? t : t0];
			}
#pragma unroll
			for (uint32_t u = 0; u < PER; u++) {
				const uint64_t t = t0 + tid + (uint64_t)u * nt;
				if (t >= t1) continue;
#pragma unroll
				for (unsigned j = 0; j < MAXH; j++) {
					cbv[u][j] = j < nh ? coarse_of(hh[u], j) : 0xFFFFFFFFu;
					if (cbv[u][j] != 0xFFFFFFFFu) atomic_add_u32(&hist[cbv[u][j]], 1);
				}
			}
		} else {
			for (uint64_t t = t0 + tid; t < t1; t += nt) {
				const uint64_t h = b.e.h0[t];
				for (unsigned j = 0; j < nh; j++) { const uint32_t cb = coarse_of(h, j); if (cb != 0xFFFFFFFFu) atomic_add_u32(&hist[cb], 1); }
			}
		}
		sy.barrier();
		for (uint32_t i = tid; i < b.ncoarse; i += nt) cur[i] = hist[i] ? atomic_add_u32(&b.ccur[i], hist[i]) : 0;
		sy.barrier();
		if (keep) {
#pragma unroll
			for (uint32_t u = 0; u < PER; u++) {
				const uint64_t t = t0 + tid + (uint64_t)u * nt;
				if (t >= t1) continue;
#pragma unroll
				for (unsigned j = 0; j < MAXH; j++) if (cbv[u][j] != 0xFFFFFFFFu) put(hh[u], t, j, cbv[u][j]);
			}
		} else {
			for (uint64_t t = t0 + tid; t < t1; t += nt) {
				const uint64_t h = b.e.h0[t];
				for (unsigned j = 0; j < nh; j++) { const uint32_t cb = coarse_of(h, j); if (cb != 0xFFFFFFFFu) put(h, t, j, cb); }
			}
		}
	}
};
struct FBinFine { // item: chunk q of coarse bin cb (item = cb * chunks_per_bin + q); fast memory: 2 x 2^cshift words
	static constexpr uint32_t FAST = 2 * BIN_MAX_FINE * 4, THREADS = 512, PER = BIN_CHUNK_PAIRS / THREADS;
	BinEnv b; uint32_t chunks_per_bin;
	template <class Sync> ABG_HDN void operator()(uint64_t item, void* fast, Sync& sy) const
	{
		const uint32_t cb = (uint32_t)(item / chunks_per_bin), q = (uint32_t)(item % chunks_per_bin);
		const uint32_t filled = ld_coherent(&b.ccur[cb]);
		const uint32_t total = filled < b.ccap ? filled : b.ccap;
		const uint32_t i0 = q * BIN_CHUNK_PAIRS;
		if (i0 >= total) return;
		const uint32_t i1 = i0 + BIN_CHUNK_PAIRS < total ? i0 + BIN_CHUNK_PAIRS : total;
		const uint32_t nfine = 1u << b.cshift;
		uint32_t* hist = (uint32_t*)fast; uint32_t* cur = hist + nfine;
		const uint32_t tid = sy.tid(), nt = sy.nthreads();
		const TilePair* src = b.coarse + (uint64_t)cb * b.ccap;
		// the tile (within the coarse bin) a pair belongs to
		auto fine_of = [&](const TilePair& r) -> uint32_t {
			return (uint32_t)((pos_i(b.e.p, tp_h(r), tp_j(r)) - b.e.lo) >> BIN_BITS) & (nfine - 1);
		};
		for (uint32_t i = tid; i < nfine; i += nt) hist[i] = 0;
		sy.barrier();
		// a thread keeps its pairs (and their tiles) in registers between the count and the scatter: the coarse bin is
		// read once; a serial caller (one thread) reads it twice instead
		const bool keep = nt >= THREADS;
		TilePair mine[PER]; uint32_t fine[PER];
		if (keep) {
#pragma unroll
			for (uint32_t u = 0; u < PER; u++) { // (the loads first, none under a condition)
				const uint32_t i = i0 + tid + u * nt;
				mine[u] = src[i < i1 ? i : i0];
			}
#pragma unroll
			for (uint32_t u = 0; u < PER; u++) {
				const uint32_t i = i0 + tid + u * nt;
				if (i < i1) { fine[u] = fine_of(mine[u]); atomic_add_u32(&hist[fine[u]], 1); }
			}
		} else {
			for (uint32_t i = i0 + tid; i < i1; i += nt) atomic_add_u32(&hist[fine_of(src[i])], 1);
		}
		sy.barrier();
		const uint64_t tile0 = (uint64_t)cb << b.cshift;
		for (uint32_t i = tid; i < nfine; i += nt) cur[i] = hist[i] ? atomic_add_u32(&b.e.tcur[tile0 + i], hist[i]) : 0;
		sy.barrier();
		auto put = [&](const TilePair& r, uint32_t f) {
			const uint32_t slot = atomic_add_u32(&cur[f], 1);
			if (slot >= b.e.cap) { b.e.flags[0] = 1; return; }
			b.e.bins[(tile0 + f) * b.e.cap + slot] = r;
		};
		if (keep) {
#pragma unroll
			for (uint32_t u = 0; u < PER; u++) {
				const uint32_t i = i0 + tid + u * nt;
				if (i < i1) put(mine[u], fine[u]);
			}
		} else {
			for (uint32_t i = i0 + tid; i < i1; i += nt) put(src[i], fine_of(src[i]));
		}
	}
};
// Sync policy of the tile procedures: tid(), nthreads(), barrier(), any(bool).
// tile_purity: the pairs of one tile are grouped by counter in an open-addressing table in fast
// memory (TILE_TAB slots: the counter's offset; the earliest pair on it as op id << 12 | pair
// index; how many pairs).  A counter all of whose pairs carry the hash of its earliest pair's
// k-mer is pure, and that pair's op leads the k-mer's ops.
constexpr uint32_t TILE_TAB = 2u * TILE_PAIRS_MEAN, TILE_IDX_BITS = ABG_TILE_PAIRS_LOG2 + 1; // (4096 slots for at most 3072 pairs: the table is never full; a pair's index in its bin fits TILE_IDX_BITS)
constexpr uint32_t PUR_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t TILE_PURITY_FAST = TILE_TAB * (4 + 8 + 4);
constexpr uint32_t TILE_PURITY_THREADS = ABG_TILE_PAIRS_LOG2 > 11 ? 1024 : 512, TILE_PURITY_PER = (TILE_SORT_MAX + TILE_PURITY_THREADS - 1) / TILE_PURITY_THREADS;
template <class Sync>
ABG_HDN void tile_purity(const TileEnv& e, uint64_t tile, void* fast, Sync& sy)
{
	if (ld_coherent(&e.flags[0])) return; // a bin overflowed: the batch takes the reservation rounds as a whole
	const uint32_t filled = ld_coherent(&e.tcur[tile]);
	const uint32_t n = filled < e.cap ? filled : e.cap;
	if (!n) return;
	const TilePair* bin = e.bins + tile * e.cap;
	const uint32_t tid = sy.tid(), nt = sy.nthreads();
	uint64_t* first = (uint64_t*)fast;                 // [TILE_TAB]
	uint32_t* key = (uint32_t*)(first + TILE_TAB);     // [TILE_TAB] offset within the tile, or PUR_EMPTY
	uint32_t* info = key + TILE_TAB;                   // [TILE_TAB] pairs on the counter | impure << 31
	for (uint32_t i = tid; i < TILE_TAB; i += nt) { key[i] = PUR_EMPTY; first[i] = ~0ULL; info[i] = 0; }
	sy.barrier();
	// a thread keeps its pairs (and the table slot of each one's counter) in registers through the phases;
	// a serial caller (one thread, all pairs) looks everything up again instead
	const bool keep = nt >= TILE_PURITY_THREADS;
	TilePair mine[TILE_PURITY_PER];
	uint32_t slot[TILE_PURITY_PER] = {}; // (table slot | offset within the tile << 16)
	auto off_of = [&](const TilePair& r) -> uint32_t { return (uint32_t)(pos_i(e.p, tp_h(r), tp_j(r)) - e.lo) & (BIN_COUNTERS - 1); };
	auto slot_of = [&](uint32_t off) -> uint32_t {
		uint32_t s = ((off * 0x9E3779B1u) >> 20) & (TILE_TAB - 1);
		for (;;) {
			const uint32_t cur = cas_u32(&key[s], PUR_EMPTY, off);
			if (cur == PUR_EMPTY || cur == off) return s | (off << 16);
			s = (s + 1) & (TILE_TAB - 1);
		}
	};
	auto pairs = [&](auto&& body) { // body(pair, index in the bin, its slot, its offset)
		if (keep) {
#pragma unroll
			for (uint32_t q = 0; q < TILE_PURITY_PER; q++) {
				const uint32_t i = tid + q * nt;
				if (i < n) body(mine[q], i, slot[q] & 0xFFFFu, slot[q] >> 16);
			}
		} else {
			for (uint32_t i = tid; i < n; i += nt) { const uint32_t so = slot_of(off_of(bin[i])); body(bin[i], i, so & 0xFFFFu, so >> 16); }
		}
	};
	if (keep) {
		// (the loads first and none of them under a condition -- a load in a conditional arm is waited for on the spot, and
		// these would be six round trips instead of one: a thread past the end reads pair 0 again and ignores it)
#pragma unroll
		for (uint32_t q = 0; q < TILE_PURITY_PER; q++) {
			const uint32_t i = tid + q * nt;
			mine[q] = bin[i < n ? i : 0u];
		}
#pragma unroll
		for (uint32_t q = 0; q < TILE_PURITY_PER; q++) {
			const uint32_t i = tid + q * nt;
			if (i < n) slot[q] = slot_of(off_of(mine[q]));
		}
	}
	pairs([&](const TilePair& r, uint32_t i, uint32_t s, uint32_t) {
		atomic_min_u64(&first[s], ((uint64_t)tp_t(r) << TILE_IDX_BITS) | i);
		atomic_add_u32(&info[s], 1);
	});
	sy.barrier();
	// another k-mer on the counter -- or the earliest op a second time, through another of its hash functions
	// (about one k-mer in 10^8: its counters then count as shared, which is always safe, and a pure counter
	// holds exactly one pair per op of its k-mer)
	if (keep) {
		TilePair fst[TILE_PURITY_PER]; // (the counters' earliest pairs: again all the loads, then the comparisons)
#pragma unroll
		for (uint32_t q = 0; q < TILE_PURITY_PER; q++) {
			const uint32_t i = tid + q * nt;
			fst[q] = bin[i < n ? (uint32_t)(first[slot[q] & 0xFFFFu] & ((1u << TILE_IDX_BITS) - 1u)) : 0u];
		}
#pragma unroll
		for (uint32_t q = 0; q < TILE_PURITY_PER; q++) {
			const uint32_t i = tid + q * nt;
			const TilePair& r = mine[q]; const TilePair& f = fst[q];
			if (i < n && (r.hlo != f.hlo || r.hhi != f.hhi || ((r.tj ^ f.tj) != 0 && tp_t(r) == tp_t(f)))) atomic_or_u32(&info[slot[q] & 0xFFFFu], 0x80000000u);
		}
	} else
	pairs([&](const TilePair& r, uint32_t, uint32_t s, uint32_t) {
		const TilePair f = bin[(uint32_t)(first[s] & ((1u << TILE_IDX_BITS) - 1u))];
		if (r.hlo != f.hlo || r.hhi != f.hhi || ((r.tj ^ f.tj) != 0 && tp_t(r) == tp_t(f))) atomic_or_u32(&info[s], 0x80000000u);
	});
	sy.barrier();
	pairs([&](const TilePair& r, uint32_t, uint32_t s, uint32_t) {
		const uint32_t inf = info[s], t = tp_t(r);
		// (a partitioned run treats a counter with 254 pairs or more like a shared one: it passes the leaders' op counts
		// between the ranks in a byte.  Elsewhere n ops of a k-mer are n ops however many: min(m + n, 255))
		if ((inf >> 31) || (inf & 0x7FFFFFFFu) >= e.count_max) {
			// shared: bit j of the op's flag byte (several tiles may flag one op at once: a word-wide OR)
			const uint32_t bit = e.p.nh <= 8 ? (1u << tp_j(r)) : 0xFFu;
			atomic_or_u32((uint32_t*)e.opflag + (t >> 2), bit << (8 * (t & 3u)));
			return;
		}
		// only this k-mer's ops touch the counter, one pair each: every op of the k-mer learns how many they
		// are; the earliest leads them.  An op's pure counters all say the same, and every one of these stores is a
		// 32-byte write transaction to a random place: the first two hash functions speak for all (an op whose first
		// two counters are BOTH shared learns nothing and goes to the rounds -- op_verdict reads n == 0 as that)
		if (tp_j(r) >= e.lead_js) return;
		const bool first_op = (uint32_t)(first[s] >> TILE_IDX_BITS) == t;
		e.lead[t] = (inf & 0x7FFFFFFFu) | (first_op ? LEAD_BIT : 0u);
	});
}
// What becomes of an op once every tile has judged its pairs.  n = the ops of its k-mer K in the batch.
//  * No counter of K is shared: the n ops commute with everything else in the batch; the leader raises
//    the counters to min(m + n, 255) (see above), the others are done.
//  * Some are shared (set S; P = the rest, not empty): K's ops write a shared counter only if it equals
//    K's running minimum at one of them.  Counters only grow, and K's minimum over P runs m_P, m_P + 1,
//    ..., below tg = min(m_P + n, 255).  So if every counter of S holds at least tg NOW, no op of K ever
//    finds one of them at its minimum (or the minimum is 255 and nothing is written at all): K's ops
//    raise P exactly as if S were not there, read nothing anybody else writes in this batch and write
//    nothing anybody else reads -- they commute with the rest like the ops of a k-mer with pure counters
//    only.  (The typical case: a sequencing-error k-mer at count 1 landing on a counter a genome k-mer
//    has taken to 30.)  The other k-mers on S still see S as shared and go to the rounds.
//  * Anything else -- and every op when a bin overflowed -- goes through the reservation rounds.
// Every op of K reads the same flags, the same n and the same counters (nothing writes them between
// the previous batch's rounds and tile_apply), so all of K's ops get the same verdict.
ABG_HD uint32_t bad_slot(const TileEnv& e, uint64_t pos) { return (uint32_t)(pos ^ (pos >> 27)) & e.bad_mask; }
ABG_HD void mark_bad(const TileEnv& e, uint64_t h, unsigned js)
{
	for (unsigned j = 0; j < e.p.nh && j < 8; j++)
		if ((js >> j) & 1u) { const uint32_t s = bad_slot(e, pos_i(e.p, h, j)); atomic_or_u32(&e.bad[s >> 5], 1u << (s & 31u)); }
}
// (plain loads: a mark set by another workgroup during this very pass may be missed -- the pass that set it has changed something,
// so another pass follows and sees it; a device-coherent load goes past the L2 of its XCD, at several times the latency)
ABG_HD bool any_bad(const TileEnv& e, uint64_t h, unsigned js)
{
	bool b = false;
	for (unsigned j = 0; j < e.p.nh && j < 8; j++)
		if ((js >> j) & 1u) { const uint32_t s = bad_slot(e, pos_i(e.p, h, j)); b |= ((e.bad[s >> 5] >> (s & 31u)) & 1u) != 0; }
	return b;
}
// ... and (round 5) the k-mers that DO write a shared counter, as long as nothing about it depends on the order:
//  * K's shared counters all hold at least m_P, the minimum of its own counters P.  Then a shared counter c never falls
//    below K's running minimum over P (an op of K that finds c AT the minimum raises it along with the others, and the
//    other k-mers only ever raise it): K's n ops take P to tg exactly as if S were not there, and each of them raises c
//    only when c equals the running minimum, by one, never past tg.
//  * If EVERY k-mer on c is of that kind (or cannot write c at all: the case above), c ends at the maximum of its old value
//    and the targets of the k-mers on it, in whatever order their ops run: it is at least each target (see above), and no
//    op takes it past the target of its own k-mer.  None of them reads anything an op outside the group writes.
//  * If some k-mer on c goes through the rounds, it reads c at its place in the order: every k-mer that may raise c has to
//    go through the rounds with it.  That is a fixed point over the k-mers and the counters they share: op_verdict marks
//    the counters of the k-mers the rounds get (`bad`, a bit per hashed counter position: a collision sends a k-mer
//    to the rounds that need not go, never the reverse) and leaves the others as CANDIDATES; FCoSettle takes a candidate
//    back that may raise a marked counter and marks its counters, pass after pass until a pass changes nothing (chains
//    of k-mers sharing counters are short: a batch touches ~3 % of the counters); FCoFinal settles what is left, or
//    sends every candidate to the rounds when the last pass still found something.
// (tests/hostcheck; the rule and its closure were first checked against the sequential filter in a simulation of 59
// batches at configs[1]'s and configs[2]'s occupancy, and with counters driven into saturation: notes/README.md.)
ABG_HD void op_verdict(const TileEnv& e, uint64_t t, uint8_t& tgt, uint8_t& pend, uint32_t& centry)
{
	tgt = 0; pend = PEND_ROUNDS; centry = 0;
	if (e.flags[0]) return;
	const uint32_t fl = e.opflag[t], L = e.lead[t], n = L & ~LEAD_BIT;
	if (fl && (!n || e.p.nh > 8 || !e.benign)) { if (e.bad && e.p.nh <= 8) mark_bad(e, e.h0[t], fl); return; }
	if (!fl && !(L & LEAD_BIT)) { pend = PEND_NO; return; } // (its k-mer's leader does the raising)
	const uint64_t h = e.h0[t];
	unsigned mp = 256, ms = 256;
	unsigned cs[8];
	for (unsigned j0 = 0; j0 < e.p.nh; j0 += 4) { // (four counters in flight: a load right before its use is a round trip each)
		unsigned c[4];
#pragma unroll
		for (unsigned q = 0; q < 4; q++) {
			const unsigned jq = j0 + q < e.p.nh ? j0 + q : 0u;
			c[q] = e.cval ? 255u - e.cval[t * e.p.nh + jq] : e.cnt[pos_i(e.p, h, jq)];
		}
#pragma unroll
		for (unsigned q = 0; q < 4; q++) {
			const unsigned j = j0 + q;
			if (j >= e.p.nh) continue;
			if (j < 8) cs[j] = c[q];
			if ((fl >> j) & 1u) ms = c[q] < ms ? c[q] : ms; else mp = c[q] < mp ? c[q] : mp;
		}
	}
	const unsigned tg = mp + n > 255 ? 255u : mp + n;
	if (fl && ms < tg) {
		if (!e.bad) return;
		if (ms < mp) { mark_bad(e, h, fl); return; }
		// a candidate: the shared counters it may raise
		unsigned w = 0;
		for (unsigned j = 0; j < e.p.nh && j < 8; j++) if (((fl >> j) & 1u) && cs[j] < tg) w |= 1u << j;
		e.wmask[t] = (uint8_t)w;
		pend = PEND_CANDIDATE;
		if (w & (w - 1)) centry = CO_MORE;
		else for (unsigned j = 0; j < e.p.nh && j < 8; j++) if ((w >> j) & 1u) centry = bad_slot(e, pos_i(e.p, h, j));
		if (L & LEAD_BIT) tgt = (uint8_t)tg; // (mp < 255: ms < tg <= 255 and ms >= mp)
		return;
	}
	pend = PEND_NO;
	if ((L & LEAD_BIT) && mp < 255) tgt = (uint8_t)tg;
}
struct FOpTarget { // one op per item: its target (leaders) and whether it is left to the reservation rounds
	TileEnv e;
	ABG_HD void operator()(uint64_t t, uint32_t) const
	{
		uint8_t tg, pd;
		uint32_t ce;
		op_verdict(e, t, tg, pd, ce);
		e.tgt[t] = tg; e.pendf[t] = pd;
		if (!e.bad) return;
		// the candidates of this group of ops, packed: what FCoSettle's passes read instead of every op's flags and hashes
		uint32_t cnt;
		const uint32_t r = wave_rank(pd == PEND_CANDIDATE, cnt);
		const uint64_t g0 = (t >> e.glog2) << e.glog2;
		if (pd == PEND_CANDIDATE) e.colist[g0 + r] = (uint32_t)(t - g0) << CO_LANE_SHIFT | ce;
		e.cocnt[t >> e.glog2] = (uint8_t)cnt; // (every lane of the group the same value)
	}
};
struct FCoSettle { // one group of the candidate list per item, pass `pass` >= 1 of the fixed point (op_verdict): a candidate that may raise a marked counter goes to the rounds
	TileEnv e; uint32_t pass;
	ABG_HD void operator()(uint64_t g, uint32_t) const
	{
		if (pass > 1 && !e.chg[pass - 1]) return; // (the pass before changed nothing: done)
		const uint32_t cnt = e.cocnt[g];
		const uint64_t base = g << e.glog2;
		// four entries a turn, and their probes in flight together (a lane's entries are 4 .. 40 bytes in a row: read one at a time
		// they are a chain of round trips, ~100 us a pass of a 30 M-op batch)
		for (uint32_t r0 = 0; r0 < cnt; r0 += 4) {
			uint32_t en[4];
#if defined(__HIP_DEVICE_COMPILE__)
			{ const uint4 v = *(const uint4*)(e.colist + base + r0); en[0] = v.x; en[1] = v.y; en[2] = v.z; en[3] = v.w; } // (the list has 64 entries of slack, a group starts at a multiple of 64)
#else
			for (uint32_t q = 0; q < 4; q++) en[q] = r0 + q < cnt ? e.colist[base + r0 + q] : CO_DEAD;
#endif
			uint32_t word[4];
#pragma unroll
			for (uint32_t q = 0; q < 4; q++) {
				if (r0 + q >= cnt) en[q] = CO_DEAD;
				word[q] = e.bad[((en[q] & CO_MORE) ? 0u : (en[q] & CO_SLOT_MASK)) >> 5]; // (plain loads: see any_bad)
			}
#pragma unroll
			for (uint32_t q = 0; q < 4; q++) {
				if (en[q] == CO_DEAD) continue;
				const uint64_t t = base + (en[q] >> CO_LANE_SHIFT);
				const bool bad = (en[q] & CO_MORE) ? any_bad(e, e.h0[t], e.wmask[t]) : ((word[q] >> (en[q] & 31u)) & 1u) != 0;
				if (!bad) continue;
				e.pendf[t] = PEND_ROUNDS; e.tgt[t] = 0;
				mark_bad(e, e.h0[t], e.opflag[t]);
				e.colist[base + r0 + q] = CO_DEAD;
				e.chg[pass] = 1;
			}
		}
	}
};
struct FCoFinal { // one group of the candidate list per item, when the last pass still took a candidate back: every candidate left goes to the rounds
	TileEnv e; uint32_t last; // (otherwise they stay PEND_CANDIDATE, which the compaction of the rounds' ops reads as settled)
	ABG_HD void operator()(uint64_t g, uint32_t) const
	{
		if (!e.chg[last]) return;
		const uint32_t cnt = e.cocnt[g];
		for (uint32_t r = 0; r < cnt; r++) {
			const uint32_t en = e.colist[(g << e.glog2) + r];
			if (en == CO_DEAD) continue;
			const uint64_t t = (g << e.glog2) + (en >> CO_LANE_SHIFT);
			e.pendf[t] = PEND_ROUNDS; e.tgt[t] = 0;
		}
	}
};
// The ops the tiles leave to the rounds (pendf == PEND_ROUNDS), as a list in op order.  They are few now (0.7 % of a batch):
// a lane counts those of 64 consecutive ops, a scan over the counts places them, a lane writes its ops out -- two reads of the flag
// bytes and a scan of T / 64 numbers instead of a device-wide select over T flags (0.29 -> ~0.05 ms on a 29.8 M-op batch).
constexpr uint32_t PEND_GROUP = 64;
ABG_HD uint32_t pend_bytes(uint64_t x) // how many bytes of x are PEND_ROUNDS
{
	const uint64_t y = x ^ (0x0101010101010101ull * PEND_ROUNDS), lo = 0x7F7F7F7F7F7F7F7Full;
	const uint64_t z = ~(((y & lo) + lo) | y | lo); // 0x80 in every byte of y that is zero
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__popcll(z);
#else
	return (uint32_t)__builtin_popcountll(z);
#endif
}
struct FPendCount { // one group of PEND_GROUP ops per item
	const uint8_t* fl; uint64_t T; uint64_t* cnt;
	ABG_HD void operator()(uint64_t g, uint32_t) const
	{
		const uint64_t t0 = g * PEND_GROUP;
		uint32_t c = 0;
		if (t0 + PEND_GROUP <= T) {
			const uint64_t* w = (const uint64_t*)(fl + t0);
			uint64_t x[PEND_GROUP / 8];
#pragma unroll
			for (uint32_t q = 0; q < PEND_GROUP / 8; q++) x[q] = w[q];
#pragma unroll
			for (uint32_t q = 0; q < PEND_GROUP / 8; q++) c += pend_bytes(x[q]);
		} else
			for (uint64_t t = t0; t < T; t++) c += fl[t] == PEND_ROUNDS;
		cnt[g] = c;
	}
};
struct FPendWrite { // ... after the inclusive scan of cnt: the group's ops to their places, the total to count[0]
	const uint8_t* fl; uint64_t T; const uint64_t* incl; uint32_t* out; uint32_t* count;
	ABG_HD void operator()(uint64_t g, uint32_t) const
	{
		const uint64_t t0 = g * PEND_GROUP, t1 = t0 + PEND_GROUP < T ? t0 + PEND_GROUP : T;
		const uint64_t end = incl[g], beg = g ? incl[g - 1] : 0;
		if (t1 == T) count[0] = (uint32_t)end;
		if (end == beg) return;
		uint64_t o = beg;
		for (uint64_t t = t0; t < t1; t += 8) {
			uint64_t x = *(const uint64_t*)(fl + t); // (the flags have eight bytes of slack)
			if (!pend_bytes(x)) continue;
			for (uint32_t q = 0; q < 8 && t + q < t1; q++, x >>= 8) if ((x & 0xFFu) == PEND_ROUNDS) out[o++] = (uint32_t)(t + q);
		}
	}
};
// `lds`: TILE_COUNTERS bytes
template <class Sync>
ABG_HDN void tile_apply(const TileEnv& e, uint64_t tile, uint8_t* lds, Sync& sy)
{
	if (ld_coherent(&e.flags[0])) return;
	// the tile's bins (TILE_SPLIT of them; the last tile of the range may have fewer)
	const uint64_t nbins = (e.m - e.lo + BIN_COUNTERS - 1) >> BIN_BITS;
	uint32_t nb[TILE_SPLIT], n = 0;
#pragma unroll
	for (uint32_t q = 0; q < TILE_SPLIT; q++) {
		const uint64_t bi = tile * TILE_SPLIT + q;
		const uint32_t f = bi < nbins ? ld_coherent(&e.tcur[bi]) : 0u;
		nb[q] = f < e.cap ? f : e.cap;
		n += nb[q];
	}
	if (!n) return;
	const uint32_t tid = sy.tid(), nt = sy.nthreads();
	const uint64_t base = e.lo + (tile << TILE_BITS);
	const uint32_t span = (uint32_t)(e.m - base < TILE_COUNTERS ? e.m - base : TILE_COUNTERS); // (lo and m are multiples of 8)
	uint64_t* l8 = (uint64_t*)lds;
	const uint64_t* g8 = (const uint64_t*)(e.cnt + base);
	for (uint32_t i = tid; i < span / 8; i += nt) l8[i] = g8[i];
	sy.barrier();
	bool any = false;
	// (measured: fetching several pairs and their targets ahead, or the tile in batches of eight loads, does not help here --
	// 1.56 against 1.50 ms a launch: the kernel streams the array in and out and lives on its many workgroups in flight)
	constexpr uint32_t PER = TILE_SPLIT * ((TILE_SORT_MAX + 1023) / 1024);
	uint32_t mine[PER]; // (a thread's raises as offset | target << 16, for the passes below)
	uint32_t kept = 0;
	const bool keep = e.bad && nt >= 1024;
	// body(pair): its target, when there is one, against the counter
	auto raise = [&](const TilePair& r, bool first) -> bool {
		const uint8_t tg = e.tgt[tp_t(r)];
		if (!tg) return false;
		// (a pure counter has one writer -- its k-mer's leader, possibly through two hash functions
		// with the same value -- so plain byte stores do)
		const uint32_t off = (uint32_t)(pos_i(e.p, tp_h(r), tp_j(r)) - e.lo) & (TILE_COUNTERS - 1);
		if (first && keep && kept < PER) mine[kept++] = off | ((uint32_t)tg << 16);
		if (lds[off] < tg) { lds[off] = tg; return true; }
		return false;
	};
#pragma unroll
	for (uint32_t q = 0; q < TILE_SPLIT; q++) {
		const TilePair* bin = e.bins + (tile * TILE_SPLIT + q) * e.cap;
		for (uint32_t i = tid; i < nb[q]; i += nt) any |= raise(bin[i], true);
	}
	if (e.bad) {
		// a shared counter may have several writers now (op_verdict: k-mers settled together), and it ends at the largest
		// of their targets: plain byte stores again, and passes until every raise finds its counter at or above its target
		for (bool again = any; sy.any(again);) {
			again = false;
			if (keep) {
#pragma unroll
				for (uint32_t q = 0; q < PER; q++)
					if (q < kept) { const uint32_t off = mine[q] & 0xFFFFu; const uint8_t tg = (uint8_t)(mine[q] >> 16); if (lds[off] < tg) { lds[off] = tg; again = true; } }
			} else {
				for (uint32_t q = 0; q < TILE_SPLIT; q++) {
					const TilePair* bin = e.bins + (tile * TILE_SPLIT + q) * e.cap;
					for (uint32_t i = tid; i < nb[q]; i += nt) again |= raise(bin[i], false);
				}
			}
		}
	}
	if (sy.any(any)) {
		uint64_t* o8 = (uint64_t*)(e.cnt + base);
		for (uint32_t i = tid; i < span / 8; i += nt) o8[i] = l8[i];
	}
}
// ---- a batch whose pairs do not fit its bins (Engine::insert_sorted) ----
// A bin holds what a tile sees of a batch of random k-mers, with room to spare; a k-mer that recurs thousands of times in a
// batch -- a homopolymer run, a satellite, a collapsed repeat at high coverage -- puts all its pairs on H counters and runs
// the bins of those over.  Such a batch is judged and applied without the tiles: every (op, counter) pair, sorted by counter
// position (a stable radix sort: a counter's pairs stay in op order), the segments of equal positions judged as tile_purity
// judges a counter's pairs -- same flags, same `lead` -- and the leaders' targets raised with a byte-wide atomic maximum.
// What op_verdict, the fixed point and the reservation rounds do in between is the tile path's, unchanged.
struct SortEnv {
	TileEnv e; uint64_t T, np;        // np = T * nh pairs
	uint64_t* key; uint32_t* val;     // [np] counter position; op << 4 | hash function -- sorted by position
	uint64_t* sid;                    // [np] 1 on the first pair of a position, then (inclusive sum) the pair's segment, from 1
	uint32_t* start;                  // [segments + 1] first pair of every segment; np after the last
	uint8_t* impure;                  // [segments] another k-mer on the counter (or the earliest op twice)
};
struct FSortKeys { // one pair per item, op-major: pair q belongs to op q / nh
	SortEnv s;
	ABG_HD void operator()(uint64_t q, uint32_t) const
	{
		const uint64_t t = q / s.e.p.nh; const unsigned j = (unsigned)(q - t * s.e.p.nh);
		s.key[q] = pos_i(s.e.p, s.e.h0[t], j);
		s.val[q] = (uint32_t)t << 4 | j;
	}
};
struct FSegHeads {
	SortEnv s;
	ABG_HD void operator()(uint64_t i, uint32_t) const { s.sid[i] = (i == 0 || s.key[i] != s.key[i - 1]) ? 1u : 0u; }
};
struct FSegStarts { // after the inclusive sum
	SortEnv s;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		if (i == 0 || s.key[i] != s.key[i - 1]) { s.start[s.sid[i] - 1] = (uint32_t)i; s.impure[s.sid[i] - 1] = 0; }
		if (i + 1 == s.np) s.start[s.sid[i]] = (uint32_t)s.np;
	}
};
struct FSegImpure { // tile_purity's test of a pair against the counter's earliest pair
	SortEnv s;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint64_t g = s.sid[i] - 1, f = s.start[g];
		if (i == f) return;
		const uint32_t v = s.val[i], vf = s.val[f];
		if (s.e.h0[v >> 4] != s.e.h0[vf >> 4] || (v >> 4) == (vf >> 4)) s.impure[g] = 1;
	}
};
struct FSegJudge { // ... and what it then does for the pair's op
	SortEnv s;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint64_t g = s.sid[i] - 1;
		const uint32_t f = s.start[g], cnt = s.start[g + 1] - f, v = s.val[i], t = v >> 4, j = v & 15u;
		if (s.impure[g] || cnt >= s.e.count_max) {
			const uint32_t bit = s.e.p.nh <= 8 ? (1u << j) : 0xFFu;
			atomic_or_u32((uint32_t*)s.e.opflag + (t >> 2), bit << (8 * (t & 3u)));
			return;
		}
		if (j >= s.e.lead_js) return;
		s.e.lead[t] = cnt | ((s.val[f] >> 4) == t ? LEAD_BIT : 0u);
	}
};
struct FSegApply { // tile_apply's raise: the counter to the maximum of what it holds and its leaders' targets
	SortEnv s;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint8_t tg = s.e.tgt[s.val[i] >> 4];
		if (!tg) return;
		const uint64_t pos = s.key[i];
		uint32_t* w = (uint32_t*)(s.e.cnt + (pos & ~3ull));
		const unsigned sh = 8u * (unsigned)(pos & 3u);
		for (uint32_t old = *w;;) {
			if (((old >> sh) & 0xFFu) >= tg) return;
			const uint32_t want = (old & ~(0xFFu << sh)) | ((uint32_t)tg << sh);
			const uint32_t got = cas_u32(w, old, want);
			if (got == old) return;
			old = got;
		}
	}
};
struct FClaimList { // FClaim over a list of ops
	Params p; const uint64_t* h0; const uint32_t* pend; uint64_t* claim; uint64_t cmask; uint32_t epoch;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t t = pend[i];
		const uint64_t h = h0[t];
		const uint64_t v = claim_val(epoch, t);
		for (unsigned j = 0; j < p.nh; j++) atomic_min_u64(&claim[pos_i(p, h, j) & cmask], v);
	}
};

// ---- the tiles of a partitioned run (Engine::insert_range): every rank bins, judges and applies
// the pairs on the counters it owns; what an op needs to know from the other ranks travels in
// two bytes per op through one all_reduce(MAX):
//   [t]          255: some counter of the op's k-mer is shared with another k-mer (whoever owns it saw
//                that); else n < 254: the op leads the n ops of its k-mer (every owner of a pure counter
//                says the same; tile_purity sends k-mers with more ops to the rounds)
//   [T + t]      255 - (minimum of the leader's counters this rank owns); 0 from a rank that owns none
//   [2T]         1: a bin overflowed somewhere (the whole batch takes the reservation rounds, on every rank)
struct FDistPack {
	TileEnv e; uint64_t T; uint8_t* buf;
	ABG_HD void operator()(uint64_t t, uint32_t) const
	{
		const uint32_t n = (e.lead[t] & LEAD_BIT) ? (e.lead[t] & ~LEAD_BIT) : 0u; // (what the op LEADS)
		buf[t] = (uint8_t)(e.opflag[t] ? 255u : n);
		uint8_t v = 0;
		if (n) {
			const uint64_t h = e.h0[t];
			unsigned mn = 255;
			for (unsigned j = 0; j < e.p.nh; j++) {
				const uint64_t q = pos_i(e.p, h, j);
				if (q - e.lo >= e.m - e.lo) continue;
				const unsigned c = e.cnt[q];
				mn = c < mn ? c : mn;
			}
			v = (uint8_t)(255u - mn);
		}
		buf[T + t] = v;
		if (t == 0) buf[2 * T] = e.flags[0] ? 1 : 0;
	}
};
struct FDistTarget { // FOpTarget from the combined bytes
	TileEnv e; uint64_t T; const uint8_t* buf;
	ABG_HD void operator()(uint64_t t, uint32_t) const
	{
		if (t == 0 && buf[2 * T]) e.flags[0] = 1;
		const bool flag = buf[t] == 255;
		const unsigned n = buf[t], mn = 255u - buf[T + t];
		e.pendf[t] = flag ? 1 : 0;
		uint8_t tg = 0;
		if (!flag && !buf[2 * T] && n && mn < 255) tg = (uint8_t)(mn + n > 255 ? 255u : mn + n);
		e.tgt[t] = tg;
	}
};
// ... and with the rules of rounds 4 and 5 (op_verdict: k-mers that cannot write their shared counters, k-mers that raise shared
// counters settled together), which ask more of the other ranks: every counter of the op, and WHICH of them are shared.
//   mx (all_reduce MAX): [t] n, the ops of the k-mer in the batch (0: this rank owns no pure counter of it; < 254, tile_purity)
//                        [T + t] 1: the op is the earliest of them
//                        [2T + t nh + j] 255 - counter j of the op, from the rank that owns it (the others: 0 -- and 0 from the
//                        owner is a counter at 255, which is what the maximum then says)
//                        [(2 + nh) T] 1: a bin overflowed somewhere
//   sm (all_reduce SUM): [t] bit j: counter j is shared with another k-mer of the batch (one rank owns a counter: the bits of the
//                        ranks are disjoint and their sum is their union)
// Every rank then holds what op_verdict reads on one GPU -- for EVERY op: verdicts, the table of marked counters and the passes
// of the fixed point come out the same on every rank, and nothing else has to travel.
struct FDistPack2 {
	TileEnv e; uint64_t T; uint8_t* mx; uint8_t* sm;
	ABG_HD void operator()(uint64_t t, uint32_t) const
	{
		const uint32_t L = e.lead[t];
		mx[t] = (uint8_t)(L & 0xFFu);
		mx[T + t] = (L & LEAD_BIT) ? 1 : 0;
		sm[t] = e.opflag[t];
		const uint64_t h = e.h0[t];
		uint8_t* cv = mx + 2 * T + t * e.p.nh;
		for (unsigned j0 = 0; j0 < e.p.nh; j0 += 4) { // (the loads of four counters together, none under a condition)
			unsigned c[4]; bool own[4];
#pragma unroll
			for (unsigned q = 0; q < 4; q++) {
				const uint64_t pos = pos_i(e.p, h, j0 + q < e.p.nh ? j0 + q : 0u);
				own[q] = pos - e.lo < e.m - e.lo;
				c[q] = e.cnt[own[q] ? pos : e.lo];
			}
#pragma unroll
			for (unsigned q = 0; q < 4; q++) if (j0 + q < e.p.nh) cv[j0 + q] = own[q] ? (uint8_t)(255u - c[q]) : (uint8_t)0;
		}
		if (t == 0) mx[(2 + e.p.nh) * T] = e.flags[0] ? 1 : 0;
	}
};
struct FDistUnpack2 { // the combined bytes back where op_verdict reads them
	TileEnv e; uint64_t T; const uint8_t* mx; const uint8_t* sm;
	ABG_HD void operator()(uint64_t t, uint32_t) const
	{
		e.lead[t] = (uint32_t)mx[t] | (mx[T + t] ? LEAD_BIT : 0u);
		e.opflag[t] = sm[t];
		if (t == 0 && mx[(2 + e.p.nh) * T]) e.flags[0] = 1;
	}
};
struct FClaimOwned { // FClaim / FClaimList on the counters of [lo, lo + span) only (pend == NULL: all ops)
	Params p; const uint64_t* h0; const uint32_t* pend; uint64_t* claim; uint64_t cmask; uint32_t epoch; uint64_t lo, span;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t t = pend ? pend[i] : (uint32_t)i;
		const uint64_t h = h0[t];
		const uint64_t v = claim_val(epoch, t);
		for (unsigned j = 0; j < p.nh; j++) {
			const uint64_t q = pos_i(p, h, j);
			if (q - lo < span) atomic_min_u64(&claim[q & cmask], v);
		}
	}
};


// ---- the routed form of a partitioned batch (Engine::insert_tiles_routed): instead of every rank holding every
// op's hash and picking out the pairs on its own counters, a rank hashes its slice of the ops and SENDS each
// (op, counter) pair to the rank that owns the counter -- one personalised exchange (abg_comm::all_to_all_v) of
// 12-byte records, the role Parallel/NetworkSequenceCollection.cpp:1499-1506 (computeNodeID) + one MPI message per
// k-mer play in the reference.  What comes back the same way: two bytes per pair (FRouteReply: the FDistPack
// bytes, per pair), then one byte per pair out again (the leader's target).  A rank bins, judges and applies
// what it received; binning work and exchanged bytes per rank fall as 1 / ranks.
struct RouteEnv {
	Params p; const uint64_t* h0;
	uint64_t a, b;               // this rank hashed the ops [a, b) of the batch
	uint64_t chunk; uint32_t world; // position pos belongs to rank pos / chunk
	TilePair* send; uint32_t cap;   // [world][cap] records by destination
	uint32_t* scur;              // [world] records per destination
	uint32_t* slot;              // [(b - a) * nh] where each (op, hash function) went: destination * cap + index (~0: nowhere, overflow)
	uint32_t* flags;             // [0] a destination's room ran out
};
struct FRoutePack { // item: a chunk of BIN_CHUNK_OPS of the rank's own ops; a workgroup-local counting sort by destination
	static constexpr uint32_t FAST = 2 * MAX_RANKS * 4, THREADS = 256;
	RouteEnv r;
	ABG_HD uint32_t dest(uint64_t pos) const
	{
		uint32_t q = 0;
		while (q + 1 < r.world && pos >= (uint64_t)(q + 1) * r.chunk) q++;
		return q;
	}
	template <class Sync> ABG_HDN void operator()(uint64_t c, void* fast, Sync& sy) const
	{
		uint32_t* hist = (uint32_t*)fast; uint32_t* cur = hist + MAX_RANKS;
		const uint32_t tid = sy.tid(), nt = sy.nthreads();
		const uint64_t t0 = r.a + c * BIN_CHUNK_OPS, t1 = t0 + BIN_CHUNK_OPS < r.b ? t0 + BIN_CHUNK_OPS : r.b;
		for (uint32_t i = tid; i < r.world; i += nt) hist[i] = 0;
		sy.barrier();
		for (uint64_t t = t0 + tid; t < t1; t += nt) {
			const uint64_t h = r.h0[t];
			for (unsigned j = 0; j < r.p.nh; j++) atomic_add_u32(&hist[dest(pos_i(r.p, h, j))], 1);
		}
		sy.barrier();
		for (uint32_t i = tid; i < r.world; i += nt) cur[i] = hist[i] ? atomic_add_u32(&r.scur[i], hist[i]) : 0;
		sy.barrier();
		for (uint64_t t = t0 + tid; t < t1; t += nt) {
			const uint64_t h = r.h0[t];
			for (unsigned j = 0; j < r.p.nh; j++) {
				const uint32_t q = dest(pos_i(r.p, h, j));
				const uint32_t s = atomic_add_u32(&cur[q], 1);
				uint32_t where = 0xFFFFFFFFu;
				if (s >= r.cap) r.flags[0] = 1;
				else {
					where = q * r.cap + s;
					TilePair& o = r.send[where];
					o.hlo = (uint32_t)h; o.hhi = (uint32_t)(h >> 32); o.tj = (uint32_t)t | ((uint32_t)j << TP_T_BITS);
				}
				r.slot[(t - r.a) * r.p.nh + j] = where;
			}
		}
	}
};
struct FBinCoarseRec { // FBinCoarse over received records: item = a chunk of BIN_CHUNK_PAIRS of them, each ONE pair on a counter of this rank
	static constexpr uint32_t FAST = 2 * BIN_MAX_COARSE * 4, THREADS = 256;
	BinEnv b; const TilePair* recs; uint64_t nrec;
	template <class Sync> ABG_HDN void operator()(uint64_t c, void* fast, Sync& sy) const
	{
		uint32_t* hist = (uint32_t*)fast; uint32_t* cur = hist + b.ncoarse;
		const uint32_t tid = sy.tid(), nt = sy.nthreads();
		const uint64_t i0 = c * BIN_CHUNK_PAIRS, i1 = i0 + BIN_CHUNK_PAIRS < nrec ? i0 + BIN_CHUNK_PAIRS : nrec;
		const uint64_t span = b.e.m - b.e.lo;
		auto coarse_of = [&](const TilePair& r) -> uint32_t {
			const uint64_t pos = pos_i(b.e.p, tp_h(r), tp_j(r)) - b.e.lo;
			return pos < span ? (uint32_t)((pos >> BIN_BITS) >> b.cshift) : 0xFFFFFFFFu; // (never: the sender computed the same owner)
		};
		for (uint32_t i = tid; i < b.ncoarse; i += nt) hist[i] = 0;
		sy.barrier();
		for (uint64_t i = i0 + tid; i < i1; i += nt) {
			const uint32_t cb = coarse_of(recs[i]);
			if (cb == 0xFFFFFFFFu) { b.e.flags[0] = 1; continue; }
			atomic_add_u32(&hist[cb], 1);
		}
		sy.barrier();
		for (uint32_t i = tid; i < b.ncoarse; i += nt) cur[i] = hist[i] ? atomic_add_u32(&b.ccur[i], hist[i]) : 0;
		sy.barrier();
		for (uint64_t i = i0 + tid; i < i1; i += nt) {
			const TilePair r = recs[i];
			const uint32_t cb = coarse_of(r);
			if (cb == 0xFFFFFFFFu) continue;
			const uint32_t slot = atomic_add_u32(&cur[cb], 1);
			if (slot >= b.ccap) { b.e.flags[0] = 1; continue; }
			b.coarse[(uint64_t)cb * b.ccap + slot] = r;
		}
	}
};
struct FRouteReply { // one received record per item: what its owner's tiles found out about THIS pair's counter
	// [0] 255: the counter is shared with another k-mer of the batch; else the k-mer's op count n (1 .. 126: a counter with 127 pairs
	//     or more counts as shared on this path, TileEnv::count_max) | 0x80 when the op is the earliest of them
	// [1] 255 - the counter, shared or not (round 6: what the hashing rank needs for op_verdict's round-4 rule)
	TileEnv e; const TilePair* recs; uint8_t* rep;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const TilePair r = recs[i];
		const uint32_t t = tp_t(r), j = tp_j(r), L = e.lead[t];
		const unsigned fl = e.opflag[t];
		const bool shared = e.p.nh <= 8 ? ((fl >> j) & 1u) != 0 : fl != 0;
		rep[2 * i] = (uint8_t)(shared ? 255u : ((L & LEAD_BIT) ? 0x80u : 0u) | (L & 0x7Fu));
		rep[2 * i + 1] = (uint8_t)(255u - e.cnt[pos_i(e.p, tp_h(r), j)]);
	}
};
struct FRouteCombine { // one of the rank's own ops per item: the replies of its nh pairs -> verdict, and the target back out to each pair
	// op_verdict without the candidates (abg_engine.h, round 4's rule): a k-mer none of whose counters is shared is settled by its
	// leader; one with shared counters too, when every one of them already holds the target min(m_P + n, 255) -- its ops then
	// never find a shared counter at their minimum, read nothing another k-mer writes and write nothing another reads; anything
	// else takes the partitioned reservation rounds.  (benign == 0: round 2's rule -- any shared counter sends the k-mer there.)
	Params p; const uint32_t* slot; const uint8_t* rep; uint8_t* tgt_out; uint8_t* pendf; uint32_t benign; // (pendf: [own ops])
	ABG_HD void operator()(uint64_t u, uint32_t) const
	{
		unsigned n = 0, mp = 256, ms = 256; bool leader = false, any_shared = false;
		for (unsigned j = 0; j < p.nh; j++) {
			const uint32_t s = slot[u * p.nh + j];
			const unsigned a = rep[2ull * s], c = 255u - rep[2ull * s + 1];
			if (a == 255u) { any_shared = true; ms = c < ms ? c : ms; }
			else { n = (a & 0x7Fu) > n ? (a & 0x7Fu) : n; leader = leader || (a & 0x80u) != 0; mp = c < mp ? c : mp; }
		}
		const unsigned tg = mp + n > 255 ? 255u : mp + n;
		bool rounds = false;
		if (any_shared) rounds = !benign || !n || p.nh > 8 || ms < tg;
		uint8_t out = 0;
		if (!rounds && leader && n && mp < 255) out = (uint8_t)tg;
		pendf[u] = rounds ? 1 : 0;
		for (unsigned j = 0; j < p.nh; j++) tgt_out[slot[u * p.nh + j]] = out;
	}
};
struct FRouteTgt { // one received record per item: the target its op's hashing rank sent back
	const TilePair* recs; const uint8_t* tgt_in; uint8_t* tgt;
	ABG_HD void operator()(uint64_t i, uint32_t) const { tgt[tp_t(recs[i])] = tgt_in[i]; }
};
struct FRoutePendRec { // the rank's own ops that go to the rounds, as records for everybody: {hash, op}
	const uint64_t* h0; const uint32_t* list; uint64_t a; TilePair* out;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint64_t t = a + list[i], h = h0[t];
		out[i].hlo = (uint32_t)h; out[i].hhi = (uint32_t)(h >> 32); out[i].tj = (uint32_t)t;
	}
};
struct FRoutePendTake { // ... and everybody's, gathered in rank (= op) order: the hashes into place, the op list
	const TilePair* recs; uint64_t* h0; uint32_t* pend;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const TilePair r = recs[i];
		h0[r.tj] = tp_h(r);
		pend[i] = r.tj;
	}
};

struct FTilePurity { // tile procedures as items of Backend::launch_tiles: f(tile, FAST bytes of fast memory, sync), THREADS per item
	static constexpr uint32_t FAST = TILE_PURITY_FAST, THREADS = TILE_PURITY_THREADS;
	TileEnv e;
	template <class Sync> ABG_HDN void operator()(uint64_t tile, void* fast, Sync& sy) const { tile_purity(e, tile, fast, sy); }
};
struct FTileApply {
	static constexpr uint32_t FAST = TILE_COUNTERS, THREADS = 1024;
	TileEnv e;
	template <class Sync> ABG_HDN void operator()(uint64_t tile, void* fast, Sync& sy) const { tile_apply(e, tile, (uint8_t*)fast, sy); }
};

// ============================================================ PASS 2 functors
// provisional per-read result of the classify step
constexpr uint8_t RES_CANDIDATE = 0x80;

// ---- the committed contigs as the classification reads them (round 6) ----
// allKmersInBloom(seq, solidKmerSet) and allKmersInBloom(seq, assembledKmerSet) (bloom-dbg.h:58-77,816-828) ask 2 x H bits of
// every k-mer of every read: 87 k-mers x 4 sectors of a 2 GB array for a 150 bp read, nine tenths of them for reads that lie on
// sequence some earlier read has assembled.  Every k-mer of an inserted contig is solid (the walk's vertices are: successor()
// follows solid neighbours, the seed read is entirely solid) and visited (addKmersToBloom, bloom-dbg.h:79-90, FPcApply), and a
// filter's answer depends on a k-mer's canonical hash only.  So the commit copies every contig it inserts into an archive -- a
// byte per base, contigs separated by ARC_SEP bytes, which no base equals -- and leaves "the k-mer with this hash starts at
// archive position P" in a direct-mapped table; the classification looks its current k-mer up, compares the read with the
// archive base by base (either strand), and every k-mer of the read inside the run that matched needs no probe: both answers
// are yes.  A wrong or stale table entry, a run cut short by a separator, an archive half written by a commit running beside
// the classification of the next batch: the comparison fails or ends early and the k-mers are probed as before.  The archive
// only ever holds contigs of batches before the one being classified, so a verdict drawn from it is one the visited filter
// gives at the read's turn (visited bits are never cleared while it lives: Engine::reset and visited_dev drop it).
constexpr uint8_t ARC_SEP = 0xFF;
constexpr uint64_t ARC_HEAD = 8, ARC_PAD = 16; // separators before the first contig / readable bytes after the last position
constexpr uint32_t ARC_POS_BITS = 40;
struct ContigArchive {
	uint8_t* seq = nullptr;   // [cap + ARC_PAD] 0..3 a base (4: the 'N' of a contig column under a spaced seed), ARC_SEP everywhere else
	uint64_t cap = 0;
	uint64_t* used = nullptr; // [1] bytes handed out (from ARC_HEAD; beyond cap: contigs that found no room and are not there)
	uint64_t* tab = nullptr;  // [mask + 1] bits 40..63 of the k-mer's hash << 40 | archive position of its first base; 0: empty
	uint64_t mask = 0;
	uint64_t* nreads = nullptr; // [2] statistics: reads the archive answered for at least one k-mer; ... for everything (arc_ends_decided)
};
ABG_HD uint64_t arc_slot(uint64_t h, uint64_t mask) { return ((h * 0x9E3779B97F4A7C15ULL) >> 22) & mask; }
ABG_HD uint64_t arc_entry(uint64_t h, uint64_t pos) { return (h & ~((1ULL << ARC_POS_BITS) - 1)) | pos; }
ABG_HD uint64_t arc_load8(const uint8_t* q) { uint64_t x; __builtin_memcpy(&x, q, 8); return x; }
// bases i .. i + 7 of a packed read of L bases, a byte each (past the end: whatever the read's last word holds)
ABG_HD uint64_t arc_read8(const uint32_t* words, uint64_t woff, uint32_t L, uint32_t i)
{
	const uint32_t a = i >> 4, last = (L - 1) >> 4, b = a < last ? a + 1 : last;
	const uint64_t w = (uint64_t)words[woff + a] | ((uint64_t)words[woff + b] << 32);
	uint64_t t = (w >> (2u * (i & 15u))) & 0xFFFFu;
	t = (t | (t << 24)) & 0x000000FF000000FFULL;
	t = (t | (t << 12)) & 0x000F000F000F000FULL;
	t = (t | (t << 6)) & 0x0303030303030303ULL;
	return t;
}
// How many k-mers of the read, from k-mer j on, are consecutive k-mers of ONE archived contig, the first of them the k-mer at
// archive position P read forwards or backwards (0: the k-mer at P is another one).  The loads of a stretch of ARC_CHUNK x 8
// bases go out together, none under a condition (a comparison that stops at its first difference is a round trip per word):
// the way the contig reads is settled by the first eight bases, then two or three stretches cover a 150 bp read.  An address
// past the end of a run is clamped into the archive -- what it yields is only looked at when every base before it matched, and
// then (a separator ends every run, the first of them at ARC_HEAD - 1) it was not clamped.
constexpr uint32_t ARC_CHUNK = 8;
ABG_HD uint32_t arc_cover(const ContigArchive& a, uint64_t P, const uint32_t* words, uint64_t woff, uint32_t L, uint32_t j, unsigned k, bool* same_strand = nullptr)
{
	if (P < ARC_HEAD || P + k > a.cap) return 0;
	const uint32_t n = L - j;
	const uint64_t E = P + k - 1, top = a.cap + ARC_PAD - 8, C3 = 0x0303030303030303ULL;
	// read base j + t against seq[P + t], or against the complement of seq[E - t]: eight bytes ending there, turned round
	const uint64_t r0 = arc_read8(words, woff, L, j);
	const uint64_t f0 = arc_load8(a.seq + P) ^ r0, b0 = (__builtin_bswap64(arc_load8(a.seq + E - 7)) ^ C3) ^ r0;
	const uint32_t first = n < 8 ? n : 8u;
	const bool fwd = (f0 ? (uint32_t)__builtin_ctzll(f0) >> 3 : 8u) >= first;
	if (!fwd && (b0 ? (uint32_t)__builtin_ctzll(b0) >> 3 : 8u) < first) return 0;
	if (same_strand) *same_strand = fwd;
	uint32_t t = first;
	while (t < n) {
		uint64_t d[ARC_CHUNK];
#pragma unroll
		for (uint32_t c = 0; c < ARC_CHUNK; c++) {
			const uint32_t tt = t + 8 * c, jj = j + tt < L ? j + tt : L - 1;
			uint64_t av;
			if (fwd) { const uint64_t q = P + tt; av = arc_load8(a.seq + (q < top ? q : top)); }
			else { const uint64_t q = E - 7 - (tt < E - 7 ? tt : E - 7); av = __builtin_bswap64(arc_load8(a.seq + q)) ^ C3; }
			d[c] = av ^ arc_read8(words, woff, L, jj);
		}
		bool done = false;
#pragma unroll
		for (uint32_t c = 0; c < ARC_CHUNK; c++) {
			if (done || t >= n) continue;
			const uint32_t m = d[c] ? (uint32_t)__builtin_ctzll(d[c]) >> 3 : 8u;
			t += m < n - t ? m : n - t;
			done = m < 8;
		}
		if (done) break;
	}
	return t >= k ? t - k + 1 : 0u;
}

// ---- hasBluntEnd's two look-aheads answered from the archive (round 6) ----
// A read that lies on ONE archived contig with at least FP_TRIM more contig bases beyond either end starts both of its
// look-aheads (bloom-dbg.h:489-532: lookAhead(REVERSE, 5) from its first k-mer and from the reverse complement of its last) on
// a path of five solid vertices c_1 .. c_5, the contig's own.  That alone does not make the search return true: lookAhead
// (ExtendPath.h:100-161) shares ONE visited set among its branches and never erases it, and identity is strand-agnostic, so a
// branch tried earlier that reaches a vertex with the identity of some c_i by another route, and fails from there, blocks the
// contig's path.  When can that happen?  Suppose the search fails.  Every vertex it entered lies at depth <= 4 (depth 5
// returns true) and had all its solid neighbours tried, so each of them was entered too, then or earlier.  c_1 is a solid
// neighbour of the start: something with c_1's identity was entered -- c_1 itself (at whatever depth), or its reverse
// complement reached within 4 steps (B).  If never (B), c_1 was expanded as itself, so c_2 was entered, ..., so something with
// c_5's identity was entered at depth d <= 4: c_5 as a d-step shift of the start (A) or its reverse complement (B).  With
// z(i), i in [-5, k), the start k-mer's bases (i >= 0) and the contig's |i|-th base beyond it (i < 0), as the search sees them:
//   (A) makes the start's first k - 4 bases periodic with period 5 - d: excluded when for every p in 1..5 some i in [0, 8) has
//       z(i) != z(i + p);
//   (B), for c_i and depth d with T = i + d in 1..9, makes z(q) = comp(z(k + 4 - T - ... )) a reverse-palindrome over a range that
//       always holds q in [k - 12, k - 4): excluded when for every T some u in [0, 8) has z(k - 12 + u) != comp(z(11 - T - u)).
// Both are properties of 26 bases; where they hold (all but low-complexity and hairpin ends) the search cannot fail, whatever
// the order it tries the neighbours in.  Where they do not, or the margins are short, the searches are run.
// z for the first look-ahead: z(i) = r(i); for the second (from the reverse complement of the last k-mer): z(i) = comp(r(L - 1 - i)),
// with r(j) the read's strand of the contig extended beyond the read: seq[A + j] (same strand, read base 0 at A) or
// comp(seq[E - j]) (other strand, read base 0 at E).
ABG_HD bool arc_ends_decided(const ContigArchive& a, bool same, uint64_t anchor, uint32_t L, unsigned k)
{
	if (k < 20) return false;
	// r(j), j in [-5, L + 5): inside the archive?  (the read itself matched, so [0, L) is; a separator or an 'N' fails the tests below)
	if (same ? (anchor < ARC_HEAD + 5 || anchor + L + 5 > a.cap) : (anchor < ARC_HEAD + L + 4 || anchor + 6 > a.cap)) return false;
	auto r = [&](int j) -> unsigned { const unsigned c = same ? a.seq[anchor + j] : a.seq[anchor - j]; return c > 3u ? 0x80u : (same ? c : 3u - c); };
	bool ok = true;
	for (int end = 0; end < 2; end++) {
		unsigned zl[18], zh[8]; // z(-5 .. 12], z(k - 12 .. k - 5]
#pragma unroll
		for (int q = 0; q < 18; q++) { const int i = q - 5; zl[q] = end ? r((int)L - 1 - i) : r(i); }
#pragma unroll
		for (int q = 0; q < 8; q++) { const int i = (int)k - 12 + q; zh[q] = end ? r((int)L - 1 - i) : r(i); }
		if (end) { // (comp; 0x80 stays out of range)
#pragma unroll
			for (int q = 0; q < 18; q++) zl[q] = zl[q] > 3u ? 0x80u : 3u - zl[q];
#pragma unroll
			for (int q = 0; q < 8; q++) zh[q] = zh[q] > 3u ? 0x80u : 3u - zh[q];
		}
		unsigned bad = 0;
#pragma unroll
		for (int q = 0; q < 18; q++) bad |= zl[q];
#pragma unroll
		for (int q = 0; q < 8; q++) bad |= zh[q];
		ok = ok && !(bad & 0x80u); // five contig bases beyond the end, no separator, no 'N' anywhere looked at
#pragma unroll
		for (int pp = 1; pp <= 5; pp++) { // (A)
			bool diff = false;
#pragma unroll
			for (int i = 0; i < 8; i++) diff = diff || zl[i + 5] != zl[i + pp + 5];
			ok = ok && diff;
		}
#pragma unroll
		for (int T = 1; T <= 9; T++) { // (B)
			bool diff = false;
#pragma unroll
			for (int u = 0; u < 8; u++) diff = diff || zh[u] != 3u - zl[11 - T - u + 5];
			ok = ok && diff;
		}
	}
	return ok;
}

// (A wave-per-read form of this kernel -- k-mers over the lanes, hashed from scratch, the two
// blunt-end searches in lock step -- was measured at 375 ms per config-1 step against 155 ms for
// this one: hashing every k-mer from scratch costs more ALU than the probes cost memory time, and
// the 6 G probes of a step are what bounds the kernel.  One read per lane it stays; the next
// batch's classification is queued on a side stream ahead of the current batch's walkers.)
#ifndef ABG_CLS_GROUP
#define ABG_CLS_GROUP 8 // (8 with the two-bit array -- one load a position, 32 in flight a round -- 602.7 vs 607.3 ms a step for 4)
#endif
constexpr uint32_t CLS_GROUP = ABG_CLS_GROUP; // k-mers of a read probed per round of FClassify / FRefilter
template <int NW>
struct FClassify { // processRead up to the visited test (bloom-dbg.h:798-828), one read per item
	Params p; Batch b; uint64_t first; const uint8_t* cnt; const uint8_t* vis; uint8_t* result;
	VKey* la_pool; // [slots][LA_MAX_VISITED]
	const uint8_t* both = nullptr; // the solid plane's and the visited filter's bits side by side (FBothBuild), or NULL
	ContigArchive arc{};           // the contigs committed so far (seq == NULL: none kept)
	uint32_t dbg_skip = 0;         // diagnosis only, WRONG verdicts (ABG_CLS_DEBUG_SKIP): 1 no look-aheads, 2 no sweep -- what each part of the kernel costs
	ABG_HDN void operator()(uint64_t i, uint32_t slot) const
	{
		uint64_t r = first + i;
		uint32_t L = b.len[r];
		unsigned k = p.k;
		uint32_t nk = L - k + 1;
		SearchScratch<NW> sc;
		sc.tb = nullptr; sc.tb_keys = nullptr; sc.tb_cap = 0; sc.overflow = 0; sc.dbg_nodes = 0;
		sc.tbf = nullptr; sc.tbf_keys = nullptr; sc.tbf_cap = 0; sc.tbk_cap = 0; sc.coop = false;
		sc.guide = Guide{ nullptr, 0, nullptr, 0, nullptr }; sc.bulk = nullptr; sc.dbg_chain = 0; sc.dbg_on = 0; sc.n_chain_steps = 0; sc.dbg_la = 0; sc.dbg_la_calls = 0;
		sc.memo = SuccMemo{ nullptr, nullptr, nullptr, 0 }; sc.n_memo_hits = 0; sc.n_memo_adds = 0; sc.wstats = nullptr; sc.mcache = nullptr; sc.la_fast = nullptr; sc.la_fast_cap = 0;
		sc.la = sc.la_local;
		sc.la_visited = la_pool + (uint64_t)slot * LA_MAX_VISITED;
		// hasBluntEnd (bloom-dbg.h:489-532): lookAhead(REVERSE, 5) from the first k-mer of
		// the read and from the first k-mer of its reverse complement
		Vtx<NW> v;
		v.s = batch_kmer<NW>(b, r, 0, k);
		vtx_rehash(p, v);
		Vtx<NW> first_v = v;
		// A read lying on one archived contig, well inside it: both look-aheads are known to succeed (arc_ends_decided), every k-mer
		// is solid and visited -- the verdict without a probe.  (Plain builds: under a spaced seed identities ignore the masked bases.)
		uint32_t cov0 = 0; // k-mers from the read's first that the archive answers for (the sweep below starts behind them)
		if constexpr (!MASKED_BUILD<NW>) {
			if (arc.seq && !dbg_skip) {
				const uint64_t h0 = vtx_hash(p, v), en0 = arc.tab[arc_slot(h0, arc.mask)];
				if (en0 && !((en0 ^ h0) >> ARC_POS_BITS)) {
					bool same = true;
					const uint64_t P0 = en0 & ((1ULL << ARC_POS_BITS) - 1);
					cov0 = arc_cover(arc, P0, b.words, b.woff[r], L, 0, k, &same);
					if (cov0 == nk && arc_ends_decided(arc, same, same ? P0 : P0 + k - 1, L, k)) {
						if (arc.nreads) { wave_count_add(arc.nreads, true); wave_count_add(arc.nreads + 1, true); }
						result[r] = (uint8_t)RR_ALL_KMERS_VISITED;
						return;
					}
				}
			}
		}
		// (the search in registers, look_ahead_reg; look_ahead_t under a spaced seed and for the rare search whose visited set outgrows them)
		auto la = [&](const Vtx<NW>& s) -> bool {
			if constexpr (!MASKED_BUILD<NW>) { const unsigned a = look_ahead_reg(p, cnt, s, REVERSE); if (a != 2) return a != 0; }
			return look_ahead(p, cnt, s, REVERSE, FP_TRIM, sc);
		};
		if (!(dbg_skip & 1u)) {
		if (!la(v)) { result[r] = RR_BLUNT_END; return; }
		Vtx<NW> lastv;
		lastv.s = batch_kmer<NW>(b, r, nk - 1, k);
		vtx_rehash(p, lastv);
		vtx_revcomp(p, lastv);
		if (!la(lastv)) { result[r] = RR_BLUNT_END; return; }
		}
		if (dbg_skip & 2u) { result[r] = RR_NOT_SOLID; return; }
		// allKmersInBloom(seq, solidKmerSet), then allKmersInBloom(seq, assembledKmerSet) against the
		// snapshot (bloom-dbg.h:58-77,816-828).  One sweep over the k-mers, CLS_GROUP of them per
		// round: their 2 x H probes go out together, so a round costs one memory latency instead of
		// 2 x CLS_GROUP.  The verdicts are those of the two separate loops: not solid wins over
		// everything later, "visited" only counts for an entirely solid read.
		v = first_v;
		bool solid = true, visited = true, covered = false;
		const uint64_t rwoff = b.woff[r];
		uint32_t at = 0; // v is the vertex of k-mer `at` of the read
		// the archive's entry for the k-mer the sweep stands at (ContigArchive): fetched a round ahead, with the probes of the group before
		uint64_t en = arc.seq && !cov0 ? arc.tab[arc_slot(vtx_hash(p, v), arc.mask)] : 0;
		if (cov0) covered = true;
		for (uint32_t j0 = cov0; j0 < nk && solid;) {
			if (at != j0) { // (after a run the archive answered for: the hashes start over)
				v.s = batch_kmer<NW>(b, r, j0, k); vtx_rehash(p, v);
				at = j0;
				en = arc.tab[arc_slot(vtx_hash(p, v), arc.mask)];
			}
			if (en && !((en ^ vtx_hash(p, v)) >> ARC_POS_BITS)) {
				// the k-mers from here on that lie on a contig some earlier batch committed: solid and visited, no probe
				const uint32_t nc = arc_cover(arc, en & ((1ULL << ARC_POS_BITS) - 1), b.words, rwoff, L, j0, k);
				if (nc) { j0 += nc; covered = true; continue; }
			}
			uint64_t h[CLS_GROUP];
			// the group's incoming bases sit in one packed word or two: both read here, not a load per base under a condition
			// (one base more than the group's: the k-mer after it is hashed for the archive's next entry)
			const uint32_t i0 = j0 + k - 1, i1 = (i0 + CLS_GROUP < L ? i0 + CLS_GROUP : L - 1);
			const uint64_t two = (uint64_t)b.words[rwoff + (i0 >> 4)] | ((uint64_t)b.words[rwoff + (i1 >> 4)] << 32);
			auto base_at = [&](uint32_t bi) -> unsigned { return (unsigned)(two >> (((bi >> 4) != (i0 >> 4) ? 32u : 0u) + 2u * (bi & 15u))) & 3u; };
#pragma unroll
			for (uint32_t q = 0; q < CLS_GROUP; q++) {
				const uint32_t j = j0 + q;
				if (j < nk) {
					if (q) vtx_shift(p, v, SENSE, base_at(j + k - 1));
					h[q] = vtx_hash(p, v);
					at = j;
				} else {
					h[q] = h[0]; // (past the end: the group's first k-mer again)
				}
			}
			en = 0;
			if (j0 + CLS_GROUP < nk) {
				vtx_shift(p, v, SENSE, base_at(j0 + CLS_GROUP + k - 1));
				at = j0 + CLS_GROUP;
				if (arc.seq) en = arc.tab[arc_slot(vtx_hash(p, v), arc.mask)];
			}
			bool so = true, vi = true;
			for (unsigned base = 0; base < p.nh; base += 4) {
				uint8_t c[CLS_GROUP][4], w[CLS_GROUP][4];
#pragma unroll
				for (uint32_t q = 0; q < CLS_GROUP; q++) {
#pragma unroll
					for (unsigned t = 0; t < 4; t++) {
						const uint64_t pos = pos_i(p, h[q], base + t < p.nh ? base + t : 0u);
						if (both) {
							const unsigned two = (unsigned)(both[pos >> 2] >> (2u * (unsigned)(pos & 3u))) & 3u;
							c[q][t] = (two & 1u) ? 255 : 0;
							w[q][t] = (uint8_t)(two >> 1);
						} else {
							c[q][t] = (uint8_t)probe_c(p, cnt, pos);
							w[q][t] = (uint8_t)((vis[pos >> 3] >> (pos & 7)) & 1u);
						}
					}
				}
#pragma unroll
				for (uint32_t q = 0; q < CLS_GROUP; q++) {
#pragma unroll
					for (unsigned t = 0; t < 4; t++) { so = so & (c[q][t] >= p.kc); vi = vi & (w[q][t] != 0); }
				}
			}
			solid = solid & so; visited = visited & vi;
			j0 += CLS_GROUP;
		}
		if (arc.nreads) wave_count_add(arc.nreads, covered);
		if (!solid) { result[r] = RR_NOT_SOLID; return; }
		result[r] = visited ? (uint8_t)RR_ALL_KMERS_VISITED : RES_CANDIDATE;
	}
};

// A classification made against an older visited snapshot (see Engine::prefetch_classify) is
// brought up to date: BLUNT_END / NOT_SOLID do not depend on the snapshot and "all k-mers
// visited" is final once true, so only the candidates are tested again.
template <int NW>
struct FRefilter {
	Params p; Batch b; const uint8_t* vis; uint8_t* result;
	ABG_HDN void operator()(uint64_t r, uint32_t) const
	{
		if (result[r] != RES_CANDIDATE) return;
		const unsigned k = p.k;
		const uint32_t nk = b.len[r] - k + 1;
		Vtx<NW> v;
		v.s = batch_kmer<NW>(b, r, 0, k);
		vtx_rehash(p, v);
		for (uint32_t j0 = 0; j0 < nk; j0 += CLS_GROUP) {
			uint64_t h[CLS_GROUP];
#pragma unroll
			for (uint32_t q = 0; q < CLS_GROUP; q++) {
				const uint32_t j = j0 + q;
				if (j < nk) {
					if (j) vtx_shift(p, v, SENSE, batch_base(b, r, j + k - 1));
					h[q] = vtx_hash(p, v);
				} else {
					h[q] = h[0];
				}
			}
			bool vi = true;
#pragma unroll
			for (uint32_t q = 0; q < CLS_GROUP; q++) vi = vi & visited_contains(p, vis, h[q]);
			if (!vi) return;
		}
		result[r] = (uint8_t)RR_ALL_KMERS_VISITED;
	}
};

// The guide of the walkers' bulk steps (Guide, abg_walk.h): every k-mer of every `stride`-th read
// that may be solid (the first of its H counters says so: one probe drops nearly all k-mers with a
// sequencing error) leaves a hint "this k-mer is k-mer j of the read at word offset woff" in the
// slot of its canonical hash.  Plain stores: whichever read wrote last serves as the guide.
constexpr uint32_t GUIDE_READS_PER_ITEM = 4; // FGuideBuild: sampled reads a wavefront takes side by side, 16 lanes each (a 150 bp read keeps 11 lanes busy with runs of 8 k-mers)
template <int NW>
struct FGuideBuild {
	Params p; Batch b; const uint8_t* cnt; uint64_t* tab; uint64_t mask; uint32_t stride; uint64_t nsampled;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		const uint32_t per = nlanes >= GUIDE_READS_PER_ITEM ? nlanes / GUIDE_READS_PER_ITEM : nlanes; // lanes per read
		const uint32_t sub = lane / per, nsub = nlanes / per, sl = lane - sub * per;
		for (uint32_t q = sub; q < GUIDE_READS_PER_ITEM; q += nsub) {
			const uint64_t si = i * GUIDE_READS_PER_ITEM + q;
			if (si >= nsampled) break;
			one_read(si * stride, sl, per);
		}
	}
	ABG_HDN void one_read(uint64_t r, uint32_t lane, uint32_t nlanes) const
	{
		const uint32_t L = b.len[r];
		if (L < p.k) return;
		const uint32_t nk = L - p.k + 1;
		const uint64_t woff = b.woff[r];
		if (nk > GUIDE_MAX_NK || woff > GUIDE_MAX_WOFF) return;
		if (!p.mask) {
			// (without a spaced seed the key is the canonical hash itself: a lane rolls along a run of
			// consecutive k-mers instead of hashing every one of them from scratch -- 24 -> 7 ms per configs[1] step;
			// the run's probes go out together, then the hints of the k-mers that passed)
			constexpr uint32_t RUN = 8;
			for (uint32_t j0 = lane * RUN; j0 < nk; j0 += nlanes * RUN) {
				const uint32_t j1 = j0 + RUN < nk ? j0 + RUN : nk;
				uint64_t hm[RUN];
				for (uint32_t q = 0; q < RUN; q++) hm[q] = 0;
				kmer_hash_run(p, [&](unsigned q) { return batch_base(b, r, q); }, j0, j1, [&](uint32_t j, uint64_t h) { hm[j - j0] = h; });
				unsigned c[RUN];
#pragma unroll
				for (uint32_t q = 0; q < RUN; q++) c[q] = probe_c(p, cnt, pos_i(p, hm[j0 + q < j1 ? q : 0u], 0));
#pragma unroll
				for (uint32_t q = 0; q < RUN; q++)
					if (j0 + q < j1 && c[q] >= p.kc) tab[guide_slot(hm[q], mask)] = guide_pack(woff, j0 + q, nk, guide_tag(hm[q]));
			}
			return;
		}
		for (uint32_t j = lane; j < nk; j += nlanes) {
			const Kmer<NW> s = window_kmer<NW>(b.words, woff, j, p.k);
			uint64_t fh, rh;
			kmer_hashes(s, p.k, fh, rh);
			const uint64_t hm = rh < fh ? rh : fh; // the key: the walkers' rolling state, whatever the seed
			uint64_t df = 0, dr = 0;
			if constexpr (MASKED_BUILD<NW>) masked_terms(p, s, df, dr);
			const uint64_t fs = fh ^ df, rs = rh ^ dr;
			if (probe_c(p, cnt, pos_i(p, rs < fs ? rs : fs, 0)) < p.kc) continue;
			tab[guide_slot(hm, mask)] = guide_pack(woff, j, nk, guide_tag(hm));
		}
	}
};

constexpr uint32_t WALK_FAST_WITH_CACHE = 16384; // fast memory of at least this size ends with a MaskCache
template <int NW>
struct FWalk { // one walker per item; `list` selects the candidates to walk
	WalkEnv<NW> e; const uint32_t* list;
	// fast: fast_bytes of memory private to this walker (LDS on the device); coop: called by all
	// 64 lanes of a wavefront in lock step (HIP backend) or by one thread (serial)
	ABG_HDN void operator()(uint64_t i, uint32_t slot, void* fast, uint32_t fast_bytes, bool coop)
	{
		// The environment goes to fast memory as well (see walk_read): handed by reference to
		// out-of-line functions it would otherwise be a per-lane copy in scratch.  Every lane of a
		// cooperative caller stores the same values.
		WalkEnv<NW>* env = (WalkEnv<NW>*)fast;
		const uint32_t a = (uint32_t)((sizeof(WalkEnv<NW>) + 15) & ~15ull);
		*env = e;
		// the tail of the fast memory is the walkers' neighbour-mask cache (MaskCache): the launcher
		// zeroed its valid flags, and it carries over from one walker of this slot to the next
		env->mcache = nullptr;
		if (fast_bytes >= WALK_FAST_WITH_CACHE) {
			fast_bytes -= (uint32_t)sizeof(MaskCache);
			env->mcache = (MaskCache*)((char*)fast + fast_bytes);
		}
		env->fast = (char*)fast + a;
		env->fast_bytes = fast_bytes - a;
		env->coop = coop;
		walk_read<NW>(*env, list[i], slot);
	}
};

// (contig sequences hold code 4 = 'N' in columns no '1' of the spaced seed covers: never read)
template <int NW>
ABG_HDN uint64_t seq_kmer_hash(const Params& p, const uint8_t* seq, uint64_t j)
{
	return scratch_hash(p, [&](unsigned i) { return (unsigned)seq[j + i]; });
}
template <int NW>
ABG_HDN uint64_t read_kmer_hash(const Params& p, const Batch& b, uint64_t r, uint32_t j)
{
	return scratch_hash(p, [&](unsigned i) { return batch_base(b, r, j + i); });
}
constexpr uint32_t PREP_RUN = 8; // consecutive k-mers per lane
ABG_HD uint64_t dup_mix(uint64_t x) { x ^= x >> 31; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 29; x *= 0x9E3779B97F4A7C15ULL; x ^= x >> 32; return x | 1; }
template <int NW>
struct FReadPrep { // canonical hashes of the candidates' read k-mers (one wave per candidate)
	Params p; Batch b; const uint32_t* cand_read; const uint64_t* rkoff; uint64_t* rkh; uint32_t first;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		uint32_t c = first + (uint32_t)i;
		uint64_t r = cand_read[c];
		uint32_t nk = b.len[r] - p.k + 1;
		uint64_t* out = rkh + rkoff[c];
		for (uint32_t j0 = lane * PREP_RUN; j0 < nk; j0 += nlanes * PREP_RUN)
			kmer_hash_run(p, [&](unsigned q) { return batch_base(b, r, q); }, j0, j0 + PREP_RUN < nk ? j0 + PREP_RUN : nk,
			    [&](uint32_t j, uint64_t h) { out[j] = h; });
	}
};
template <int NW>
struct FContigPrep { // per contig record: k-mer hashes for the commit and (with_cov) Sum minCount
	Params p; const uint8_t* cnt; ContigRec* recs; uint32_t first; const uint8_t* pool; uint64_t* kh;
	uint32_t with_cov; // 0: the parallel commit sums the coverage of the contigs it inserts (FPcApply) -- most records are redundant copies
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		ContigRec& rec = recs[first + i];
		const uint8_t* seq = pool + rec.seq_off;
		uint64_t* out = kh + rec.seq_off;
		uint32_t cnk = rec.len - p.k + 1;
		uint32_t cov = 0; // getSeqAbsoluteKmerCoverage (bloom-dbg.h:92-109): a pure function of the solid filter
		uint64_t fp = 0;  // (see link_duplicates)
		for (uint32_t j0 = lane * PREP_RUN; j0 < cnk; j0 += nlanes * PREP_RUN)
			kmer_hash_run(p, [&](unsigned q) { return (unsigned)seq[q]; }, j0, j0 + PREP_RUN < cnk ? j0 + PREP_RUN : cnk,
			    [&](uint32_t j, uint64_t h) { out[j] = h; fp += dup_mix(h); if (with_cov) cov += solid_min_count(p, cnt, h); });
		if (cov) atomic_add_u32(&rec.coverage, cov);
		if (fp) atomic_add_u64(&rec.fp, fp);
	}
};

// ---- copies of one contig among a batch's records (Engine::link_duplicates).  The candidates of a batch are walked side by
// side, nothing committed between them, so a unitig reached from several reads of the batch is recorded once per read: on
// configs[1] 166 M record k-mers for 31.7 M unitig k-mers.  The commit decides every record, and every decision about a long
// contig is made of one probe (and one time stamp) per k-mer and hash function.  A record D whose k-mers are exactly those of a
// record O of a LOWER candidate shares O's fate where O's is known: if O is inserted -- at an earlier position -- every bit of D
// is set before D's turn, so D is dropped; what was settled for O ahead of the commit (all k-mers visited already) holds for D.
// Only where O is not inserted (its read turned out visited, or it was dropped itself) is D looked at bit by bit.
struct FDupKeys { // sort key of every record: contigs too short for the bit test, and records already linked, never group
	const ContigRec* recs; uint32_t k; uint64_t* key; uint32_t* val;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const ContigRec& r = recs[i];
		key[i] = r.len < k + FP_TRIM - 1 ? ~0ULL - i : dup_mix(r.fp ^ ((uint64_t)r.len << 40));
		val[i] = (uint32_t)i;
	}
};
struct FDupLink { // one sorted position per item: the lowest-candidate record of its group of equal keys
	ContigRec* recs; const uint64_t* key; const uint32_t* val; uint64_t n; uint32_t first_new; // records >= first_new are linked
	ABG_HD void operator()(uint64_t q, uint32_t) const
	{
		const uint32_t me = val[q];
		if (me < first_new) return;
		const uint64_t k = key[q];
		uint64_t g = q;
		for (unsigned s = 0; s < 64 && g > 0 && key[g - 1] == k; s++) g--;
		uint32_t best = REC_END, best_cand = recs[me].cand;
		for (unsigned s = 0; s < 128 && g < n && key[g] == k; s++, g++) {
			const uint32_t o = val[g];
			if (o != me && recs[o].cand < best_cand && recs[o].len == recs[me].len && recs[o].fp == recs[me].fp) { best = o; best_cand = recs[o].cand; }
		}
		recs[me].dup_of = best;
	}
};
struct FDupVerify { // one wave per record: the link holds only if the two hash sequences are equal, read the same or the opposite way
	ContigRec* recs; const uint64_t* kh; uint32_t k; uint32_t first_new;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		ContigRec& d = recs[first_new + i];
		const uint32_t o = d.dup_of;
		if (o == REC_END) return;
		const uint64_t* a = kh + d.seq_off; const uint64_t* b = kh + recs[o].seq_off;
		const uint32_t nk = d.len - k + 1;
		bool fwd = true, rev = true;
		for (uint32_t j = lane; j < nk; j += nlanes) { const uint64_t x = a[j]; fwd = fwd & (x == b[j]); rev = rev & (x == b[nk - 1 - j]); }
		fwd = wave_all_lanes(fwd, nlanes); rev = wave_all_lanes(rev, nlanes);
		if (!fwd && !rev && lane == 0) d.dup_of = REC_END;
	}
};
// Settles, in parallel and against the current visited snapshot, what the ordered commit
// would otherwise test one by one: a candidate read / a contig whose k-mers are ALL visited
// already stays so (the visited set only grows), so "read is visited" and "contig is
// redundant" are final for them.  flags: bit 0 of read_flag[c] = read entirely visited;
// rec.pre_redundant = contig entirely visited (long contigs only: short ones use the exact
// contigEndKmers rule, bloom-dbg.h:576-584).
// Partitioned run (part_c != NULL): a rank looks at the bits of its own range [lo, lo + span) only
// -- every rank holds the whole visited filter, but the tests are random reads and R ranks would
// each make all of them -- and leaves one byte per candidate (part_c) and per record (part_r,
// indexed like recs) that an all_reduce(MIN) turns into the verdicts; FPreCommitFin applies them.
ABG_HD bool visited_contains_owned(const Params& p, const uint8_t* __restrict__ vis, uint64_t h, uint64_t lo, uint64_t span)
{
	bool ok = true;
	if (p.nh <= 4) { // (the four loads together: see pc_bit_before)
		uint64_t q4[4]; uint8_t b4[4];
#pragma unroll
		for (unsigned i = 0; i < 4; i++) { q4[i] = pos_i(p, h, i < p.nh ? i : 0u); b4[i] = vis[q4[i] >> 3]; }
#pragma unroll
		for (unsigned i = 0; i < 4; i++) if (i < p.nh && q4[i] - lo < span) ok = ok & (((b4[i] >> (q4[i] & 7)) & 1u) != 0);
		return ok;
	}
	for (unsigned i = 0; i < p.nh; i++) {
		const uint64_t q = pos_i(p, h, i);
		if (q - lo < span) ok = ok & (((vis[q >> 3] >> (q & 7)) & 1u) != 0);
	}
	return ok;
}
template <int NW>
struct FPreCommit {
	Params p; Batch b; const uint32_t* cand_read; const uint32_t* status; const uint32_t* first_rec;
	ContigRec* recs; const uint8_t* vis; const uint64_t* kh; const uint64_t* rkh; const uint64_t* rkoff;
	uint8_t* read_flag; uint32_t first;
	uint64_t lo, span; uint8_t* part_c; uint8_t* part_r; // (whole filter: 0, ~0, NULL, NULL)
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		uint32_t c = first + (uint32_t)i;
		uint64_t r = cand_read[c];
		uint32_t nk = b.len[r] - p.k + 1;
		bool all = true;
		for (uint32_t j = lane; j < nk; j += nlanes) all = all & visited_contains_owned(p, vis, rkh[rkoff[c] + j], lo, span);
		all = wave_all_lanes(all, nlanes);
		if (part_c) {
			if (lane == 0) part_c[i] = all ? 1 : 0;
			if (status[c] != WS_COMPLETE) return;
		} else {
			if (lane == 0) read_flag[c] = all ? 1 : 0;
			if (all || status[c] != WS_COMPLETE) return;
		}
		for (uint32_t ri = first_rec[c]; ri != REC_END; ri = recs[ri].next) {
			ContigRec& rec = recs[ri];
			uint32_t cnk = rec.len - p.k + 1;
			if (rec.len < p.k + FP_TRIM - 1 || rec.pre_redundant) continue;
			// (a verified copy of a record this very call tests -- a lower candidate's, same k-mers, same snapshot -- takes that
			// one's answer afterwards, FPreCommitCopies: four fifths of a batch's record k-mers are such copies)
			if (rec.dup_of != REC_END && recs[rec.dup_of].cand >= first) continue;
			bool red = true;
			for (uint32_t j = lane; j < cnk; j += nlanes) red = red & visited_contains_owned(p, vis, kh[rec.seq_off + j], lo, span);
			red = wave_all_lanes(red, nlanes);
			if (part_r) { if (lane == 0) part_r[ri] = red ? 1 : 0; }
			else if (lane == 0 && red) rec.pre_redundant = 1;
		}
	}
};
struct FPreCommitCopies { // one candidate per item, after FPreCommit: the copies take their original's answer ("entirely visited" is a fact about the k-mer set)
	const uint32_t* status; const uint32_t* first_rec; ContigRec* recs; const uint8_t* read_flag; uint32_t first; uint32_t k;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t c = first + (uint32_t)i;
		if (read_flag[c] || status[c] != WS_COMPLETE) return;
		for (uint32_t ri = first_rec[c]; ri != REC_END; ri = recs[ri].next) {
			ContigRec& rec = recs[ri];
			if (rec.len < k + FP_TRIM - 1 || rec.pre_redundant || rec.dup_of == REC_END) continue;
			const ContigRec& o = recs[rec.dup_of];
			// (an original whose read turned out visited was not tested: its copy then stays "not settled" and is decided bit by bit)
			if (o.cand >= first && o.pre_redundant) rec.pre_redundant = 1;
		}
	}
};
struct FPreCommitFin { // the combined verdicts of a partitioned FPreCommit, one candidate per item
	const uint32_t* status; const uint32_t* first_rec; ContigRec* recs; uint8_t* read_flag; uint32_t first;
	uint32_t k; const uint8_t* part_c; const uint8_t* part_r;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t c = first + (uint32_t)i;
		const bool all = part_c[i] != 0;
		read_flag[c] = all ? 1 : 0;
		if (all || status[c] != WS_COMPLETE) return;
		for (uint32_t ri = first_rec[c]; ri != REC_END; ri = recs[ri].next) {
			ContigRec& rec = recs[ri];
			if (rec.len < k + FP_TRIM - 1 || rec.pre_redundant) continue;
			if (part_r[ri]) rec.pre_redundant = 1;
		}
	}
};

// Which candidates without a result will still be unvisited at their turn?  A read is
// predicted "covered" when each of its k-mers is already visited or lies in the territory
// of a lower-numbered walker that completed.  Mispredictions only cost time: the ordered
// commit stops at a needed candidate without a result and it is then walked (`force`).
template <int NW>
struct FPredict {
	Params p; Batch b; const uint32_t* cand_read; const uint32_t* status; const uint8_t* vis;
	const uint64_t* rkoff; const uint64_t* rkh;
	uint32_t* need_list; uint32_t* need_n;
	uint32_t first; uint32_t force;
	uint32_t rank, world; // partitioned run: need_n[1] counts every needed candidate, the list holds the ones this rank walks
	ABG_HDN void operator()(uint64_t i, uint32_t) const
	{
		uint32_t c = first + (uint32_t)i;
		if (status[c] == WS_COMPLETE) return;
		uint64_t r = cand_read[c];
		uint32_t nk = b.len[r] - p.k + 1;
		bool covered = c != force;
		for (uint32_t j = 0; covered && j < nk; j++) {
			uint64_t hm = rkh[rkoff[c] + j];
			if (visited_contains(p, vis, hm)) continue;
			covered = false;
		}
		if (!covered) {
			if (world > 1) {
				atomic_add_u32(need_n + 1, 1);
				if (c % world != rank) return;
			}
			need_list[atomic_add_u32(need_n, 1)] = c;
		}
	}
};
// Partitioned run: contig records gathered from the other ranks keep their rank-local numbering;
// bring pool offsets and record links into the merged numbering.  Records [g_rec + rbase[q],
// g_rec + rbase[q + 1]) came from rank q, whose pool block moved from g_pool to g_pool + pbase[q].
struct FRecFix {
	ContigRec* recs; uint32_t g_rec; uint32_t world;
	uint32_t rbase[MAX_RANKS + 1]; uint64_t pbase[MAX_RANKS + 1];
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		uint32_t q = 0;
		while (q + 1 < world && i >= rbase[q + 1]) q++;
		ContigRec& r = recs[g_rec + i];
		r.seq_off += pbase[q];
		if (r.next != REC_END) r.next += rbase[q];
	}
};
struct FFirstFix { // first record of the candidates this rank walked in the launch
	const uint32_t* list; uint32_t* first_rec; uint32_t delta;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t c = list[i];
		if (first_rec[c] != REC_END) first_rec[c] += delta;
	}
};
struct FAddU64 { uint64_t* a; uint64_t d; ABG_HD void operator()(uint64_t i, uint32_t) const { a[i] += d; } };
// ---- -g: outputGraph (bloom-dbg.h:1171-1242)
// trimSeq (bloom-dbg.h:399-451) on one clean segment: the longest run of consecutive k-mers the
// solid filter contains; the first such run wins a tie (a later one must be strictly longer).
template <int NW>
struct FTrimRun {
	Params p; Batch b; const uint8_t* cnt; uint32_t* best_start; uint32_t* best_len;
	ABG_HDN void operator()(uint64_t r, uint32_t) const
	{
		const unsigned k = p.k;
		const uint32_t nk = b.len[r] - k + 1;
		Vtx<NW> v;
		v.s = batch_kmer<NW>(b, r, 0, k);
		vtx_rehash(p, v);
		uint32_t bs = 0, bl = 0, cs = 0, cl = 0;
		for (uint32_t j = 0; j < nk; j++) {
			if (j) vtx_shift(p, v, SENSE, batch_base(b, r, j + k - 1));
			if (solid_contains(p, cnt, vtx_hash(p, v))) {
				if (!cl) cs = j;
				cl++;
			} else {
				if (cl > bl) { bl = cl; bs = cs; }
				cl = 0;
			}
		}
		if (cl > bl) { bl = cl; bs = cs; }
		best_start[r] = bs;
		best_len[r] = bl;
	}
};
// The breadth-first searches of outputGraph as ONE sequential walk (a single lane: the output IS
// the visiting order).  breadthFirstSearchImpl (Graph/BreadthFirstSearch.h:93-167) with a colour
// map shared by all searches: every search drains its queue, so "seen before" is all the map has
// to say, a start vertex seen before is skipped, and the order in which vertices leave the queue
// is the order in which they were discovered.  Per vertex leaving the queue one byte is recorded:
// bits 0-3 = out-edge to the successor ending in A, C, G, T exists (examine_edge), bits 4-7 = that
// successor was new (discover_vertex).  The host replays k-mer strings from that (abg_host.h).
// Resumable: when the node buffer or the vertex table is about to run out the walk stops, the
// host enlarges it and launches again.
struct GraphState {
	uint64_t s_next;   // next start vertex
	uint64_t head, count; // queue = nodes[head, count)
	uint64_t tab_used; // entries in the seen table
	uint64_t edges;
	uint32_t stop;     // 0 finished, 1 node buffer full, 2 table full
	uint32_t pad_;
};
template <int NW>
struct FGraphBfs {
	Params p; const uint8_t* cnt; Batch starts; uint64_t nstarts; WalkTab tab; uint64_t tab_limit;
	Vtx<NW>* nodes; uint64_t node_cap; uint8_t* ev; uint8_t* start_used; GraphState* st;
	ABG_HDN void operator()(uint64_t, uint32_t) const
	{
		GraphState s = *st;
		s.stop = 0;
		while (!s.stop) {
			while (s.head < s.count && !s.stop) {
				if (s.count + 4 > node_cap) { s.stop = 1; break; }
				if (s.tab_used + 4 > tab_limit) { s.stop = 2; break; }
				const Vtx<NW> u = nodes[s.head];
				unsigned e = 0;
				for (unsigned b = 0; b < 4; b++) { // out_edge_iterator, RollingBloomDBG.h:299-330
					Vtx<NW> w = neighbour_vertex(p, u, SENSE, b);
					if (!solid_contains(p, cnt, vtx_hash(p, w))) continue;
					e |= 1u << b;
					s.edges++;
					const VKey key = vtx_key(p, w);
					if (wt_find(tab, key, 0) != WT_EMPTY) continue;
					wt_insert(tab, key, 0, 0);
					s.tab_used++;
					nodes[s.count++] = w;
					e |= 16u << b;
				}
				ev[s.head++] = (uint8_t)e;
			}
			if (s.stop || s.s_next >= nstarts) break;
			if (s.count + 1 > node_cap) { s.stop = 1; break; }
			if (s.tab_used + 1 > tab_limit) { s.stop = 2; break; }
			Vtx<NW> v;
			v.s = batch_kmer<NW>(starts, s.s_next, 0, p.k);
			vtx_rehash(p, v);
			const VKey key = vtx_key(p, v);
			if (wt_find(tab, key, 0) != WT_EMPTY) {
				start_used[s.s_next] = 0; // not white: nothing printed, nothing queued (:117-131)
			} else {
				wt_insert(tab, key, 0, 0);
				s.tab_used++;
				nodes[s.count++] = v;
				start_used[s.s_next] = 1;
			}
			s.s_next++;
		}
		*st = s;
	}
};

// ---- ordered commit (outputContig, bloom-dbg.h:538-620), cooperative over T threads
struct CommitState {
	Counters counters;
	uint32_t break_at;   // first candidate that could not be committed
	uint32_t pad_;       // set when the contigEndKmers table overflowed (must not happen: it is grown ahead)
	uint64_t cend_count; // entries in the contigEndKmers table
};
struct FRehash { // move every entry of one vertex table into another (owner 0)
	WalkTab from, to; uint32_t* failed;
	ABG_HD void operator()(uint64_t s, uint32_t) const
	{
		uint64_t h = from.hmin[s];
		if (h == WT_EMPTY) return;
		VKey key; key.fh = h; key.rh = from.hmax[s];
		WalkTab t = to;
		if (wt_insert(t, key, 0, 0) == WT_FULL) *failed = 1;
	}
};
template <int NW>
struct CommitEnv {
	Params p; Batch b; uint32_t* vis32; uint32_t* both32 = nullptr; // (both32: the classification's copy of the visited bits, or NULL)
	const uint32_t* cand_read; const uint32_t* status; const uint32_t* first_rec;
	ContigRec* recs; const uint8_t* pool; uint8_t* result;
	const uint64_t* kh;      // hash of the k-mer starting at each pool offset (FContigPrep)
	const uint64_t* rkh;     // hashes of the candidates' read k-mers (FReadPrep)
	const uint64_t* rkoff;
	const uint8_t* read_flag; // FPreCommit: 1 = the read is entirely visited already
	WalkTab cend;            // contigEndKmers (bloom-dbg.h:992), owner 0
	CommitState* st;
	uint32_t* order;         // [rec_cap] records in commit order
	uint32_t* order_n;
};
ABG_HD bool visited_contains_coherent(const Params& p, const uint32_t* vis32, uint64_t h)
{
	bool ok = true;
	for (unsigned i = 0; i < p.nh; i++) {
		uint64_t q = pos_i(p, h, i);
		ok = ok & (((ld_coherent(&vis32[q >> 5]) >> (q & 31)) & 1u) != 0);
	}
	return ok;
}
// Per-candidate descriptor staged in fast memory by commit_candidates (one per thread,
// COMMIT_CHUNK at a time) so that the sequential loop does not pay a chain of dependent
// global loads per candidate.
struct CommitDesc {
	uint64_t r;          // read index
	uint64_t rkoff;      // offset of the read's k-mer hashes in rkh
	uint64_t seq_off;    // first contig record: pool offset
	uint32_t nk;         // read k-mers
	uint32_t status;     // WalkStatus
	uint32_t first;      // first contig record (REC_END: none)
	uint32_t len, next;  // first contig record: length, next record
	uint8_t visited;     // FPreCommit: the read is entirely visited already
	uint8_t pre_redundant; // first contig record
	uint8_t pad_[2];
};
constexpr uint32_t COMMIT_CHUNK = 256;

// The only inherently sequential part of PASS 2: in read order, decide "all k-mers
// visited?" for the read, then for each of its contigs the redundancy test and the
// insertion into the visited filter (outputContig, bloom-dbg.h:538-620).  All hashing and
// the coverage sums were done in parallel beforehand; this loop only tests and sets bits.
// Sync policy: tid(), nthreads(), barrier(), all(bool), bcast(uint32_t from tid 0),
// descs() -> CommitDesc[COMMIT_CHUNK] in memory shared by the threads.
template <int NW, class Sync>
ABG_HDN void commit_candidates(CommitEnv<NW>& e, uint32_t c_begin, uint32_t c_end, Sync& sy)
{
	const Params& p = e.p;
	const unsigned k = p.k;
	const uint32_t tid = sy.tid(), T = sy.nthreads();
	CommitDesc* descs = sy.descs();
	uint32_t c = c_begin;
	bool stop = false;
	for (uint32_t c0 = c_begin; c0 < c_end && !stop; c0 += COMMIT_CHUNK) {
		const uint32_t nchunk = c_end - c0 < COMMIT_CHUNK ? c_end - c0 : COMMIT_CHUNK;
		sy.barrier();
		for (uint32_t i = tid; i < nchunk; i += T) {
			CommitDesc d;
			uint32_t cc = c0 + i;
			d.r = e.cand_read[cc];
			d.nk = e.b.len[d.r] - k + 1;
			d.rkoff = e.rkoff[cc];
			d.status = e.status[cc];
			d.first = e.first_rec[cc];
			d.visited = e.read_flag[cc];
			d.seq_off = 0; d.len = 0; d.next = REC_END; d.pre_redundant = 0; d.pad_[0] = d.pad_[1] = 0;
			if (d.status == WS_COMPLETE && d.first != REC_END) {
				const ContigRec& rec = e.recs[d.first];
				d.seq_off = rec.seq_off; d.len = rec.len; d.next = rec.next; d.pre_redundant = rec.pre_redundant;
			}
			descs[i] = d;
		}
		sy.barrier();
		for (uint32_t i = 0; i < nchunk; i++) {
			c = c0 + i;
			const CommitDesc d = descs[i];
			const uint64_t r = d.r;
			const uint32_t nk = d.nk;
			// allKmersInBloom(seq, assembledKmerSet) at this read's turn (bloom-dbg.h:823)
			bool visited;
			if (d.visited) {
				visited = true; // settled ahead: every k-mer was visited before this commit started
			} else {
				bool mine = true;
				const uint64_t* rh = e.rkh + d.rkoff;
				for (uint32_t j = tid; j < nk; j += T)
					mine = mine & visited_contains_coherent(p, e.vis32, rh[j]);
				visited = sy.all(mine);
			}
			if (visited) {
				if (tid == 0) { e.result[r] = RR_ALL_KMERS_VISITED; e.st->counters.visited_reads++; }
				continue;
			}
			if (d.status != WS_COMPLETE) { stop = true; break; }
			if (tid == 0) e.result[r] = RR_GENERATED_CONTIGS;
			uint32_t ri = d.first;
			uint64_t seq_off = d.seq_off;
			uint32_t len = d.len, next = d.next;
			uint32_t pre = d.pre_redundant;
			while (ri != REC_END) {
				ContigRec& rec = e.recs[ri];
				const uint8_t* seq = e.pool + seq_off;
				const uint64_t* ch = e.kh + seq_off;
				const uint32_t cnk = len - k + 1;
				uint32_t redundant = 0;
				if (len < k + FP_TRIM - 1) {
					// short contigs: exact set of canonical end k-mers (bloom-dbg.h:576-584)
					if (tid == 0) {
						VKey k1 = canonical_end_key(p, seq), k2 = canonical_end_key(p, seq + len - k);
						if (wt_find(e.cend, k1, 0) != WT_EMPTY && wt_find(e.cend, k2, 0) != WT_EMPTY) {
							redundant = 1;
						} else {
							int a = wt_insert(e.cend, k1, 0, 0), bb = wt_insert(e.cend, k2, 0, 0);
							if (a == WT_FULL || bb == WT_FULL) redundant = 2; // table overflow: reported
							e.st->cend_count += (a == WT_NEW) + (bb == WT_NEW);
						}
					}
					redundant = sy.bcast(redundant);
				} else if (pre) {
					redundant = 1; // settled ahead of the commit
				} else {
					bool all = true;
					for (uint32_t j = tid; j < cnk; j += T)
						all = all & visited_contains_coherent(p, e.vis32, ch[j]);
					redundant = sy.all(all) ? 1u : 0u;
				}
				if (redundant == 2) { if (tid == 0) e.st->pad_ = 1; redundant = 0; }
				if (!redundant) {
					// addKmersToBloom (bloom-dbg.h:79-90)
					for (uint32_t j = tid; j < cnk; j += T) {
						uint64_t h = ch[j];
						for (unsigned q = 0; q < p.nh; q++) {
							uint64_t pos = pos_i(p, h, q);
							atomic_or_u32(&e.vis32[pos >> 5], 1u << (pos & 31));
							both_set(e.both32, pos);
						}
					}
				}
				if (tid == 0) {
					rec.redundant = (uint8_t)redundant;
					if (!redundant) {
						rec.contig_id = e.st->counters.contig_id++;
						e.st->counters.bases_assembled += len;
					}
					e.order[(*e.order_n)++] = ri;
				}
				if (!redundant) sy.barrier(); // publishes the inserted bits to the next test
				// next record of this read (fields from global memory: reads rarely have more than 2-3)
				ri = next;
				if (ri != REC_END) {
					const ContigRec& nr = e.recs[ri];
					seq_off = nr.seq_off; len = nr.len; next = nr.next; pre = nr.pre_redundant;
				}
			}
		}
		if (!stop) c = c0 + nchunk;
	}
	if (tid == 0) e.st->break_at = c;
	sy.barrier();
}

// ---- parallel form of the ordered commit ---------------------------------------------
// The sequential loop above decides, in read order: read c is skipped when all its k-mers are
// visited; else each of its contigs is dropped when all its k-mers are visited (short ones: when
// both end k-mers are in contigEndKmers) and otherwise inserted.  Every decision depends only
// on the insertions made EARLIER in that order, so the decisions are the unique fixed point of
//     T[bit]   = earliest commit position of an inserted contig holding the bit
//     visited  = every bit of the read/contig was set before the range, or T[bit] < own position
//     inserted = read not visited, contig not dropped
// and can be found by iterating from "every contig is inserted": compute T with one atomicMin
// per (k-mer, hash), re-decide everything in parallel, repeat while a decision changed.  A
// contig that turns out to be dropped was, by definition, not the earliest setter of any of
// its bits, so removing it leaves T as it was: only a skipped read whose contigs would have
// been inserted (or a dropped short contig) moves T, and the iteration typically settles in
// one or two passes.  By induction on the commit position the fixed point is what the
// sequential loop computes.  Costs 4 bytes per filter bit for T; without that memory the
// engine uses the sequential kernel.
constexpr uint32_t T_NEVER = 0xFFFFFFFFu;
// T is not cleared between passes: a stamp carries the tag of the pass that wrote it in its
// high bits, newer passes have SMALLER tags (they win atomicMin against anything older), and a
// stamp with another tag reads as "never".  The array is cleared when the tags run out.
constexpr uint32_t T_TIME_BITS = 22, T_TAGS = 1u << (32 - T_TIME_BITS); // positions < 2^22 (<= rec_cap records)
ABG_HD uint32_t t_stamp(uint32_t tag, uint32_t time) { return (tag << T_TIME_BITS) | time; }
// position recorded for this pass, or T_NEVER
ABG_HD uint32_t t_read(uint32_t v, uint32_t tag) { return (v >> T_TIME_BITS) == tag ? (v & ((1u << T_TIME_BITS) - 1)) : T_NEVER; }
struct ParCommit {
	Params p; Batch b;
	const uint32_t* cand_read; const uint32_t* status; const uint32_t* first_rec;
	ContigRec* recs; const uint8_t* pool; uint8_t* result;
	const uint64_t* kh; const uint64_t* rkh; const uint64_t* rkoff; const uint8_t* read_flag;
	const uint8_t* cnt8;   // the counting filter (FPcApply: coverage of the inserted contigs)
	uint32_t* vis32;       // the visited filter: its state before the range until FPcApply runs
	uint32_t* both32 = nullptr; // the classification's copy of the visited bits (FBothBuild), or NULL
	ContigArchive arc{};   // where FPcApply leaves a copy of every contig it inserts, for the classification of later batches (seq == NULL: nowhere)
	uint32_t* T;           // [filter bits] time stamps: (tag << T_TIME_BITS) | position, see t_stamp;
	                       // or NULL: the stamps live in the hash table below, keyed by bit position
	uint64_t* Tk;          // [Tmask + 1] bit positions (T_KEY_EMPTY: free)
	uint32_t* Tv;          // [Tmask + 1] their stamps
	uint64_t Tmask;
	uint32_t tag;          // tag of the current pass (smaller = newer, so a newer pass wins every atomicMin)
	WalkTab cend;          // contigEndKmers (bloom-dbg.h:992), owner 0
	WalkTab tcend;         // end k-mers of this range's inserted short contigs; meta = earliest position
	uint32_t* off;         // [n + 1] commit position of each candidate's first contig
	uint32_t* cnt;         // [n] scratch: records per candidate / ... (see the functors)
	uint32_t* cnt2;        // [n]
	uint64_t* cnt3;        // [n]
	uint8_t* active;       // [n] the read is not visited at its turn
	uint32_t* short_list;  // records of short contigs (unordered)
	VKey* short_keys;      // [2 per entry of short_list] their canonical end k-mers (FPcShortKeys)
	uint32_t* scal;        // [0] changed  [1] break candidate  [2] short_list length  [3] new contigEndKmers entries
	uint32_t c_begin, c_end, brk;
	// partitioned run: the stamps and the bit tests of [own_lo, own_lo + own_span) only (whole filter: 0, ~0),
	// and a byte per candidate / per record for what the ranks combine (see FPcDecideA)
	uint64_t own_lo, own_span;
	uint8_t* part_c; uint8_t* part_r;
};
// The stamp of a filter bit.  One 4-byte stamp per bit of the filter costs 4 bytes x m (7.6 GB for
// B=2G, 152 GB for B=40G); beyond par_commit_max_bytes the stamps of the bits a commit actually
// touches -- the k-mers of its contigs x H -- are kept in an open-addressing table instead.
constexpr uint64_t T_KEY_EMPTY = ~0ULL;
ABG_HD uint64_t pc_T_hash(const ParCommit& e, uint64_t pos)
{
	uint64_t x = pos * 0x9E3779B97F4A7C15ULL;
	x ^= x >> 32;
	return x & e.Tmask;
}
ABG_HD uint32_t* pc_T_slot(const ParCommit& e, uint64_t pos) // find or create
{
	if (e.T) return &e.T[pos];
	uint64_t s = pc_T_hash(e, pos);
	for (;;) {
		const uint64_t cur = cas_u64(&e.Tk[s], T_KEY_EMPTY, pos);
		if (cur == T_KEY_EMPTY || cur == pos) return &e.Tv[s];
		s = (s + 1) & e.Tmask;
	}
}
ABG_HD uint32_t pc_T_get(const ParCommit& e, uint64_t pos) // the raw stamp, or "never"
{
	if (e.T) return e.T[pos];
	uint64_t s = pc_T_hash(e, pos);
	for (;;) {
		const uint64_t cur = e.Tk[s];
		if (cur == pos) return e.Tv[s];
		if (cur == T_KEY_EMPTY) return T_NEVER;
		s = (s + 1) & e.Tmask;
	}
}
ABG_HD bool pc_bit_before(const ParCommit& e, uint64_t h, uint32_t time)
{
	bool ok = true;
	if (e.T && e.p.nh <= 4) {
		// (a stamp per filter bit, up to four hash functions: the eight loads together, none under a condition -- in the loop below
		// each is a round trip of its own)
		uint64_t pos[4]; uint32_t vw[4], tw[4];
#pragma unroll
		for (unsigned q = 0; q < 4; q++) { pos[q] = pos_i(e.p, h, q < e.p.nh ? q : 0u); vw[q] = e.vis32[pos[q] >> 5]; tw[q] = e.T[pos[q]]; }
#pragma unroll
		for (unsigned q = 0; q < 4; q++) {
			if (q >= e.p.nh || pos[q] - e.own_lo >= e.own_span) continue;
			ok = ok & ((((vw[q] >> (pos[q] & 31)) & 1u) != 0) | (t_read(tw[q], e.tag) < time));
		}
		return ok;
	}
	for (unsigned q = 0; q < e.p.nh; q++) {
		uint64_t pos = pos_i(e.p, h, q);
		if (pos - e.own_lo >= e.own_span) continue;
		bool set = ((e.vis32[pos >> 5] >> (pos & 31)) & 1u) != 0;
		ok = ok & (set | (t_read(pc_T_get(e, pos), e.tag) < time));
	}
	return ok;
}
struct FPcCount { // records of each candidate whose walk completed
	ParCommit e;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		uint32_t c = e.c_begin + (uint32_t)i, n = 0;
		if (e.status[c] == WS_COMPLETE)
			for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) n++;
		e.cnt[i] = n;
	}
};
struct FPcStamp { // commit positions, the optimistic first assumption, the list of short contigs
	ParCommit e;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		uint32_t c = e.c_begin + (uint32_t)i;
		if (e.status[c] != WS_COMPLETE) return;
		uint32_t t = e.off[i];
		for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next, t++) {
			ContigRec& rec = e.recs[ri];
			rec.time = t;
			const bool is_short = rec.len < e.p.k + FP_TRIM - 1;
			rec.ins = (!e.read_flag[c] && (is_short || !rec.pre_redundant)) ? 1u : 0u;
			if (rec.ins && rec.dup_of != REC_END) {
				// a copy of a lower candidate's contig: dropped if that one is (assumed) inserted
				const ContigRec& o = e.recs[rec.dup_of];
				if (!e.read_flag[o.cand] && !o.pre_redundant) rec.ins = 0;
			}
			rec.ins_prev = rec.ins;
			if (is_short) e.short_list[atomic_add_u32(&e.scal[2], 1)] = ri;
		}
	}
};
struct FPcSnapshot { // ins as the pass before left it (the copies of a contig decide by their original's)
	ParCommit e;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t c = e.c_begin + (uint32_t)i;
		if (e.status[c] != WS_COMPLETE) return;
		for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) e.recs[ri].ins_prev = e.recs[ri].ins;
	}
};
struct FPcTimeMin { // T: one wave per candidate
	ParCommit e;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		uint32_t c = e.c_begin + (uint32_t)i;
		if (e.status[c] != WS_COMPLETE) return;
		for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) {
			const ContigRec& rec = e.recs[ri];
			if (!rec.ins) continue;
			const uint64_t* ch = e.kh + rec.seq_off;
			const uint32_t cnk = rec.len - e.p.k + 1;
			for (uint32_t j = lane; j < cnk; j += nlanes) {
				uint64_t h = ch[j];
				for (unsigned q = 0; q < e.p.nh; q++) {
					const uint64_t pos = pos_i(e.p, h, q);
					if (pos - e.own_lo < e.own_span) atomic_min_u32(pc_T_slot(e, pos), t_stamp(e.tag, rec.time));
				}
			}
		}
	}
};
// find-or-create an entry and lower its position (single caller: see FPcShort)
ABG_HD void wt_upsert_min(const WalkTab& t, const VKey& key, uint64_t time)
{
	uint64_t s = wt_slot(t, key, 0);
	for (uint64_t probes = 0; probes <= t.mask; probes++, s = (s + 1) & t.mask) {
		uint64_t cur = t.hmin[s];
		if (cur == WT_EMPTY) { t.hmin[s] = key.fh; t.hmax[s] = key.rh; t.meta[s] = time; return; }
		if (cur == key.fh && t.hmax[s] == key.rh) { if (time < t.meta[s]) t.meta[s] = time; return; }
	}
}
struct FPcShortKeys { // end k-mers of the short contigs, in parallel
	ParCommit e;
	ABG_HDN void operator()(uint64_t i, uint32_t) const
	{
		const ContigRec& rec = e.recs[e.short_list[i]];
		const uint8_t* seq = e.pool + rec.seq_off;
		e.short_keys[2 * i] = canonical_end_key(e.p, seq);
		e.short_keys[2 * i + 1] = canonical_end_key(e.p, seq + rec.len - e.p.k);
	}
};
// The end k-mers of the short contigs, by ONE thread (there are few; concurrent insertion of the
// same key would need a two-word atomic).  mode 0: positions of the ones assumed inserted go
// into tcend; mode 1 (after the decisions): the inserted ones before the break enter
// contigEndKmers for good.
struct FPcShort {
	ParCommit e; uint32_t mode;
	ABG_HDN void operator()(uint64_t, uint32_t) const
	{
		const uint32_t n = e.scal[2];
		uint32_t added = 0;
		for (uint32_t i = 0; i < n; i++) {
			const ContigRec& rec = e.recs[e.short_list[i]];
			if (!rec.ins) continue;
			if (mode == 1 && rec.cand >= e.brk) continue;
			const VKey k1 = e.short_keys[2 * i], k2 = e.short_keys[2 * i + 1];
			if (mode == 0) {
				wt_upsert_min(e.tcend, k1, rec.time);
				wt_upsert_min(e.tcend, k2, rec.time);
			} else {
				int a = wt_insert(e.cend, k1, 0, 0), b = wt_insert(e.cend, k2, 0, 0);
				added += (a == WT_NEW) + (b == WT_NEW);
				if (a == WT_FULL || b == WT_FULL) e.scal[4] = 1;
			}
		}
		if (mode == 1) e.scal[3] = added;
	}
};
ABG_HD bool pc_end_before(const ParCommit& e, const VKey& key, uint32_t time)
{
	if (wt_find(e.cend, key, 0) != WT_EMPTY) return true;
	uint64_t s = wt_find(e.tcend, key, 0);
	return s != WT_EMPTY && e.tcend.meta[s] < (uint64_t)time;
}
struct FPcDecide { // one wave per candidate: re-decide the read and its contigs against T
	ParCommit e;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		const uint32_t c = e.c_begin + (uint32_t)i;
		const unsigned k = e.p.k;
		bool visited = e.read_flag[c] != 0;
		if (!visited) {
			const uint32_t nk = e.b.len[e.cand_read[c]] - k + 1;
			const uint64_t* rh = e.rkh + e.rkoff[c];
			const uint32_t t0 = e.off[i];
			bool mine = true;
			for (uint32_t j = lane; j < nk; j += nlanes) mine = mine & pc_bit_before(e, rh[j], t0);
			visited = wave_all_lanes(mine, nlanes);
		}
		if (lane == 0) e.active[i] = visited ? 0 : 1;
		if (e.status[c] != WS_COMPLETE) return;
		bool changed = false;
		for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) {
			ContigRec& rec = e.recs[ri];
			uint32_t ins = 0;
			if (!visited) {
				bool red;
				if (rec.len < k + FP_TRIM - 1) {
					// short contigs: exact set of canonical end k-mers (bloom-dbg.h:576-584)
					const uint8_t* seq = e.pool + rec.seq_off;
					red = pc_end_before(e, canonical_end_key(e.p, seq), rec.time) &&
					      pc_end_before(e, canonical_end_key(e.p, seq + rec.len - k), rec.time);
				} else if (rec.pre_redundant) {
					red = true;
				} else if (rec.dup_of != REC_END && e.recs[rec.dup_of].ins_prev) {
					red = true; // its lower copy holds every one of its bits, earlier (ContigRec::dup_of)
				} else {
					const uint64_t* ch = e.kh + rec.seq_off;
					const uint32_t cnk = rec.len - k + 1;
					bool mine = true;
					for (uint32_t j = lane; j < cnk; j += nlanes) mine = mine & pc_bit_before(e, ch[j], rec.time);
					red = wave_all_lanes(mine, nlanes);
				}
				ins = red ? 0u : 1u;
			}
			// Does the change move T?  A long contig dropped because all its bits were set earlier
			// was not the earliest setter of any bit that matters, and neither is a contig of a
			// skipped read unless some unset bit carries its own position: then T, and with it every
			// other decision, stays as it is.  Anything else calls for another pass.
			if (ins != rec.ins) {
				bool moves = true;
				if (rec.ins && rec.len >= k + FP_TRIM - 1) {
					moves = false;
					if (visited) {
						const uint64_t* ch = e.kh + rec.seq_off;
						const uint32_t cnk = rec.len - k + 1;
						bool mine = false;
						for (uint32_t j = lane; j < cnk; j += nlanes)
							for (unsigned q = 0; q < e.p.nh; q++) {
								uint64_t pos = pos_i(e.p, ch[j], q);
								mine = mine | (t_read(pc_T_get(e, pos), e.tag) == rec.time && !((e.vis32[pos >> 5] >> (pos & 31)) & 1u));
							}
						moves = !wave_all_lanes(!mine, nlanes);
					}
				}
				changed = changed | moves;
			}
			if (lane == 0) rec.ins = ins;
		}
		if (changed && lane == 0) e.scal[0] = 1;
	}
};
// ---- FPcDecide of a partitioned run, in three steps around two small all_reduces.  A: every rank
// tests the bits it owns -- "the read is visited at its turn" into part_c[i], "every k-mer of the
// (long, not pre-redundant) contig was set earlier" into part_r[record]; all_reduce(MIN).  B: the
// decisions, identical on every rank; where FPcDecide asks whether a change moves T, the rank
// answers for its own bits into part_c[i]; all_reduce(MAX).  C: a moved T calls for another pass.
struct FPcDecideA {
	ParCommit e;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		const uint32_t c = e.c_begin + (uint32_t)i;
		const unsigned k = e.p.k;
		bool pv = true;
		if (!e.read_flag[c]) {
			const uint32_t nk = e.b.len[e.cand_read[c]] - k + 1;
			const uint64_t* rh = e.rkh + e.rkoff[c];
			const uint32_t t0 = e.off[i];
			bool mine = true;
			for (uint32_t j = lane; j < nk; j += nlanes) mine = mine & pc_bit_before(e, rh[j], t0);
			pv = wave_all_lanes(mine, nlanes);
		}
		if (lane == 0) e.part_c[i] = pv ? 1 : 0;
		if (e.status[c] != WS_COMPLETE) return;
		for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) {
			const ContigRec& rec = e.recs[ri];
			if (rec.len < k + FP_TRIM - 1 || rec.pre_redundant) continue;
			const uint64_t* ch = e.kh + rec.seq_off;
			const uint32_t cnk = rec.len - k + 1;
			bool mine = true;
			for (uint32_t j = lane; j < cnk; j += nlanes) mine = mine & pc_bit_before(e, ch[j], rec.time);
			mine = wave_all_lanes(mine, nlanes);
			if (lane == 0) e.part_r[ri] = mine ? 1 : 0;
		}
	}
};
struct FPcDecideB {
	ParCommit e;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		const uint32_t c = e.c_begin + (uint32_t)i;
		const unsigned k = e.p.k;
		const bool visited = e.read_flag[c] != 0 || e.part_c[i] != 0;
		bool moved_here = false; // (this rank's bits)
		bool changed = false;
		if (e.status[c] == WS_COMPLETE) {
			for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) {
				ContigRec& rec = e.recs[ri];
				uint32_t ins = 0;
				if (!visited) {
					bool red;
					if (rec.len < k + FP_TRIM - 1) {
						const uint8_t* seq = e.pool + rec.seq_off;
						red = pc_end_before(e, canonical_end_key(e.p, seq), rec.time) &&
						      pc_end_before(e, canonical_end_key(e.p, seq + rec.len - k), rec.time);
					} else if (rec.pre_redundant) {
						red = true;
					} else {
						red = e.part_r[ri] != 0;
					}
					ins = red ? 0u : 1u;
				}
				if (ins != rec.ins) {
					if (rec.ins && rec.len >= k + FP_TRIM - 1) {
						if (visited) {
							const uint64_t* ch = e.kh + rec.seq_off;
							const uint32_t cnk = rec.len - k + 1;
							bool mine = false;
							for (uint32_t j = lane; j < cnk; j += nlanes)
								for (unsigned q = 0; q < e.p.nh; q++) {
									uint64_t pos = pos_i(e.p, ch[j], q);
									if (pos - e.own_lo >= e.own_span) continue;
									mine = mine | (t_read(pc_T_get(e, pos), e.tag) == rec.time && !((e.vis32[pos >> 5] >> (pos & 31)) & 1u));
								}
							moved_here = moved_here | !wave_all_lanes(!mine, nlanes);
						}
					} else {
						changed = true;
					}
				}
				if (lane == 0) rec.ins = ins;
			}
		}
		if (lane == 0) {
			e.active[i] = visited ? 0 : 1;
			e.part_c[i] = moved_here ? 1 : 0;
			if (changed) e.scal[0] = 1;
		}
	}
};
struct FPcDecideC {
	ParCommit e;
	ABG_HD void operator()(uint64_t i, uint32_t) const { if (e.part_c[i]) e.scal[0] = 1; }
};
struct FPcBreak { // first candidate that is needed but has no result
	ParCommit e;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		uint32_t c = e.c_begin + (uint32_t)i;
		if (e.active[i] && e.status[c] != WS_COMPLETE) atomic_min_u32(&e.scal[1], c);
	}
};
struct FPcApply { // one wave per candidate before the break: results, visited bits, per-candidate totals
	ParCommit e;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		const uint32_t c = e.c_begin + (uint32_t)i;
		if (c >= e.brk) { if (lane == 0) { e.cnt[i] = 0; e.cnt2[i] = 0; e.cnt3[i] = 0; } return; }
		const uint64_t r = e.cand_read[c];
		uint32_t nrec = 0, nins = 0; uint64_t bases = 0;
		if (!e.active[i]) {
			if (lane == 0) e.result[r] = RR_ALL_KMERS_VISITED;
		} else {
			if (lane == 0) e.result[r] = RR_GENERATED_CONTIGS;
			for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) {
				ContigRec& rec = e.recs[ri];
				nrec++;
				if (lane == 0) rec.redundant = rec.ins ? 0 : 1;
				if (!rec.ins) continue;
				nins++; bases += rec.len;
				// addKmersToBloom (bloom-dbg.h:79-90), and getSeqAbsoluteKmerCoverage (bloom-dbg.h:92-109) of what is output
				const uint64_t* ch = e.kh + rec.seq_off;
				const uint32_t cnk = rec.len - e.p.k + 1;
				uint32_t cov = 0;
				// ... and the contig into the archive the classification of later batches reads (ContigArchive): the bases, then
				// every k-mer's position under its hash (plain stores, the last writer of a slot stays)
				uint64_t apos = 0;
				if (e.arc.seq) {
					if (lane == 0) apos = atomic_add_u64(e.arc.used, (uint64_t)rec.len + 1);
					apos = uni64<true>(apos);
					if (apos + rec.len + 1 > e.arc.cap) apos = 0; // (no room: the contig is not archived)
					else {
						const uint8_t* src = e.pool + rec.seq_off;
						for (uint32_t q = lane; q < rec.len; q += nlanes) e.arc.seq[apos + q] = src[q];
					}
				}
				for (uint32_t j = lane; j < cnk; j += nlanes) {
					uint64_t h = ch[j];
					if (apos) e.arc.tab[arc_slot(h, e.arc.mask)] = arc_entry(h, apos + j);
					unsigned mn = 255;
					if (e.cnt8 && e.p.nh <= 4) {
						// (the counters' loads together, then the bits: in the loop below a load waits behind the atomics before it)
						uint64_t pos[4]; unsigned c[4];
#pragma unroll
						for (unsigned q = 0; q < 4; q++) { pos[q] = pos_i(e.p, h, q < e.p.nh ? q : 0u); c[q] = e.cnt8[pos[q]]; }
#pragma unroll
						for (unsigned q = 0; q < 4; q++) {
							if (q >= e.p.nh) continue;
							atomic_or_u32(&e.vis32[pos[q] >> 5], 1u << (pos[q] & 31));
							both_set(e.both32, pos[q]);
							mn = c[q] < mn ? c[q] : mn;
						}
						cov += mn;
						continue;
					}
					for (unsigned q = 0; q < e.p.nh; q++) {
						uint64_t pos = pos_i(e.p, h, q);
						atomic_or_u32(&e.vis32[pos >> 5], 1u << (pos & 31));
						both_set(e.both32, pos);
						if (e.cnt8) { const unsigned c = e.cnt8[pos]; mn = c < mn ? c : mn; }
					}
					if (e.cnt8) cov += mn;
				}
				if (cov) atomic_add_u32(&rec.coverage, cov);
			}
		}
		if (lane == 0) { e.cnt[i] = nrec; e.cnt2[i] = nins; e.cnt3[i] = bases; }
	}
};
// Coverage of the inserted contigs on a sliced filter (the ranks hold a range of the counters each): a k-mer's minimum over the
// counters THIS rank owns, a byte per k-mer of the contig pool (255: none of them here); all_reduce(MIN); the sums.
struct FPcCover {
	ParCommit e; uint8_t* kmin; uint32_t sum;
	ABG_HDN void operator()(uint64_t i, uint32_t lane, uint32_t nlanes) const
	{
		const uint32_t c = e.c_begin + (uint32_t)i;
		if (c >= e.brk || !e.active[i]) return;
		for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) {
			ContigRec& rec = e.recs[ri];
			if (!rec.ins) continue;
			const uint64_t* ch = e.kh + rec.seq_off;
			const uint32_t cnk = rec.len - e.p.k + 1;
			uint32_t cov = 0;
			for (uint32_t j = lane; j < cnk; j += nlanes) {
				if (sum) { cov += kmin[rec.seq_off + j]; continue; }
				const uint64_t h = ch[j];
				unsigned mn = 255;
				for (unsigned q = 0; q < e.p.nh; q++) {
					const uint64_t pos = pos_i(e.p, h, q);
					if (pos - e.own_lo < e.own_span) { const unsigned v = e.cnt8[pos]; mn = v < mn ? v : mn; }
				}
				kmin[rec.seq_off + j] = (uint8_t)mn;
			}
			if (sum && cov) atomic_add_u32(&rec.coverage, cov);
		}
	}
};
struct FPcWrite { // commit order of the records and the contig ids (off = record offsets, cnt2 = id offsets)
	ParCommit e; uint32_t* order; uint32_t order_base; uint64_t id_base;
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint32_t c = e.c_begin + (uint32_t)i;
		if (c >= e.brk || !e.active[i]) return;
		uint32_t o = order_base + e.cnt[i];
		uint64_t id = id_base + e.cnt2[i];
		for (uint32_t ri = e.first_rec[c]; ri != REC_END; ri = e.recs[ri].next) {
			ContigRec& rec = e.recs[ri];
			order[o++] = ri;
			if (rec.ins) rec.contig_id = id++;
		}
	}
};

// ================================================================== Engine
// Backend concept:
//   void* alloc(size_t); void free(void*); void memset(void*, int, size_t);
//   void h2d(void*, const void*, size_t); void d2h(void*, const void*, size_t);
//   uint32_t max_slots();                       // upper bound on concurrent items of launch()
//   template<class F> void launch(uint64_t n, F f, const char* name);               // f(i, slot); 2^ITEM_GROUP_LOG2 consecutive items run as one wavefront (wave_rank)
//   template<class F> void launch_slots(uint64_t n, F f, uint32_t slots, const char* name);
//   template<class F> void launch_walkers(uint64_t n, F f, uint32_t slots, const char* name); // one item per wave, lane 0
//   template<class F> void launch_wave(uint64_t n, F f, const char* name);   // f(item, lane, nlanes): one item per wave
//   template<int NW> void launch_commit(CommitEnv<NW>, uint32_t c_begin, uint32_t c_end);
//   void launch_drain(InsertDrainEnv);          // one workgroup running insert_drain
//   template<class F> void launch_tiles(uint64_t n, F f, const char* name); // f(tile, fast memory [F::FAST bytes], sync): one workgroup per item
template <class BE>
class Engine {
  public:
	Engine(BE& be, const Config& cfg) : be_(be), cfg_(cfg)
	{
		uint64_t rem = cfg_.counters % 8; // CountingBloomFilter ctor, hpp:40-50
		m_ = rem ? cfg_.counters + 8 - rem : cfg_.counters;
		p_ = make_params(cfg_.k, cfg_.nh, cfg_.kc, cfg_.trim, m_);
		if (!cfg_.spaced_seed.empty()) {
			MaskTab* mt = new MaskTab;
			make_mask(cfg_.spaced_seed.c_str(), cfg_.k, *mt, p_);
			mask_d_ = (MaskTab*)be_.alloc(sizeof(MaskTab));
			be_.h2d(mask_d_, mt, sizeof(MaskTab));
			delete mt;
			p_.mask = mask_d_;
		}
		if (cfg_.cascade_levels) {
			// `abyss-bloom build -t rolling-hash -l N`: N bit filters of m_ bits (Bloom/bloom.cc:585-602)
			casc_.levels = cfg_.cascade_levels;
			casc_.level_words = (m_ + 31) / 32;
			casc_.bits = (uint32_t*)be_.alloc(casc_.levels * casc_.level_words * 4);
			be_.memset(casc_.bits, 0, casc_.levels * casc_.level_words * 4);
			cnt_ = (uint8_t*)be_.alloc(8);
			vis_bytes_ = 8;
		} else {
			// a filter beyond this device: the counters wait for the communicator, whose ranks keep a range each (attach_comm)
			const uint64_t dev = be_.device_mem_bytes();
			cnt_deferred_ = cfg_.slice_filter == 1 || (cfg_.slice_filter == 0 && dev && (double)m_ * 1.3 > (double)dev * 0.9);
			if (!cnt_deferred_) {
				// (slack: the shards of a partitioned run are gathered in equal chunks of roundUp64(m / ranks))
				cnt_ = (uint8_t*)be_.alloc(m_ + 64 * (MAX_RANKS + 1));
				be_.memset(cnt_, 0, m_);
			}
			vis_bytes_ = (m_ / 8 + 4 + 3) & ~3ull;
		}
		vis_ = (uint8_t*)be_.alloc(vis_bytes_);
		be_.memset(vis_, 0, vis_bytes_);
		cstate_ = (CommitState*)be_.alloc(sizeof(CommitState));
		be_.memset(cstate_, 0, sizeof(CommitState));
		scal_ = (uint64_t*)be_.alloc(64);
	}
	~Engine()
	{
		free_counters(); be_.free(vis_); if (both_) be_.free(both_); be_.free(cstate_); be_.free(scal_);
		free_archive();
		if (casc_.bits) be_.free(casc_.bits);
		if (mask_d_) be_.free(mask_d_);
		if (T_) be_.free(T_);
		if (Tk_) { be_.free(Tk_); be_.free(Tv_); }
		if (la_pool_c_) be_.free(la_pool_c_);
		if (la_pool_c2_) be_.free(la_pool_c2_);
		if (guide_tab_) be_.free(guide_tab_);
		if (guide_seen_) be_.free(guide_seen_);
		if (gtab_.hmin) free_tab(gtab_);
		free_shared();
		free_insert();
		free_walk();
		if (cend_.hmin) free_tab(cend_);
		if (memo_tab_.hmin) free_tab(memo_tab_);
		if (plane_) be_.free(plane_);
		if (wstats_) be_.free(wstats_);
	}
	// Back to the state right after construction -- empty filters, zero counters, empty
	// contigEndKmers -- without giving any memory back (claim tables and time stamps are
	// epoch-tagged and need no clearing).
	void reset()
	{
		if (casc_.bits) be_.memset(casc_.bits, 0, casc_.levels * casc_.level_words * 4);
		else if (sliced_) be_.memset(cnt_ + own_lo_, 0, own_span_);
		else if (cnt_) be_.memset(cnt_, 0, m_);
		be_.memset(vis_, 0, vis_bytes_);
		arc_valid_ = false; loaded_ops_ = 0;
		be_.memset(cstate_, 0, sizeof(CommitState));
		counters_ = Counters();
		stats_ = Stats();
		cnt_partial_ = false; cnt_loaded_ = false;
		memo_valid_ = false; plane_valid_ = false;
		last_rounds_ = 0;
		p2_batch_ = cfg_.p2_first_batch;
		last_candidates_ = 0;
		if (cend_.hmin) {
			be_.memset(cend_.hmin, 0xFF, (cend_.mask + 1) * 8);
			be_.memset(cend_.meta, 0xFF, (cend_.mask + 1) * 8);
		}
		cend_count_ = 0;
		if (wstats_) clear_wstats();
		ovf_seen_[0] = ovf_seen_[1] = 0;
		if (gtab_.hmin) { free_tab(gtab_); gtab_ = WalkTab{ nullptr, nullptr, nullptr, 0 }; gtab_used_ = 0; }
	}
	const Params& params() const { return p_; }
	uint64_t size() const { return m_; }
	// (a partitioned run leaves only the rank's own range current until the shards are gathered)
	uint8_t* counters_dev() { need_whole_filter("direct access to the counters"); gather_counters(); memo_valid_ = false; plane_valid_ = false; arc_valid_ = false; /* (the caller may write: the archived contigs' k-mers may no longer be solid) */ return cnt_; }

	// ---- partitioned multi-GPU run (include/abyss_amd.h, abg_comm): the counting filter is
	// range-partitioned by position over the ranks of a communicator during PASS 1 -- rank q owns
	// positions [q * chunk, (q + 1) * chunk) -- and all-gathered for PASS 2.  Every rank holds every
	// read (share_reads); the ops are hashed once and every rank settles the ones on its own counters
	// (insert_tiles_dist), the commit of PASS 2 tests and stamps the bits of its own range (commit_par).
	struct Comm {
		int rank = 0, world = 1;
		bool stream_ordered = false;
		void* user = nullptr;
		int (*all_gather_v)(void*, void*, const uint64_t*, const uint64_t*, void*) = nullptr;
		int (*all_reduce)(void*, void*, uint64_t, int32_t, int32_t, void*) = nullptr;
		int (*all_to_all_v)(void*, const void*, const uint64_t*, const uint64_t*, void*, const uint64_t*, const uint64_t*, void*) = nullptr;
	};
	enum { DT_U8 = 0, DT_U32 = 1, DT_U64 = 2, OP_SUM = 0, OP_MAX = 1, OP_MIN = 2 };
	bool attach_comm(const Comm& c)
	{
		if (c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world || casc_.bits) return false;
		uint64_t chunk = (m_ + c.world - 1) / c.world;
		chunk = (chunk + 63) & ~63ull;
		// (a sliced filter that holds counters cannot be laid out anew: this rank keeps [own_lo_, own_lo_ + own_span_) and nothing
		// else, and another rank or world would need counters that are not here -- export and import them instead)
		if (sliced_ && cnt_loaded_ && (c.rank != comm_.rank || c.world != comm_.world))
			fail_now(FAIL_INVAL, "a sliced filter that holds counters cannot take a communicator with another rank or world (abg_counters_export / _import move them)");
		const bool keep_window = sliced_ && c.rank == comm_.rank && c.world == comm_.world && own_chunk_ == chunk;
		comm_ = c;
		free_insert(); // (PASS 1's scratch is laid out for the partition it was made under: the next load makes it anew)
		own_lo_ = std::min<uint64_t>(m_, (uint64_t)c.rank * chunk);
		own_span_ = std::min<uint64_t>(m_, own_lo_ + chunk) - own_lo_;
		own_chunk_ = chunk;
		if (sliced_ || cnt_deferred_) {
			// the sliced filter: this rank's range of the counters and nothing else.  cnt_ stays the address of counter 0 -- every
			// kernel of the partitioned PASS 1 takes global positions and touches [own_lo_, own_lo_ + own_span_) only -- and PASS 2
			// probes the bit plane all ranks gather (ensure_plane), asks for coverage through an all-reduce (commit_par)
			if (!cfg_.solid_plane || (m_ & 63) || !cfg_.par_commit)
				fail_now(FAIL_INVAL, "a sliced filter needs the bit plane, the parallel commit and a multiple of 64 counters");
			if (!keep_window) { // (the same rank of the same world again: the window and what it holds stay)
				free_counters(); // (frees the window as it was made: win_lo_, not the new own_lo_)
				if (plane_) { be_.free(plane_); plane_ = nullptr; }
				win_total_ = (uint64_t)c.world * chunk + 64;
				win_span_ = chunk + 64;
				win_lo_ = own_lo_;
				cnt_ = (uint8_t*)be_.alloc_window(win_total_, win_lo_, win_span_);
				be_.memset(cnt_ + own_lo_, 0, win_span_);
				cnt_loaded_ = false;
			}
			sliced_ = true; cnt_deferred_ = false;
			cfg_.drain_threshold = 0; // (the drain of the last pending ops replays them on scratch copies of the other ranks' counters)
			memo_valid_ = false; plane_valid_ = false;
		}
		// ABG_FORCE_DIST=1: take the partitioned code path with a single rank too (its kernels,
		// compaction, merges and collectives, each of which is then an identity) -- for tests and for
		// measuring what the partitioned path itself costs
		const char* f = getenv("ABG_FORCE_DIST");
		force_dist_ = f && atoi(f) != 0;
		if (c.world > 1 && !comm_scaled_) {
			// R ranks walk a batch's candidates side by side, but a batch still ends with its slowest
			// walker: fewer, larger batches.  (Measured on one GPU with the 8x schedule: 7 batches
			// instead of 10 for config 2, 1.3x the candidates, 1.16x the walk work -- split R ways.)
			comm_scaled_ = true;
			cfg_.p2_first_batch *= (uint64_t)c.world;
			cfg_.p2_max_batch *= (uint64_t)c.world;
			cfg_.p2_starved *= (uint32_t)c.world;
			cfg_.p2_crowded *= (uint32_t)std::min(c.world, 4); // (a commit holds at most 2^22 contig records)
			p2_batch_ = cfg_.p2_first_batch;
		}
		return true;
	}
	bool dist() const { return comm_.world > 1 || force_dist_ || sliced_; }
	bool sliced() const { return sliced_; }
	uint64_t counter_bytes_held() const { return sliced_ ? win_span_ : cnt_ ? m_ : 0; }
	void need_counters() const
	{
		if (!cnt_) fail_now(FAIL_INVAL, "this filter does not fit the device: attach a communicator first, its ranks keep a range each (abg_attach_comm)");
	}
	void need_whole_filter(const char* what) const
	{
		need_counters();
		if (sliced_) fail_now(FAIL_INVAL, std::string(what) + " is not available on a sliced filter (each rank holds a range of the counters)");
	}
	void free_counters()
	{
		if (sliced_) be_.free_window(cnt_, win_total_, win_lo_, win_span_);
		else if (cnt_) be_.free(cnt_);
		cnt_ = nullptr; sliced_ = false;
	}
	// All-gather of the ranks' packed read sets (rank-major order) into buffers the engine keeps
	// until the next call: what every rank then hands to load_packed / assemble_packed.
	Batch share_reads(const Batch& loc)
	{
		const uint32_t R = (uint32_t)comm_.world, me = (uint32_t)comm_.rank;
		uint64_t w0 = 0, w1 = 0;
		if (loc.n) { be_.d2h(&w0, loc.woff, 8); be_.d2h(&w1, loc.woff + loc.n, 8); }
		std::vector<uint64_t> cnt(2 * R, 0);
		cnt[2 * me] = loc.n; cnt[2 * me + 1] = w1 - w0;
		host_all_reduce_sum(cnt.data(), 2 * R);
		std::vector<uint64_t> nb(R + 1, 0), wb(R + 1, 0);
		for (uint32_t q = 0; q < R; q++) { nb[q + 1] = nb[q] + cnt[2 * q]; wb[q + 1] = wb[q] + cnt[2 * q + 1]; }
		free_shared();
		sh_words_ = (uint32_t*)be_.alloc(wb[R] * 4 + 64);
		sh_woff_ = (uint64_t*)be_.alloc((nb[R] + 1) * 8);
		sh_len_ = (uint32_t*)be_.alloc(nb[R] * 4 + 4);
		sh_koff_ = (uint64_t*)be_.alloc((nb[R] + 1) * 8);
		if (loc.n) {
			be_.d2d(sh_words_ + wb[me], loc.words + w0, (w1 - w0) * 4);
			be_.d2d(sh_len_ + nb[me], loc.len, loc.n * 4);
			be_.d2d(sh_woff_ + nb[me], loc.woff, loc.n * 8);
			FAddU64 f{ sh_woff_ + nb[me], wb[me] - w0 };
			be_.launch(loc.n, f, "share_fix");
		}
		std::vector<uint64_t> c(R), d(R);
		for (uint32_t q = 0; q < R; q++) { c[q] = cnt[2 * q + 1] * 4; d[q] = wb[q] * 4; }
		c_all_gather_v(sh_words_, c.data(), d.data());
		for (uint32_t q = 0; q < R; q++) { c[q] = cnt[2 * q] * 4; d[q] = nb[q] * 4; }
		c_all_gather_v(sh_len_, c.data(), d.data());
		for (uint32_t q = 0; q < R; q++) { c[q] = cnt[2 * q] * 8; d[q] = nb[q] * 8; }
		c_all_gather_v(sh_woff_, c.data(), d.data());
		be_.h2d(sh_woff_ + nb[R], &wb[R], 8);
		// k-mer prefix sums, on the device (PASS 1 cuts its batches there too: cut_ranges)
		device_koff(sh_len_, nb[R], sh_koff_);
		return Batch{ sh_words_, sh_woff_, sh_len_, sh_koff_, nb[R] };
	}
	bool cascade_mode() const { return casc_.bits != nullptr; }
	uint32_t cascade_levels() const { return casc_.levels; }
	// The whole counter array to / from host memory (checkpoints, tests).  A sliced filter passes it rank by rank through one
	// spare chunk; every rank ends up with the whole array on the HOST.
	void counters_to_host(uint8_t* out)
	{
		need_counters();
		if (!sliced_) { be_.d2h(out, counters_dev(), m_); return; }
		uint8_t* tmp = (uint8_t*)be_.alloc(own_chunk_);
		std::vector<uint64_t> c(comm_.world), d(comm_.world, 0);
		for (int q = 0; q < comm_.world; q++) {
			std::fill(c.begin(), c.end(), 0);
			c[q] = own_chunk_;
			if (q == comm_.rank) be_.d2d(tmp, cnt_ + own_lo_, own_chunk_);
			c_all_gather_v(tmp, c.data(), d.data());
			const uint64_t lo = (uint64_t)q * own_chunk_;
			if (lo < m_) be_.d2h(out + lo, tmp, std::min<uint64_t>(own_chunk_, m_ - lo));
		}
		be_.free(tmp);
	}
	void counters_from_host(const uint8_t* in)
	{
		need_counters();
		cnt_loaded_ = true;
		if (!sliced_) { be_.h2d(counters_dev(), in, m_); return; }
		be_.h2d(cnt_ + own_lo_, in + own_lo_, own_span_);
		memo_valid_ = false; plane_valid_ = false; arc_valid_ = false;
	}
	uint8_t* cascade_level_dev(uint32_t l) { return (uint8_t*)(casc_.bits + (uint64_t)l * casc_.level_words); }
	const uint8_t* visited_dev_ro() const { return vis_; } // (for reading: an export leaves the archive of committed contigs as it is)
	uint8_t* visited_dev() { arc_valid_ = false; /* (the caller may write: what the archive of committed contigs says may no longer hold) */ return vis_; }
	uint64_t visited_bytes() const { return m_ / 8; }
	Counters counters() const { return counters_; }
	void set_counters(const Counters& c) { counters_ = c; }
	uint64_t last_insert_rounds() const { return last_rounds_; }

	void popcounts(uint64_t* nonzero, uint64_t* filtered)
	{
		need_counters();
		gather_counters();
		be_.memset(scal_, 0, 16);
		FPopcount f{ (const uint64_t*)(cnt_ + (sliced_ ? own_lo_ : 0)), p_.kc, scal_ };
		be_.launch((sliced_ ? own_span_ : m_) / 8, f, "popcount");
		uint64_t out[2];
		be_.d2h(out, scal_, 16);
		if (sliced_) host_all_reduce_sum(out, 2);
		*nonzero = out[0];
		*filtered = out[1];
	}

	// ---- -g: per clean segment, the longest run of solid k-mers (trimSeq)
	void trim_runs(const Batch& b, uint32_t* best_start_h, uint32_t* best_len_h)
	{
		if (!b.n) return;
		need_whole_filter("-g");
		gather_counters();
		uint32_t* bs = (uint32_t*)be_.alloc(b.n * 4);
		uint32_t* bl = (uint32_t*)be_.alloc(b.n * 4);
		dispatch_nw([&](auto nw) {
			FTrimRun<decltype(nw)::value> f{ p_, b, cnt_, bs, bl };
			be_.launch(b.n, f, "graph_trim");
		});
		be_.d2h(best_start_h, bs, b.n * 4);
		be_.d2h(best_len_h, bl, b.n * 4);
		be_.free(bs); be_.free(bl);
	}
	// ---- -g: the breadth-first searches from `starts` (sequences of k bases) in order; ev gets one
	// byte per vertex in discovery order, used one flag per start (see FGraphBfs).  The set of seen
	// vertices carries over between calls (the reference's colour map spans all input files).
	void graph_bfs(const Batch& starts, std::vector<uint8_t>& ev, std::vector<uint8_t>& used, uint64_t* edges)
	{
		ev.clear(); used.assign(starts.n, 0);
		*edges = 0;
		if (!starts.n) return;
		need_whole_filter("-g");
		gather_counters();
		dispatch_nw([&](auto nw) { graph_bfs_nw<decltype(nw)::value>(starts, ev, used, edges); });
	}

	// ---- PASS 1 on a device-resident packed batch (ops are inserted in batch order).
	// koff[i + 1] = koff[i] + k-mers of sequence i, computed on the device; returns whether a sequence is shorter than k
	bool device_koff(const uint32_t* len_d, uint64_t n, uint64_t* koff_d)
	{
		be_.memset(koff_d, 0, 8);
		if (!n) return false;
		uint32_t* flag = (uint32_t*)be_.alloc(4);
		be_.memset(flag, 0, 4);
		FKmerCounts f{ len_d, p_.k, koff_d + 1, flag };
		be_.launch(n, f, "kmer_counts");
		be_.inclusive_sum_u64(koff_d + 1, n);
		uint32_t any = 0;
		be_.d2h(&any, flag, 4);
		be_.free(flag);
		return any != 0;
	}
	// op ranges of at most batch_ops_ ops along sequence boundaries, from the device's prefix sums
	std::vector<OpRange> cut_ranges(const uint64_t* koff_d, uint64_t n)
	{
		need_counters();
		ensure_insert();
		std::vector<OpRange> out;
		if (!n) return out;
		uint64_t total = 0;
		be_.d2h(&total, koff_d + n, 8);
		uint32_t cap = (uint32_t)std::min<uint64_t>(n, 2 * (total / std::max<uint64_t>(batch_ops_, 1)) + 16);
		for (;;) {
			OpRange* d = (OpRange*)be_.alloc((uint64_t)cap * sizeof(OpRange));
			uint32_t* cnt = (uint32_t*)be_.alloc(4);
			FCutRanges f{ koff_d, n, batch_ops_, d, cap, cnt };
			be_.launch(1, f, "cut_ranges");
			uint32_t r = 0;
			be_.d2h(&r, cnt, 4);
			if (r <= cap) { out.resize(r); be_.d2h(out.data(), d, (uint64_t)r * sizeof(OpRange)); }
			be_.free(d); be_.free(cnt);
			if (r <= cap) return out;
			cap = r; // (sequences longer than a batch make more ranges than planned for)
		}
	}
	// the same from a host copy of the prefix sums
	std::vector<OpRange> cut_ranges_host(const uint64_t* koff_h, uint64_t n)
	{
		need_counters();
		ensure_insert();
		std::vector<OpRange> out;
		for (uint64_t s = 0; s < n;) {
			uint64_t e = (uint64_t)(std::upper_bound(koff_h + s + 1, koff_h + n + 1, koff_h[s] + batch_ops_) - koff_h) - 1;
			if (e <= s) e = s + 1;
			out.push_back(OpRange{ s, e, koff_h[s], koff_h[e] });
			s = e;
		}
		return out;
	}
	void load_packed(const Batch& b, const uint64_t* koff_h) { load_packed(b, cut_ranges_host(koff_h, b.n)); }
	void load_packed(const Batch& b) { load_packed(b, cut_ranges(b.koff, b.n)); } // (b.koff: device_koff)
	void load_packed(const Batch& b, const std::vector<OpRange>& ranges)
	{
		need_counters();
		cnt_loaded_ = true;
		{
			// Run length of FHashOps: a lane hashes one k-mer from scratch (k rounds) and rolls the rest, so longer
			// runs are less work -- and a run that ends where the read ends starts no second hash.  The reads'
			// k-mer count divided into the fewest equal parts of at most 32: 87 -> 29 (PASS 1 404 -> 390 ms on configs[1]).
			uint32_t len0 = 0;
			if (b.n) be_.d2h(&len0, b.len, 4);
			const uint32_t nk0 = len0 >= p_.k ? len0 - p_.k + 1 : 0;
			const uint32_t parts = (nk0 + 31) / 32;
			hash_run_ = parts ? std::min<uint32_t>(32, std::max<uint32_t>(8, (nk0 + parts - 1) / parts)) : 8;
		}
		ensure_insert();
		memo_valid_ = false; plane_valid_ = false;
		last_rounds_ = 0;
		// Tiled: hashing and binning a batch reads nothing but the reads, so the NEXT batch is hashed
		// and binned on the side stream while this one's tiles are judged and applied and its left-over
		// ops go through the reservation rounds (small kernels and host round trips that leave the
		// machine idle).  Two sets of hashes and bins take turns.
		const bool pipe = tiled_ && bins_alt_ != nullptr && !routed();
		for (size_t i = 0; i < ranges.size(); i++) {
			if (!pipe) { insert_range(b, ranges[i], false); continue; }
			if (i == 0) stage_bins(b, ranges[0]);
			be_.wait_side_scope();
			// (what was staged is now current; the other set takes the next batch)
			std::swap(h0_, h0_alt_); std::swap(bins_, bins_alt_); std::swap(tcur_, tcur_alt_); stage_flag_ ^= 1u;
			if (lead_alt_) { std::swap(lead_, lead_alt_); std::swap(opflag_, opflag_alt_); }
			const bool staged = staged_ok_;
			purity_done_ = staged && staged_purity_;
			const uint32_t cur_flag = stage_flag_ ^ 1u; // the flag word the batch just staged wrote
			// (queued behind this batch's tile kernels -- see insert_range -- so that it runs beside the rounds, not beside them)
			stage_next_ = nullptr;
			if (i + 1 < ranges.size()) stage_next_ = [this, &b, &ranges, i]() { stage_bins(b, ranges[i + 1]); };
			insert_range(b, ranges[i], staged, 2 + cur_flag);
			if (stage_next_) { stage_next_(); stage_next_ = nullptr; }
		}
	}
	// hash_ops, bin_coarse and bin_fine of the op range [s, e) into the alternate set, on the side stream
	void stage_bins(const Batch& b, const OpRange& rg)
	{
		const uint64_t s = rg.s, e = rg.e;
		const uint64_t T = rg.k1 - rg.k0;
		staged_ok_ = false;
		if (T == 0 || T > batch_ops_ || T >= 0xFFFFFFFFull) return; // (insert_range deals with those)
		Batch v = b;
		v.woff = b.woff + s; v.len = b.len + s; v.n = e - s; v.koff = b.koff + s;
		const uint64_t kbase = rg.k0;
		uint32_t* flag = pend_n_ + 2 + stage_flag_;
		uint64_t* h0 = h0_alt_;
		const bool part = dist();
		TileEnv te{ p_, cnt_, part ? own_lo_ : 0, part ? own_lo_ + own_span_ : m_, h0_alt_, bins_alt_, tile_cap_, tcur_alt_, lead_alt_ ? lead_alt_ : lead_,
			opflag_alt_ ? opflag_alt_ : opflag_, tgt_, pendf_, flag, cfg_.benign_sharers ? 1u : 0u, part ? 16u : 2u, part ? 254u : 0x7FFFFFFFu };
		const uint64_t R = part ? (uint64_t)comm_.world : 1, me = part ? (uint64_t)comm_.rank : 0;
		if (part && R > cfg_.dist_hash_all_ranks) {
			// partitioned run: this rank's slice of the hashes on the side stream, the all-gather on the main
			// stream (every collective of the run stays on that one stream, in program order), the bins of
			// the rank's own range on the side stream again
			const uint64_t chunk = ((T + R - 1) / R + 7) & ~7ull;
			std::vector<uint64_t> c(R, chunk * 8), d(R);
			for (uint64_t q = 0; q < R; q++) d[q] = q * chunk * 8;
			const uint64_t a = std::min(T, me * chunk), bnd = std::min(T, a + chunk);
			if (bnd > a) {
				be_.side_scope_begin("hash_staged");
				dispatch_nw([&](auto nw) { FHashOps<(decltype(nw)::value & 7)> f{ p_, v, h0, bnd, kbase, a, hash_run_ }; be_.launch((bnd - a + hash_run_ - 1) / hash_run_, f, "hash_ops"); });
				be_.side_scope_end();
				be_.wait_side_scope();
			}
			c_all_gather_v(h0, c.data(), d.data());
		}
		be_.side_scope_begin("hash_bin_staged");
		be_.memset(tcur_alt_, 0, ntiles_ * 4);
		be_.memset(flag, 0, 4);
		be_.memset(ccur_, 0, ncoarse_ * 4);
		if (!part || R <= cfg_.dist_hash_all_ranks)
			dispatch_nw([&](auto nw) { FHashOps<(decltype(nw)::value & 7)> f{ p_, v, h0, T, kbase, 0, hash_run_ }; be_.launch((T + hash_run_ - 1) / hash_run_, f, "hash_ops"); });
		BinEnv bn{ te, T, coarse_, coarse_cap_, ccur_, cshift_, ncoarse_ };
		FBinCoarse f1{ bn };
		be_.launch_tiles((T + BIN_CHUNK_OPS - 1) / BIN_CHUNK_OPS, f1, "bin_coarse");
		const uint32_t cpb = (coarse_cap_ + BIN_CHUNK_PAIRS - 1) / BIN_CHUNK_PAIRS;
		FBinFine f2{ bn, cpb };
		be_.launch_tiles((uint64_t)ncoarse_ * cpb, f2, "bin_fine");
		// tile_purity reads nothing but the bins either: judged here too, into a second set of per-op outputs.  (Round 2
		// measured this as a loss -- the side stream was the longer branch then, 461 vs 452 ms per configs[1] step; with
		// 12-byte pairs and the one-read fine binning it is the shorter one, and the main stream keeps target, apply, rounds.)
		staged_purity_ = false;
		if (lead_alt_) {
			be_.memset(lead_alt_, 0, T * 4);
			be_.memset(opflag_alt_, 0, (T + 3) & ~3ull);
			FTilePurity fp{ te };
			be_.launch_tiles(ntiles_, fp, "tile_purity");
			staged_purity_ = true;
		}
		be_.side_scope_end();
		staged_ok_ = true;
	}
	bool staged_ok_ = false; uint32_t stage_flag_ = 0;
	bool purity_done_ = false; // the batch about to be inserted had its tiles judged while it was staged
	std::function<void()> stage_next_; // stages the next batch; insert_range calls it once its tile kernels are queued

	// ---- PASS 2 on a device-resident packed batch of reads.  results_host (b.n bytes,
	// may be NULL) receives a ReadResult per read; contigs are delivered in commit order.
	//
	// The reads go through in batches, one at a time: classify -> walk the candidates -> ordered commit; the next batch is
	// classified on the side stream beside this batch's walkers.  (Several batches in flight, a deferral stage in which the
	// walkers yield to lower-numbered ones, and a pre-search of the candidates' own branching k-mers were built, measured
	// and taken out again: notes/README.md.)
	void assemble_packed(const Batch& b, uint8_t* results_host,
	    const std::function<void(const ContigOut&)>& sink)
	{
		// (ABG_HOST_TIMING: where a call's time goes before its first batch -- the marks wait for the device, so the run is a little slower)
		const bool timing = getenv("ABG_HOST_TIMING") != nullptr;
		const auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		const double t_in = tnow();
		const auto mark = [&](const char* what) { if (timing) { be_.sync(); fprintf(stderr, "[host] assemble_packed +%.3f s: %s\n", tnow() - t_in, what); } };
		need_counters();
		gather_counters();
		// (PASS 1's bins and claim tables grow with the filter -- 60 GB at B=40G: PASS 2 gets that memory)
		if (insert_scratch_bytes_ > cfg_.keep_insert_scratch_bytes) free_insert();
		mark("counters final, PASS 1's scratch given back where it is large");
		ensure_walk();
		mark("walkers' pools and tables");
		ensure_plane();
		ensure_both();
		mark("bit plane, two-bit array");
		ensure_archive(b);
		mark("archive");
		build_guide(b);
		mark("guide");
		ensure_memo();
		mark("memo");
		uint8_t* result_d = (uint8_t*)be_.alloc(b.n ? b.n : 1);
		dispatch_nw([&](auto nw) { assemble_nw<decltype(nw)::value>(b, result_d, results_host, sink); });
		mark("batches");
		deliveries_wait(); // (the last batch's contigs are with the caller)
		be_.sync_side();
		mark("last delivery");
		pre_n_ = 0; prefetch_ = nullptr;
		guide_.tab = nullptr; // its hints point into this call's reads
		be_.free(result_d);
	}
	// The memo of successor() answers (SuccMemo): emptied whenever the solid filter may have changed.
	void ensure_memo()
	{
		if (!cfg_.memo) { memo_.k0 = nullptr; return; }
		if (!memo_tab_.hmin) {
			uint32_t log2 = cfg_.memo_log2;
			if (!log2) { log2 = 16; while (log2 < 26 && (1ull << log2) < m_ / 128) log2++; }
			alloc_tab(memo_tab_, log2);
			memo_valid_ = false; plane_valid_ = false;
		}
		if (!memo_valid_) {
			be_.memset(memo_tab_.hmin, 0xFF, (memo_tab_.mask + 1) * 8);
			be_.memset(memo_tab_.hmax, 0xFF, (memo_tab_.mask + 1) * 8); // (a reader that sees an entry's value before its second key word sees "no key")
			be_.memset(memo_tab_.meta, 0, (memo_tab_.mask + 1) * 8);
			memo_valid_ = true;
			memo_gen_++;
		}
		memo_ = SuccMemo{ memo_tab_.hmin, memo_tab_.hmax, memo_tab_.meta, memo_tab_.mask };
	}
	// The solid filter as PASS 2 probes it (probe_c, abg_core.h): the bit plane "counter >= kc", built once
	// the counters are final and kept until they change (the same events that empty the memo).
	void ensure_plane()
	{
		p2_ = p_; cnt2_ = cnt_;
		if (!cfg_.solid_plane || casc_.bits || (m_ & 63)) return;
		if (!plane_) plane_ = (uint8_t*)be_.alloc((sliced_ ? (uint64_t)comm_.world * own_chunk_ : m_) / 8 + 64);
		if (!plane_valid_ && sliced_) {
			// every rank judges its own counters and the ranks gather the bits: an eighth of what gather_counters moves
			FSolidPlane f{ (const uint64_t*)(cnt_ + own_lo_), p_.kc, (uint64_t*)(plane_ + own_lo_ / 8) };
			be_.launch(own_span_ / 64, f, "solid_plane");
			std::vector<uint64_t> c(comm_.world), d(comm_.world);
			for (int q = 0; q < comm_.world; q++) { d[q] = (uint64_t)q * (own_chunk_ / 8); c[q] = own_chunk_ / 8; }
			c_all_gather_v(plane_, c.data(), d.data());
			plane_valid_ = true; cnt_partial_ = false;
		} else if (!plane_valid_) {
			FSolidPlane f{ (const uint64_t*)cnt_, p_.kc, (uint64_t*)plane_ };
			be_.launch(m_ / 64, f, "solid_plane");
			plane_valid_ = true;
		}
		p2_.solid_bits = 1; cnt2_ = plane_;
	}
	// the classification's two bits per position (FBothBuild): single-GPU runs that probe the plane, rebuilt for every assemble call
	// (nothing but this call's commits writes the visited filter while it runs)
	void ensure_both()
	{
		// (filters up to cls_both_max_mb of it: on configs[2]'s 40 G counters the 10 GB array gains 1 % a pass and costs the first pass 6 s)
		const bool want = cfg_.cls_both && p2_.solid_bits && !dist() && !sliced_ && !(m_ & 63) && m_ / 4 <= ((uint64_t)cfg_.cls_both_max_mb << 20);
		if (!want) { if (both_) { be_.free(both_); both_ = nullptr; } return; }
		if (!both_) both_ = (uint8_t*)be_.try_alloc(m_ / 4 + 64);
		if (!both_) return;
		FBothBuild f{ (const uint32_t*)plane_, (const uint32_t*)vis_, (uint64_t*)both_ };
		be_.launch(m_ / 32, f, "both_build");
	}
	uint8_t* both_ = nullptr;
	// The archive of committed contigs the classification reads (ContigArchive): single-GPU runs with the parallel commit (FPcApply
	// fills it).  Sized by the call's reads -- a byte for every fourth read k-mer, a table slot for every eighth: a genome sequenced
	// four-fold or deeper fits, and what does not fit is simply not archived -- and kept from call to call as long as the visited
	// filter is (reset, visited_dev: emptied).
	void free_archive()
	{
		if (arc_.seq) { be_.free(arc_.seq); be_.free(arc_.used); be_.free(arc_.tab); }
		arc_ = ContigArchive(); arc_valid_ = false;
	}
	void ensure_archive(const Batch& b)
	{
		const bool want = cfg_.cls_archive && !dist() && !sliced_ && !casc_.bits && b.n && use_par_commit();
		if (!want) { free_archive(); return; }
		// (the k-mers PASS 1 loaded, wherever they came from -- the caller may assemble its reads in several calls; a filter that
		// was imported: a sixteenth of its counters)
		const uint64_t nk = loaded_ops_ ? loaded_ops_ : m_ / 4;
		uint64_t cap = std::max<uint64_t>(nk / 4, 1ull << 16) + ARC_HEAD, slots = 1ull << 12;
		while (slots < cap / 2) slots <<= 1;
		const uint64_t budget = (uint64_t)cfg_.cls_archive_max_mb << 20;
		while (slots > (1ull << 12) && slots * 8 + cap > budget) { slots >>= 1; cap = std::min(cap, slots * 2); }
		if (cap >= (1ULL << ARC_POS_BITS)) cap = (1ULL << ARC_POS_BITS) - 1;
		if (arc_.seq && (arc_.cap < cap || arc_.mask + 1 < slots)) free_archive(); // (a larger read set than the one it was made for)
		if (!arc_.seq) {
			arc_.seq = (uint8_t*)be_.try_alloc(cap + ARC_PAD);
			arc_.tab = (uint64_t*)be_.try_alloc(slots * 8);
			arc_.used = (uint64_t*)be_.try_alloc(8);
			if (!arc_.seq || !arc_.tab || !arc_.used) {
				if (arc_.seq) be_.free(arc_.seq); if (arc_.tab) be_.free(arc_.tab); if (arc_.used) be_.free(arc_.used);
				arc_ = ContigArchive();
				return;
			}
			arc_.cap = cap; arc_.mask = slots - 1;
			arc_valid_ = false;
		}
		arc_.nreads = wstats_ ? wstats_ + WSTAT_CLS_COVERED : nullptr;
		if (!arc_valid_) {
			be_.memset(arc_.seq, ARC_SEP, arc_.cap + ARC_PAD);
			be_.memset(arc_.tab, 0, (arc_.mask + 1) * 8);
			const uint64_t head = ARC_HEAD;
			be_.h2d(arc_.used, &head, 8);
			arc_valid_ = true;
		}
	}
	ContigArchive arc_{}; bool arc_valid_ = false;
	uint64_t loaded_ops_ = 0; // k-mers inserted since the filter was last empty (ensure_archive)
	uint64_t ovf_seen_[2] = { 0, 0 }; // partitioned run: the walkers' pool / record overflow counters as last read
	uint8_t* plane_ = nullptr; bool plane_valid_ = false;
	Params p2_; const uint8_t* cnt2_ = nullptr; // what the probing kernels of PASS 2 get: p_ / cnt_, or the plane

	// The guide of the bulk steps for the reads of one assemble_packed call (see FGuideBuild):
	// sized to the sampled reads' k-mers, of which the solid ones -- a genome's worth -- stay.
	void build_guide(const Batch& b)
	{
		guide_.tab = nullptr; guide_.seen = nullptr; guide_slots_ = 0;
		if (!cfg_.guide_stride || p_.nh > 8 || !b.n) return;
		const uint64_t sampled = (b.n + cfg_.guide_stride - 1) / cfg_.guide_stride;
		uint64_t nwords = 0;
		be_.d2h(&nwords, b.woff + b.n, 8);
		const uint64_t bases = nwords * 16, minus = b.n * (uint64_t)(p_.k - 1);
		// (slots for a quarter of the reads' k-mers, whatever the stride: what stays in the table is a genome's worth of
		// solid k-mers, and in a direct-mapped table every second one of them lost to a collision is a bulk step not taken)
		const uint64_t kmers = (bases > minus ? bases - minus : b.n) / std::min<uint32_t>(cfg_.guide_stride, 4u);
		uint32_t log2 = 16;
		while ((1ull << log2) < kmers && log2 < cfg_.guide_log2_max) log2++;
		if (!guide_tab_ || log2 != guide_log2_) {
			if (guide_tab_) be_.free(guide_tab_);
			guide_tab_ = (uint64_t*)be_.try_alloc(8ull << log2);
			guide_log2_ = log2;
			if (!guide_tab_) { if (cfg_.verbose) fprintf(stderr, "abyss_amd: no memory for the walkers' guide table, walking step by step\n"); return; }
		}
		be_.memset(guide_tab_, 0, 8ull << log2);
		guide_.mask = (1ull << log2) - 1; guide_.words = b.words; guide_.nwords = nwords;
		dispatch_nw([&](auto nw) {
			FGuideBuild<decltype(nw)::value> f{ p2_, b, cnt2_, guide_tab_, guide_.mask, cfg_.guide_stride, sampled };
			be_.launch_wave((sampled + GUIDE_READS_PER_ITEM - 1) / GUIDE_READS_PER_ITEM, f, "guide_build");
		});
		guide_.tab = guide_tab_;
		guide_slots_ = guide_.mask + 1;
		// what the bulk steps find out about a read's k-mers, for the walkers that come along them later (Guide::seen): a byte per
		// base of the call's reads -- 1.5 GB for configs[1] -- when the device has that to spare
		if (cfg_.guide_seen) {
			const uint64_t need = nwords * 16, total = be_.device_mem_bytes();
			if (guide_seen_ && guide_seen_bytes_ < need) { be_.free(guide_seen_); guide_seen_ = nullptr; }
			if (!guide_seen_ && (!total || need <= total / 24)) { guide_seen_ = (uint8_t*)be_.try_alloc(need); guide_seen_bytes_ = need; }
			if (guide_seen_) { be_.memset(guide_seen_, 0, need); guide_.seen = guide_seen_; }
		}
	}
	// Batches grow geometrically up to p2_max_batch (larger ones walk too many reads of the same
	// unitigs side by side).  A batch with few candidates, though, is bound by its slowest walker,
	// not by throughput: while that lasts (the start, and the end of a read set, where nearly
	// every read is already visited) the size keeps doubling.
	uint64_t next_batch_size() const
	{
		// (a larger genome at the same read count leaves more of each batch unvisited: when a batch
		// had more candidates than the walkers' tables should have to hold, the next one is halved)
		if (last_candidates_ > cfg_.p2_crowded) return std::max<uint64_t>(p2_batch_ / 2, 1024);
		if (last_candidates_ < cfg_.p2_starved) return std::min<uint64_t>(p2_batch_ * cfg_.p2_starved_growth, 8 * cfg_.p2_max_batch);
		return std::min<uint64_t>(p2_batch_ * cfg_.p2_growth, cfg_.p2_max_batch);
	}
	void clear_wstats()
	{
		be_.memset(wstats_, 0, WSTAT_N * 8);
	}
	struct Stats { uint64_t rounds = 0, walked = 0, rewalked = 0, candidates = 0, breaks = 0, insert_rounds = 0, commit_rounds = 0, generated = 0;
	               uint64_t bulk_calls = 0, bulk_steps = 0, lin_steps = 0, guide_slots = 0, chain_steps = 0, batch_cuts = 0, overflows = 0, memo_hits = 0, memo_adds = 0;
	               uint64_t tiled_ops = 0, tiled_pending = 0, tile_overflows = 0, cls_covered_reads = 0, archive_bases = 0, cls_decided_reads = 0; };
	Stats stats()
	{
		Stats s = stats_;
		if (wstats_) {
			uint64_t v[WSTAT_N];
			be_.d2h(v, wstats_, sizeof v);
			s.bulk_calls = v[WSTAT_BULK_CALLS]; s.bulk_steps = v[WSTAT_BULK_STEPS]; s.lin_steps = v[WSTAT_LIN_STEPS]; s.chain_steps = v[WSTAT_CHAIN_STEPS];
			s.memo_hits = v[WSTAT_MEMO_HITS]; s.memo_adds = v[WSTAT_MEMO_ADDS];
			s.cls_covered_reads = v[WSTAT_CLS_COVERED]; s.cls_decided_reads = v[WSTAT_CLS_DECIDED];
		}
		if (arc_.seq && arc_valid_) { uint64_t u = 0; be_.d2h(&u, arc_.used, 8); s.archive_bases = std::min(u, arc_.cap) - ARC_HEAD; }
		s.guide_slots = guide_slots_;
		return s;
	}

  private:
	BE& be_;
	Config cfg_;
	Params p_;
	uint64_t m_ = 0, vis_bytes_ = 0;
	uint8_t* cnt_ = nullptr;
	uint8_t* vis_ = nullptr;
	CommitState* cstate_ = nullptr;
	uint64_t* scal_ = nullptr;
	Cascade casc_{ nullptr, 0, 0 };
	MaskTab* mask_d_ = nullptr;
	std::chrono::steady_clock::time_point dbg_t0_;
	uint32_t* T_ = nullptr; // parallel commit: time stamp per filter bit
	uint32_t t_tag_ = 0;    // next pass takes tag t_tag_ - 1; 0: clear T first
	// the parallel commit needs 4 bytes of time stamp per filter bit; without that memory (or when
	// switched off) the ordered single-workgroup kernel runs
	uint64_t* Tk_ = nullptr; uint32_t* Tv_ = nullptr; uint32_t Tlog2_ = 0; // ... or per touched bit (see pc_T_slot)
	uint64_t T_entries_ = 0; // upper bound of the keys in the table
	bool t_hashed() const { return t_force_hashed_ || m_ * 4ull > cfg_.par_commit_max_bytes; }
	bool t_force_hashed_ = false; // no room for a stamp per filter bit on a sliced filter, where the ordered kernel cannot run: a stamp per touched bit
	bool use_par_commit()
	{
		if (!cfg_.par_commit || t_failed_) return false;
		if (t_hashed()) return true; // (the table is sized by commit_par)
		if (!T_) {
			T_ = (uint32_t*)be_.try_alloc(m_ * 4ull);
			t_tag_ = 0;
			if (!T_ && sliced_) {
				// (the ordered commit kernel and FContigPrep's coverage sums read counters at any position: on a sliced filter those are
				// not there.  The stamps go to the hashed table instead -- sized to what a commit touches -- or the run fails cleanly.)
				t_force_hashed_ = true;
				if (cfg_.verbose) fprintf(stderr, "abyss_amd: no memory for a time stamp per filter bit, hashed stamps instead\n");
				return true;
			}
			if (!T_) { t_failed_ = true; if (cfg_.verbose) fprintf(stderr, "abyss_amd: no memory for the parallel commit's time stamps, using the ordered kernel\n"); return false; }
		}
		return true;
	}
	bool t_failed_ = false;
	// ---- partitioned run
	Comm comm_;
	bool force_dist_ = false, comm_scaled_ = false;
	bool sliced_ = false, cnt_deferred_ = false; uint64_t win_total_ = 0, win_span_ = 0, win_lo_ = 0; // the sliced filter (attach_comm): the window as it was allocated
	bool cnt_loaded_ = false; // counters were inserted or imported since the filter was last empty
	uint64_t own_lo_ = 0, own_span_ = 0, own_chunk_ = 0;
	uint8_t* tred_ = nullptr; // partitioned tiles: the two bytes per op of FDistPack
	bool cnt_partial_ = false; // PASS 1 ran partitioned since the counters were last gathered
	uint32_t* sh_words_ = nullptr; uint64_t* sh_woff_ = nullptr; uint32_t* sh_len_ = nullptr; uint64_t* sh_koff_ = nullptr;
	uint8_t* dres_ = nullptr; uint8_t* dlost_ = nullptr; // PASS 1: one byte per pending op
	// routed form (insert_tiles_routed): records by destination, records received, where each own pair went, replies and targets both ways
	TilePair* rsend_ = nullptr; TilePair* rrecv_ = nullptr; uint32_t rcap_ = 0; uint64_t rrecv_cap_ = 0; uint32_t* rslot_ = nullptr; uint32_t* rcur_ = nullptr;
	uint8_t* rrep_out_ = nullptr; uint8_t* rrep_in_ = nullptr; uint8_t* rtgt_out_ = nullptr; uint8_t* rtgt_in_ = nullptr; uint8_t* rpendf_ = nullptr;
	uint64_t rown_ = 0; // ops a rank hashes per batch, at most
	bool routed() const
	{
		return dist() && cfg_.dist_route_min_ranks && (uint32_t)comm_.world >= cfg_.dist_route_min_ranks && (comm_.all_to_all_v || comm_.world == 1) && p_.nh <= 16;
	}
	uint32_t g_rec_ = 0; uint64_t g_pool_ = 0;           // PASS 2: records / pool bytes every rank holds
  public:
  private:
	void free_shared()
	{
		if (!sh_words_) return;
		be_.free(sh_words_); be_.free(sh_woff_); be_.free(sh_len_); be_.free(sh_koff_);
		sh_words_ = nullptr;
	}
	void c_fail(const char* what) { fail_now(FAIL_INTERNAL, std::string("collective ") + what + " failed on rank " + std::to_string(comm_.rank)); }
	// in place: rank q's part lives at buf + displs[q] (bytes); on return every rank holds all parts
	void c_all_gather_v(void* buf, const uint64_t* counts, const uint64_t* displs)
	{
		if (!comm_.stream_ordered) be_.sync();
		be_.begin("comm_all_gather");
		if (comm_.all_gather_v(comm_.user, buf, counts, displs, be_.stream_handle())) c_fail("all_gather_v");
		be_.end("comm_all_gather");
	}
	void c_all_reduce(void* buf, uint64_t n, int dtype, int op)
	{
		if (!n) return;
		if (!comm_.stream_ordered) be_.sync();
		be_.begin("comm_all_reduce");
		if (comm_.all_reduce(comm_.user, buf, n, dtype, op, be_.stream_handle())) c_fail("all_reduce");
		be_.end("comm_all_reduce");
	}
	// send_counts[q] bytes at send + send_displs[q] to rank q; recv_counts[q] bytes from rank q to recv + recv_displs[q]
	void c_all_to_all_v(const void* send, const uint64_t* sc, const uint64_t* sd, void* recv, const uint64_t* rc, const uint64_t* rd)
	{
		if (comm_.world == 1) { // (ABG_FORCE_DIST on one rank: the exchange is a copy)
			if (sc[0]) be_.d2d((char*)recv + rd[0], (const char*)send + sd[0], sc[0]);
			return;
		}
		if (!comm_.stream_ordered) be_.sync();
		be_.begin("comm_all_to_all");
		if (comm_.all_to_all_v(comm_.user, send, sc, sd, recv, rc, rd, be_.stream_handle())) c_fail("all_to_all_v");
		be_.end("comm_all_to_all");
	}
	void host_all_reduce_sum(uint64_t* v, uint32_t n)
	{
		uint64_t* d = (uint64_t*)be_.alloc(n * 8ull);
		be_.h2d(d, v, n * 8ull);
		c_all_reduce(d, n, DT_U64, OP_SUM);
		be_.d2h(v, d, n * 8ull);
		be_.free(d);
	}
	// the shards of the counting filter, gathered onto every rank (PASS 2 reads it at random)
	void gather_counters()
	{
		if (!dist() || !cnt_partial_ || sliced_) return;
		std::vector<uint64_t> c(comm_.world), d(comm_.world);
		// equal chunks (one ring all-gather); the last one runs into the slack behind the m_ counters
		for (int q = 0; q < comm_.world; q++) { d[q] = (uint64_t)q * own_chunk_; c[q] = own_chunk_; }
		c_all_gather_v(cnt_, c.data(), d.data());
		cnt_partial_ = false;
	}
	Counters counters_;
	Stats stats_;
	uint64_t last_rounds_ = 0;
	uint64_t p2_batch_ = 0;
	// PASS 1 resources
	uint64_t* h0_ = nullptr; uint64_t* claim_[2] = { nullptr, nullptr };
	uint32_t* pend_[2] = { nullptr, nullptr }; uint32_t* pend_n_ = nullptr; uint32_t epoch_ = 1;
	uint64_t batch_ops_ = 0;   // ops per ordered-insert batch (ensure_insert)
	uint32_t hash_run_ = 8;    // FHashOps: consecutive ops per lane
	uint64_t insert_scratch_bytes_ = 0; // what ensure_insert holds
	uint32_t claim_log2_ = 0;  // slots per claim table
	bool tiled_ = false; uint64_t ntiles_ = 0, napply_ = 0; uint32_t tile_cap_ = 0; // PASS 1 through tiles (TileEnv)
	TilePair* coarse_ = nullptr; uint32_t* ccur_ = nullptr; uint32_t coarse_cap_ = 0, cshift_ = 0, ncoarse_ = 0;
	TilePair* bins_ = nullptr; uint32_t* tcur_ = nullptr; uint32_t* lead_ = nullptr; uint8_t* opflag_ = nullptr; uint8_t* tgt_ = nullptr; uint8_t* pendf_ = nullptr;
	// a second set of hashes, bins and bin cursors: the batch being staged on the side stream (stage_bins)
	uint64_t* h0_alt_ = nullptr; TilePair* bins_alt_ = nullptr; uint32_t* tcur_alt_ = nullptr;
	uint32_t* colist_ = nullptr; uint8_t* cocnt_ = nullptr;
	uint64_t* pgrp_ = nullptr; // the rounds' ops per group of PEND_GROUP ops, then their running sum (FPendCount)
	uint8_t* wmask_ = nullptr; uint32_t* bad_ = nullptr; uint32_t* cochg_ = nullptr; // what settles k-mers writing shared counters (op_verdict)
	uint32_t* lead_alt_ = nullptr; uint8_t* opflag_alt_ = nullptr; // ... and, when its tiles are judged there as well, of what tile_purity leaves per op
	bool staged_purity_ = false;
	// PASS 2 resources
	Guide guide_{ nullptr, 0, nullptr, 0, nullptr }; uint64_t* guide_tab_ = nullptr; uint32_t guide_log2_ = 0;
	uint8_t* guide_seen_ = nullptr; uint64_t guide_seen_bytes_ = 0;
	BulkScratch* bulk_pool_ = nullptr; uint64_t* wstats_ = nullptr; uint64_t guide_slots_ = 0;
	SuccMemo memo_{ nullptr, nullptr, nullptr, 0 }; WalkTab memo_tab_{ nullptr, nullptr, nullptr, 0 }; bool memo_valid_ = false; uint64_t memo_gen_ = 0; // (valid: filled against the solid filter as it is now)
	bool walk_ready_ = false;
	WalkTab wtab_{}, cend_{ nullptr, nullptr, nullptr, 0 };
	uint32_t wtab_log2_ = 0;
	uint64_t wtab_per_walker_ = 1536; // planning figure: vertices one walker enters (config 2 averages ~1100)
	void* tb_pool_ = nullptr; VKey* tbk_pool_ = nullptr; VKey* la_pool_ = nullptr; uint8_t* lbuf_ = nullptr; uint8_t* rbuf_ = nullptr;
	uint8_t* pool_ = nullptr; uint64_t pool_cap_ = 0; uint64_t* pool_used_ = nullptr;
	ContigRec* recs_ = nullptr; uint32_t rec_cap_ = 0; uint32_t* rec_used_ = nullptr;
	uint32_t* order_ = nullptr; uint32_t* order_n_ = nullptr;
	uint32_t walk_tb_cap_ = 0, walk_buf_cap_ = 0, wslots_ = 0, cslots_ = 0;
	uint64_t* kh_ = nullptr; uint64_t* rkh_ = nullptr; uint64_t* dbg_ = nullptr;
	uint64_t cend_count_ = 0;
	uint8_t* read_flag_ = nullptr;
	uint64_t last_candidates_ = 0;
	VKey* la_pool_c_ = nullptr; // lookAhead scratch of the classification (the walkers have their own per context)
	VKey* la_pool_c2_ = nullptr; // ... and of the classification running ahead on the side stream
	uint64_t pre_first_ = 0, pre_n_ = 0;      // range classified ahead on the side stream
	std::function<void()> prefetch_;          // queues that classification (called right before a launch of walkers)

	// ---- -g state: the vertices seen by any search so far
	WalkTab gtab_{ nullptr, nullptr, nullptr, 0 };
	uint64_t gtab_used_ = 0;
	// the build of the device code for this k (and for a spaced seed: see NW_MASKED, abg_core.h)
	template <class F>
	void dispatch_nw(F&& f)
	{
		switch (p_.mask ? NW_MASKED + p_.nw : p_.nw) {
		case 1: f(std::integral_constant<int, 1>()); break;
		case 2: f(std::integral_constant<int, 2>()); break;
		case 3: f(std::integral_constant<int, 3>()); break;
		case 4: f(std::integral_constant<int, 4>()); break;
		case 5: case 6: f(std::integral_constant<int, 6>()); break;
		case NW_MASKED + 1: f(std::integral_constant<int, NW_MASKED + 1>()); break;
		case NW_MASKED + 2: f(std::integral_constant<int, NW_MASKED + 2>()); break;
		case NW_MASKED + 3: f(std::integral_constant<int, NW_MASKED + 3>()); break;
		case NW_MASKED + 4: f(std::integral_constant<int, NW_MASKED + 4>()); break;
		default: f(std::integral_constant<int, NW_MASKED + 6>()); break;
		}
	}
	template <int NW>
	void graph_bfs_nw(const Batch& starts, std::vector<uint8_t>& ev, std::vector<uint8_t>& used, uint64_t* edges)
	{
		if (!gtab_.hmin) { alloc_tab(gtab_, 16); gtab_used_ = 0; }
		uint64_t node_cap = 1ull << 16;
		Vtx<NW>* nodes = (Vtx<NW>*)be_.alloc(node_cap * sizeof(Vtx<NW>));
		uint8_t* ev_d = (uint8_t*)be_.alloc(node_cap);
		uint8_t* used_d = (uint8_t*)be_.alloc(starts.n);
		GraphState* st_d = (GraphState*)be_.alloc(sizeof(GraphState));
		GraphState st;
		memset(&st, 0, sizeof st);
		st.tab_used = gtab_used_;
		be_.h2d(st_d, &st, sizeof st);
		for (;;) {
			FGraphBfs<NW> f{ p_, cnt_, starts, starts.n, gtab_, (gtab_.mask + 1) / 2, nodes, node_cap, ev_d, used_d, st_d };
			be_.launch(1, f, "graph_bfs");
			be_.d2h(&st, st_d, sizeof st);
			if (st.stop == 1) {
				nodes = (Vtx<NW>*)regrow(nodes, node_cap * sizeof(Vtx<NW>), 2 * node_cap * sizeof(Vtx<NW>));
				ev_d = (uint8_t*)regrow(ev_d, node_cap, 2 * node_cap);
				node_cap *= 2;
			} else if (st.stop == 2) {
				uint32_t log2 = 1;
				while ((1ull << log2) < gtab_.mask + 1) log2++;
				WalkTab bigger{};
				alloc_tab(bigger, log2 + 1);
				uint32_t* failed = (uint32_t*)be_.alloc(8);
				be_.memset(failed, 0, 4);
				FRehash fr{ gtab_, bigger, failed };
				be_.launch(gtab_.mask + 1, fr, "rehash");
				uint32_t bad = 0;
				be_.d2h(&bad, failed, 4);
				be_.free(failed);
				if (bad) fail_now(FAIL_INTERNAL, "graph table rehash failed");
				free_tab(gtab_);
				gtab_ = bigger;
			} else {
				break;
			}
		}
		gtab_used_ = st.tab_used;
		ev.resize(st.count);
		be_.d2h(ev.data(), ev_d, st.count);
		be_.d2h(used.data(), used_d, starts.n);
		*edges = st.edges;
		be_.free(nodes); be_.free(ev_d); be_.free(used_d); be_.free(st_d);
	}

	void ensure_insert()
	{
		if (h0_) return;
		// ops per ordered-insert batch.  Tiled: what keeps a tile's bin within the LDS sort (about
		// 3000 pairs per tile and batch), and for a large filter enough ops that a tile's 64 KB are
		// streamed for a few thousand pairs, not for a handful
		batch_ops_ = cfg_.insert_batch_kmers;
		tiled_ = false;
		// (partitioned run: a rank tiles its own range, and sees 1/R of a batch's pairs)
		const uint64_t R = dist() ? (uint64_t)comm_.world : 1;
		ntiles_ = ((dist() ? own_chunk_ : m_) + BIN_COUNTERS - 1) >> BIN_BITS; // (bins: what the pairs are sorted into and tile_purity judges)
		napply_ = (ntiles_ + TILE_SPLIT - 1) / TILE_SPLIT;                     // (tiles: what tile_apply holds in LDS)
		if (cfg_.tiled_insert && !casc_.bits && p_.nh <= 16) {
			uint64_t T = std::max<uint64_t>(batch_ops_, std::min<uint64_t>(1ull << 28, m_ / 114 * (TILE_PAIRS_MEAN / 2048u) * TILE_SPLIT));
			T = std::min<uint64_t>(T, (uint64_t)TILE_PAIRS_MEAN * ntiles_ * R / p_.nh);
			// (the routed path still settles by round 2's rule -- a k-mer with a shared counter takes the rounds -- and the share of such
			// k-mers grows with the batch: it keeps the batch of the rounds before the bins were halved, half of everybody else's)
			if (routed() && T / TILE_SPLIT >= (1u << 20)) T /= TILE_SPLIT; // (small filters: the batch is what their few tiles hold anyway)
			T = std::min<uint64_t>(T, 1ull << TP_T_BITS); // (a pair holds its op in 28 bits)
			if (T >= 1024) {
				tiled_ = true;
				batch_ops_ = T;
				const uint64_t mean = (T * p_.nh + ntiles_ * R - 1) / (ntiles_ * R);
				uint64_t sq = 1;
				while (sq * sq < mean) sq++;
				tile_cap_ = (uint32_t)std::min<uint64_t>(TILE_SORT_MAX, mean + 8 * sq + 64);
				if (const char* e = getenv("ABG_TILE_CAP")) tile_cap_ = (uint32_t)std::max(1, std::min<int>(TILE_SORT_MAX, atoi(e))); // (tests: forces bin overflows)
			}
		}
		uint64_t nb = batch_ops_;
		if (tiled_) {
			// coarse bins of the first binning pass: runs of 2^cshift tiles, at most BIN_MAX_COARSE of them
			cshift_ = 0;
			while ((ntiles_ >> cshift_) > 512 && cshift_ < 12) cshift_++;
			ncoarse_ = (uint32_t)((ntiles_ + (1ull << cshift_) - 1) >> cshift_);
			{
				const uint64_t mean = (nb * p_.nh + ncoarse_ * R - 1) / (ncoarse_ * R);
				uint64_t sq = 1;
				while (sq * sq < mean) sq++;
				coarse_cap_ = (uint32_t)(mean + 8 * sq + 256);
			}
			coarse_ = (TilePair*)be_.alloc((uint64_t)ncoarse_ * coarse_cap_ * sizeof(TilePair));
			ccur_ = (uint32_t*)be_.alloc(ncoarse_ * 4 + 64);
			bins_ = (TilePair*)be_.alloc(ntiles_ * tile_cap_ * sizeof(TilePair));
			tcur_ = (uint32_t*)be_.alloc(ntiles_ * 4 + 64);
			if (cfg_.overlap_bins) {
				bins_alt_ = (TilePair*)be_.try_alloc(ntiles_ * tile_cap_ * sizeof(TilePair));
				if (bins_alt_) {
					tcur_alt_ = (uint32_t*)be_.alloc(ntiles_ * 4 + 64);
					h0_alt_ = (uint64_t*)be_.alloc((nb + 8 * R + 8) * 8);
					if (cfg_.overlap_purity && !routed()) {
						lead_alt_ = (uint32_t*)be_.alloc(nb * 4);
						opflag_alt_ = (uint8_t*)be_.alloc(nb + 8);
					}
				}
			}
			lead_ = (uint32_t*)be_.alloc(nb * 4);
			opflag_ = (uint8_t*)be_.alloc(nb + 8); // (flagged with word-wide ORs)
			tgt_ = (uint8_t*)be_.alloc(nb);
			pendf_ = (uint8_t*)be_.alloc(nb + 8); // (read a word at a time: FPendWrite)
			pgrp_ = (uint64_t*)be_.alloc((nb / PEND_GROUP + 2) * 8);
			if (!routed() && cfg_.cosettle && cfg_.benign_sharers && p_.nh <= 8) {
				wmask_ = (uint8_t*)be_.alloc(nb);
				bad_ = (uint32_t*)be_.alloc((1ull << cfg_.cosettle_log2) / 8);
				cochg_ = (uint32_t*)be_.alloc((CO_MAX_PASSES + 1) * 4);
				colist_ = (uint32_t*)be_.alloc((nb + 64) * 4);
				cocnt_ = (uint8_t*)be_.alloc((nb >> BE::ITEM_GROUP_LOG2) + 64);
			}
			if (dist()) tred_ = (uint8_t*)be_.alloc((bad_ ? 3 + p_.nh : 2) * nb + 256); // (FDistPack: 2 bytes an op; FDistPack2: 2 + nh and 1)
			if (routed()) {
				// a rank hashes a slice of at most rown_ ops and sends nh records each; a destination gets 1 / R of them on average
				rown_ = (((nb + R - 1) / R + 7) & ~7ull);
				const uint64_t mean = rown_ * p_.nh / R;
				uint64_t sq = 1;
				while (sq * sq < mean) sq++;
				rcap_ = (uint32_t)(mean + 8 * sq + 1024);
				if (const char* e = getenv("ABG_ROUTE_CAP")) rcap_ = (uint32_t)std::max(1, atoi(e)); // (tests: a destination's room runs out)
				rsend_ = (TilePair*)be_.alloc((uint64_t)R * rcap_ * sizeof(TilePair));
				rrecv_cap_ = (uint64_t)R * rcap_;
				rrecv_ = (TilePair*)be_.alloc(rrecv_cap_ * sizeof(TilePair));
				rslot_ = (uint32_t*)be_.alloc(rown_ * p_.nh * 4);
				rcur_ = (uint32_t*)be_.alloc((MAX_RANKS + 2) * 4);
				rrep_out_ = (uint8_t*)be_.alloc(rrecv_cap_ * 2); rrep_in_ = (uint8_t*)be_.alloc((uint64_t)R * rcap_ * 2);
				rtgt_out_ = (uint8_t*)be_.alloc((uint64_t)R * rcap_); rtgt_in_ = (uint8_t*)be_.alloc(rrecv_cap_);
				rpendf_ = (uint8_t*)be_.alloc(rown_ + 8);
			}
		}
		h0_ = (uint64_t*)be_.alloc((nb + 8 * R + 8) * 8);
		// The claim tables of the reservation rounds: the false-conflict rate falls with the load, so
		// as many slots as the configuration allows -- but no more than 8 per (op, counter) pair the
		// rounds can see in one batch (a quarter of the batch's pairs when the tiles settle the rest),
		// and no more than the device has room for.
		{
			const uint64_t pairs = nb * p_.nh / (tiled_ ? 4 : 1);
			uint32_t log2 = 16;
			while (log2 < cfg_.claim_log2 && (1ull << log2) < 8 * pairs) log2++;
			for (;; log2--) {
				claim_[0] = (uint64_t*)be_.try_alloc(8ull << log2);
				claim_[1] = claim_[0] ? (uint64_t*)be_.try_alloc(8ull << log2) : nullptr;
				if (claim_[1] || log2 <= 16) break;
				if (claim_[0]) be_.free(claim_[0]);
			}
			if (!claim_[1]) fail_now(FAIL_NOMEM, "no device memory for the insert claim tables");
			claim_log2_ = log2;
		}
		for (int i = 0; i < 2; i++) {
			be_.memset(claim_[i], 0xFF, 8ull << claim_log2_);
			pend_[i] = (uint32_t*)be_.alloc(nb * 4);
		}
		pend_n_ = (uint32_t*)be_.alloc(16); // [0] pending ops, [1] rounds / bin overflow, [2], [3] bin overflow of a staged batch (by buffer)
		be_.memset(pend_n_, 0, 16);
		if (dist()) dres_ = (uint8_t*)be_.alloc(std::max<uint64_t>(nb, (uint64_t)cfg_.drain_threshold * 32));
		dlost_ = (uint8_t*)be_.alloc(nb);
		insert_scratch_bytes_ = nb * (8 + 4 + 4 + 1) + (16ull << claim_log2_);
		if (tiled_) insert_scratch_bytes_ += ((uint64_t)ncoarse_ * coarse_cap_ + ntiles_ * tile_cap_ * (bins_alt_ ? 2 : 1)) * sizeof(TilePair) + nb * (bins_alt_ ? 15 : 7);
	}
	void free_insert()
	{
		if (!h0_) return;
		if (tiled_) { be_.free(coarse_); be_.free(ccur_); be_.free(bins_); be_.free(tcur_); be_.free(lead_); be_.free(opflag_); be_.free(tgt_); be_.free(pendf_); be_.free(pgrp_); pgrp_ = nullptr; tiled_ = false; }
		if (bad_) { be_.free(wmask_); be_.free(bad_); be_.free(cochg_); be_.free(colist_); be_.free(cocnt_); wmask_ = nullptr; bad_ = nullptr; cochg_ = nullptr; colist_ = nullptr; cocnt_ = nullptr; }
		if (tred_) { be_.free(tred_); tred_ = nullptr; }
		if (rsend_) {
			be_.free(rsend_); be_.free(rrecv_); be_.free(rslot_); be_.free(rcur_); be_.free(rrep_out_); be_.free(rrep_in_); be_.free(rtgt_out_); be_.free(rtgt_in_); be_.free(rpendf_);
			rsend_ = nullptr;
		}
		if (bins_alt_) { be_.free(bins_alt_); be_.free(tcur_alt_); be_.free(h0_alt_); bins_alt_ = nullptr; tcur_alt_ = nullptr; h0_alt_ = nullptr; }
		if (lead_alt_) { be_.free(lead_alt_); be_.free(opflag_alt_); lead_alt_ = nullptr; opflag_alt_ = nullptr; }
		if (dres_) { be_.free(dres_); dres_ = nullptr; }
		be_.free(dlost_);
		be_.free(h0_);
		for (int i = 0; i < 2; i++) { be_.free(claim_[i]); be_.free(pend_[i]); }
		be_.free(pend_n_);
		h0_ = nullptr;
		insert_scratch_bytes_ = 0;
	}
	// staged: the range's hashes and bins are in place (stage_bins), the bin-overflow flag in pend_n_[flag_word]
	void insert_range(const Batch& b, const OpRange& rg, bool staged, uint32_t flag_word = 1)
	{
		const uint64_t s = rg.s, e = rg.e;
		uint64_t T = rg.k1 - rg.k0;
		if (T == 0) return;
		loaded_ops_ += T;
		if (T > batch_ops_ || T >= 0xFFFFFFFFull) {
			fail_now(FAIL_INTERNAL, strf("a single sequence has more k-mers (%llu) than insert_batch_kmers", T));
		}
		// a view of sequences [s, e) whose op ids start at 0
		Batch v = b;
		v.woff = b.woff + s; v.len = b.len + s; v.n = e - s;
		// (the view keeps the whole batch's k-mer prefix sums: the functors subtract kbase)
		v.koff = b.koff + s;
		const uint64_t kbase = rg.k0;
		uint64_t cmask = (1ull << claim_log2_) - 1;
		if (epoch_ > 0xFFFFFF00u) { // claim epochs exhausted: start over
			for (int i = 0; i < 2; i++) be_.memset(claim_[i], 0xFF, 8ull << claim_log2_);
			epoch_ = 1;
		}
		uint64_t* ccur = claim_[0];
		uint64_t* cnext = claim_[1];
		if (dist()) {
			if (tiled_ && routed()) insert_tiles_routed(v, T, cmask, kbase);
			else if (tiled_) insert_tiles_dist(v, T, cmask, kbase, staged, staged ? flag_word : 1);
			else {
				FHashClaimT<true> fc{ p_, v, h0_, T, ccur, cmask, epoch_, own_lo_, own_span_, kbase };
				be_.launch((T + HC_RUN - 1) / HC_RUN, fc, "hash_claim");
				insert_rounds_dist(T, cmask, nullptr, T);
			}
			return;
		}
		uint64_t npend = T;
		const uint32_t* pin = nullptr; // NULL: all ops 0..T-1
		uint32_t* pout = pend_[0];
		bool claimed = false;
		if (tiled_) {
			// the k-mers that share no counter with another k-mer of the batch are settled tile by
			// tile; what is left goes through the reservation rounds below
			if (!staged) flag_word = 1;
			TileEnv te{ p_, cnt_, 0, m_, h0_, bins_, tile_cap_, tcur_, lead_, opflag_, tgt_, pendf_, pend_n_ + flag_word, cfg_.benign_sharers ? 1u : 0u };
			const bool judged = staged && purity_done_;
			if (!judged) {
				be_.memset(lead_, 0, T * 4);
				be_.memset(opflag_, 0, (T + 3) & ~3ull);
			}
			be_.memset(pend_n_, 0, 8);
			if (!staged) {
				be_.memset(tcur_, 0, ntiles_ * 4);
				dispatch_nw([&](auto nw) { FHashOps<(decltype(nw)::value & 7)> f{ p_, v, h0_, T, kbase, 0, hash_run_ }; be_.launch((T + hash_run_ - 1) / hash_run_, f, "hash_ops"); });
				BinEnv bn{ te, T, coarse_, coarse_cap_, ccur_, cshift_, ncoarse_ };
				be_.memset(ccur_, 0, ncoarse_ * 4);
				FBinCoarse f1{ bn };
				be_.launch_tiles((T + BIN_CHUNK_OPS - 1) / BIN_CHUNK_OPS, f1, "bin_coarse");
				const uint32_t cpb = (coarse_cap_ + BIN_CHUNK_PAIRS - 1) / BIN_CHUNK_PAIRS;
				FBinFine f2{ bn, cpb };
				be_.launch_tiles((uint64_t)ncoarse_ * cpb, f2, "bin_fine");
			}
			// A bin that ran over (a k-mer recurring thousands of times in the batch): the pairs are judged and applied through a
			// sort instead of the tiles (sorted_judge).  Staged batches wrote their flag word while the batch before took its rounds.
			SortBufs sb;
			bool sorted = false;
			if (cfg_.sorted_overflow) {
				uint32_t over = 0;
				be_.d2h(&over, pend_n_ + flag_word, 4);
				if (over) {
					be_.memset(lead_, 0, T * 4);
					be_.memset(opflag_, 0, (T + 3) & ~3ull);
					sorted = sorted_judge(te, T, sb);
					if (sorted) { be_.memset(pend_n_ + flag_word, 0, 4); stats_.tile_overflows++; }
				}
			}
			// (otherwise the tile kernels do nothing once a bin has overflowed: one read-back tells both the pending count and that)
			if (bad_) {
				te.bad = bad_; te.bad_mask = (uint32_t)((1ull << cfg_.cosettle_log2) - 1); te.wmask = wmask_; te.chg = cochg_; // (cosettle_log2 <= 25: CO_SLOT_MASK)
				te.colist = colist_; te.cocnt = cocnt_; te.glog2 = BE::ITEM_GROUP_LOG2;
				be_.memset(bad_, 0, (1ull << cfg_.cosettle_log2) / 8);
				be_.memset(cochg_, 0, (CO_MAX_PASSES + 1) * 4);
			}
			if (!judged && !sorted) { FTilePurity f{ te }; be_.launch_tiles(ntiles_, f, "tile_purity"); }
			{ FOpTarget f{ te }; be_.launch(T, f, "op_target"); }
			if (bad_) {
				// the k-mers that may raise shared counters: settled together, or sent to the rounds together (op_verdict)
				const uint32_t np = std::max(1u, std::min(cfg_.cosettle_passes, CO_MAX_PASSES));
				const uint64_t ngroups = (T + (1ull << te.glog2) - 1) >> te.glog2;
				for (uint32_t q = 1; q <= np; q++) { FCoSettle f{ te, q }; be_.launch(ngroups, f, "co_settle"); }
				FCoFinal f{ te, np }; be_.launch(ngroups, f, "co_settle");
			}
			if (sorted) { SortEnv se = sb.env(te, T); FSegApply f{ se }; be_.launch(se.np, f, "tile_apply"); sorted_free(sb); }
			else { FTileApply f{ te }; be_.launch_tiles(napply_, f, "tile_apply"); }
			{
				// the ops for the rounds, in op order (FPendCount)
				const uint64_t ng = (T + PEND_GROUP - 1) / PEND_GROUP;
				FPendCount fc{ pendf_, T, pgrp_ }; be_.launch(ng, fc, "compact");
				be_.inclusive_sum_u64(pgrp_, ng);
				FPendWrite fw{ pendf_, T, pgrp_, pend_[1], pend_n_ }; be_.launch(ng, fw, "compact");
			}
			// the next batch's hashing and binning starts here, beside the rounds (queued before tile_apply it
			// slows that down by as much as it gains: 453-463 vs 446-452 ms per configs[1] step)
			if (stage_next_) { stage_next_(); stage_next_ = nullptr; }
			uint32_t nn[4] = { 0, 0, 0, 0 };
			be_.d2h(nn, pend_n_, 16);
			if (nn[flag_word]) {
				stats_.tile_overflows++; // (a bin ran over: the whole batch takes the rounds)
				FClaim fc{ p_, h0_, ccur, cmask, epoch_ };
				be_.launch(T, fc, "hash_claim");
			} else {
				stats_.tiled_ops += T; stats_.tiled_pending += nn[0];
				npend = nn[0];
				pin = pend_[1];
				if (npend) { FClaimList fc{ p_, h0_, pin, ccur, cmask, epoch_ }; be_.launch(npend, fc, "claim_list"); }
			}
			claimed = true;
		}
		if (!claimed) {
			FHashClaim fc{ p_, v, h0_, T, ccur, cmask, epoch_, 0, 0, kbase };
			be_.launch((T + HC_RUN - 1) / HC_RUN, fc, "hash_claim");
		}
		while (npend) {
			if (pin && npend <= cfg_.drain_threshold) {
				// few ops left: finish all remaining rounds inside one workgroup
				uint32_t* other = (pin == pend_[0]) ? pend_[1] : pend_[0];
				InsertDrainEnv de{ p_, h0_, cnt_, casc_, const_cast<uint32_t*>(pin), other, (uint32_t)npend,
					ccur, cnext, cmask, epoch_, pend_n_ };
				be_.launch_drain(de);
				uint32_t r[2] = { 0, 0 };
				be_.d2h(r, pend_n_, 8);
				epoch_ += r[1];
				last_rounds_ += r[1];
				stats_.insert_rounds += r[1];
				break;
			}
			be_.memset(pend_n_, 0, 4);
			const bool flagged = npend >= cfg_.compact_threshold;
			FInsertRound fr{ p_, h0_, cnt_, casc_, pin, pout, pend_n_, ccur, cnext, cmask, epoch_, flagged ? dlost_ : nullptr };
			be_.launch(npend, fr, pin ? "insert_retry" : "insert_round");
			if (flagged) be_.compact_flagged(pin, dlost_, npend, pout, pend_n_);
			uint32_t nn = 0;
			be_.d2h(&nn, pend_n_, 4);
			npend = nn;
			// losers claimed in `cnext` under epoch + 1; stale older-epoch values lose to them
			std::swap(ccur, cnext);
			pin = pout;
			pout = (pout == pend_[0]) ? pend_[1] : pend_[0];
			epoch_++;
			last_rounds_++;
			stats_.insert_rounds++;
		}
		epoch_++;
	}

	// The pairs of a batch sorted by counter and judged segment by segment (SortEnv): `lead` and `opflag` come out as
	// tile_purity would leave them with bins of any size.  false: no memory for it (the batch then takes the rounds as a whole).
	struct SortBufs {
		uint64_t* key = nullptr; uint32_t* val = nullptr; uint64_t* sid = nullptr; uint32_t* start = nullptr; uint8_t* impure = nullptr; uint64_t np = 0;
		SortEnv env(const TileEnv& te, uint64_t T) const { return SortEnv{ te, T, np, key + np, val + np, sid, start, impure }; }
	};
	void sorted_free(SortBufs& sb)
	{
		be_.sync(); // (blocks of this size are not recycled: they go back to the device once nothing queued reads them)
		be_.free(sb.key); be_.free(sb.val); be_.free(sb.sid); be_.free(sb.start); be_.free(sb.impure);
		sb = SortBufs();
	}
	bool sorted_judge(const TileEnv& te, uint64_t T, SortBufs& sb)
	{
		const uint64_t np = T * p_.nh;
		if (p_.nh > 16 || np >= 0x7FFFFFFFull) return false;
		sb.np = np;
		sb.key = (uint64_t*)be_.try_alloc(2 * np * 8); sb.val = (uint32_t*)be_.try_alloc(2 * np * 4);
		sb.sid = (uint64_t*)be_.try_alloc(np * 8); sb.start = (uint32_t*)be_.try_alloc((np + 1) * 4); sb.impure = (uint8_t*)be_.try_alloc(np);
		if (!sb.key || !sb.val || !sb.sid || !sb.start || !sb.impure) {
			if (sb.key) be_.free(sb.key); if (sb.val) be_.free(sb.val); if (sb.sid) be_.free(sb.sid); if (sb.start) be_.free(sb.start); if (sb.impure) be_.free(sb.impure);
			sb = SortBufs();
			return false;
		}
		{ SortEnv in{ te, T, np, sb.key, sb.val, sb.sid, sb.start, sb.impure }; FSortKeys f{ in }; be_.launch(np, f, "sort_judge"); }
		be_.sort_pairs_u64_u32(sb.key, sb.key + np, sb.val, sb.val + np, np);
		const SortEnv se = sb.env(te, T);
		{ FSegHeads f{ se }; be_.launch(np, f, "sort_judge"); }
		be_.inclusive_sum_u64(sb.sid, np);
		{ FSegStarts f{ se }; be_.launch(np, f, "sort_judge"); }
		{ FSegImpure f{ se }; be_.launch(np, f, "sort_judge"); }
		{ FSegJudge f{ se }; be_.launch(np, f, "sort_judge"); }
		return true;
	}

	// The tiles of insert_range over a range-partitioned filter.  Each rank hashes a slice of the
	// batch's ops and the slices are all-gathered (8 bytes per op instead of every rank hashing
	// every op); each rank then bins the pairs on the counters it owns into its own tiles and
	// judges them there; three bytes per op through one all_reduce(MAX) tell every rank which ops
	// share a counter anywhere, which ops lead their k-mer and what the minimum of a leader's
	// counters is (FDistPack); every rank applies the leaders' targets to its tiles, and the ops
	// left over -- the same list on every rank -- go through the partitioned reservation rounds.
	void insert_tiles_dist(const Batch& v, uint64_t T, uint64_t cmask, uint64_t kbase, bool staged, uint32_t flag_word)
	{
		cnt_partial_ = true;
		const uint64_t R = (uint64_t)comm_.world, me = (uint64_t)comm_.rank;
		TileEnv te{ p_, cnt_, own_lo_, own_lo_ + own_span_, h0_, bins_, tile_cap_, tcur_, lead_, opflag_, tgt_, pendf_, pend_n_ + flag_word, cfg_.benign_sharers ? 1u : 0u, 16u, 254u };
		const bool judged = staged && purity_done_;
		if (!judged) {
			be_.memset(lead_, 0, T * 4);
			be_.memset(opflag_, 0, (T + 3) & ~3ull);
		}
		be_.memset(pend_n_, 0, 8);
		if (staged) {
			// (hashes gathered and the own range's pairs binned while the batch before went through its rounds: stage_bins)
		} else if (R <= cfg_.dist_hash_all_ranks) {
			be_.memset(tcur_, 0, ntiles_ * 4);
			// (two ranks share one xGMI link: hashing the other half of the ops costs what receiving it does)
			dispatch_nw([&](auto nw) { FHashOps<(decltype(nw)::value & 7)> f{ p_, v, h0_, T, kbase, 0, hash_run_ }; be_.launch((T + hash_run_ - 1) / hash_run_, f, "hash_ops"); });
		} else {
			// equal slices (one ring all-gather); the last one may run into the slack behind the T hashes
			const uint64_t chunk = ((T + R - 1) / R + 7) & ~7ull;
			std::vector<uint64_t> c(R, chunk * 8), d(R);
			for (uint64_t q = 0; q < R; q++) d[q] = q * chunk * 8;
			const uint64_t a = std::min(T, me * chunk), b = std::min(T, a + chunk);
			be_.memset(tcur_, 0, ntiles_ * 4);
			if (b > a)
				dispatch_nw([&](auto nw) { FHashOps<(decltype(nw)::value & 7)> f{ p_, v, h0_, b, kbase, a, hash_run_ }; be_.launch((b - a + hash_run_ - 1) / hash_run_, f, "hash_ops"); });
			c_all_gather_v(h0_, c.data(), d.data());
		}
		if (!staged) {
			BinEnv bn{ te, T, coarse_, coarse_cap_, ccur_, cshift_, ncoarse_ };
			be_.memset(ccur_, 0, ncoarse_ * 4);
			FBinCoarse f1{ bn };
			be_.launch_tiles((T + BIN_CHUNK_OPS - 1) / BIN_CHUNK_OPS, f1, "bin_coarse");
			const uint32_t cpb = (coarse_cap_ + BIN_CHUNK_PAIRS - 1) / BIN_CHUNK_PAIRS;
			FBinFine f2{ bn, cpb };
			be_.launch_tiles((uint64_t)ncoarse_ * cpb, f2, "bin_fine");
		}
		if (!judged) { FTilePurity f{ te }; be_.launch_tiles(ntiles_, f, "tile_purity"); }
		if (bad_) {
			// the single-GPU rules (op_verdict, FCoSettle) on what the ranks know together: FDistPack2
			uint8_t* mx = tred_; uint8_t* sm = tred_ + (((2 + p_.nh) * T + 1 + 63) & ~63ull);
			{ FDistPack2 f{ te, T, mx, sm }; be_.launch(T, f, "dist_pack"); }
			c_all_reduce(mx, (2 + p_.nh) * T + 1, DT_U8, OP_MAX);
			c_all_reduce(sm, T, DT_U8, OP_SUM);
			{ FDistUnpack2 f{ te, T, mx, sm }; be_.launch(T, f, "dist_pack"); }
			te.cval = mx + 2 * T;
			te.bad = bad_; te.bad_mask = (uint32_t)((1ull << cfg_.cosettle_log2) - 1); te.wmask = wmask_; te.chg = cochg_;
			te.colist = colist_; te.cocnt = cocnt_; te.glog2 = BE::ITEM_GROUP_LOG2;
			be_.memset(bad_, 0, (1ull << cfg_.cosettle_log2) / 8);
			be_.memset(cochg_, 0, (CO_MAX_PASSES + 1) * 4);
			{ FOpTarget f{ te }; be_.launch(T, f, "op_target"); }
			const uint32_t np = std::max(1u, std::min(cfg_.cosettle_passes, CO_MAX_PASSES));
			const uint64_t ngroups = (T + (1ull << te.glog2) - 1) >> te.glog2;
			for (uint32_t q = 1; q <= np; q++) { FCoSettle f{ te, q }; be_.launch(ngroups, f, "co_settle"); }
			{ FCoFinal f{ te, np }; be_.launch(ngroups, f, "co_settle"); }
			{ FTileApply f{ te }; be_.launch_tiles(napply_, f, "tile_apply"); }
			const uint64_t ng = (T + PEND_GROUP - 1) / PEND_GROUP;
			FPendCount fc{ pendf_, T, pgrp_ }; be_.launch(ng, fc, "compact");
			be_.inclusive_sum_u64(pgrp_, ng);
			FPendWrite fw{ pendf_, T, pgrp_, pend_[1], pend_n_ }; be_.launch(ng, fw, "compact");
		} else {
			{ FDistPack f{ te, T, tred_ }; be_.launch(T, f, "dist_pack"); }
			c_all_reduce(tred_, 2 * T + 1, DT_U8, OP_MAX);
			{ FDistTarget f{ te, T, tred_ }; be_.launch(T, f, "op_target"); }
			{ FTileApply f{ te }; be_.launch_tiles(napply_, f, "tile_apply"); }
			be_.compact_flagged(nullptr, pendf_, T, pend_[1], pend_n_);
		}
		if (stage_next_) { stage_next_(); stage_next_ = nullptr; } // (the next batch, beside this one's rounds)
		uint32_t nn[4] = { 0, 0, 0, 0 };
		be_.d2h(nn, pend_n_, 16);
		const uint32_t* pin = nullptr;
		uint64_t npend = T;
		if (nn[flag_word]) stats_.tile_overflows++;
		else { stats_.tiled_ops += T; stats_.tiled_pending += nn[0]; npend = nn[0]; pin = pend_[1]; }
		if (npend) {
			FClaimOwned fc{ p_, h0_, pin, claim_[0], cmask, epoch_, own_lo_, own_span_ };
			be_.launch(npend, fc, "claim_list");
		}
		insert_rounds_dist(T, cmask, pin, npend);
	}

	// The tiles of insert_range over a range-partitioned filter, ROUTED: a rank hashes its slice of the batch's ops and
	// sends every (op, counter) pair to the rank that owns the counter (FRoutePack, one all_to_all_v of 12-byte records);
	// each rank bins what it received into its tiles and judges it there (FBinCoarseRec, FBinFine, tile_purity as ever);
	// two bytes per pair travel back (FRouteReply: shared / leads n ops, 255 - the counter), the hashing rank folds its
	// nh replies into the op's verdict and target (FRouteCombine) and sends the target to the pairs' owners, who apply it
	// to their tiles.  The ops left over are made known to everybody ({hash, op}, all-gathered in op order) and go through
	// the partitioned reservation rounds.  Per op of a rank's own share: nh x (12 + 2 + 1) bytes out or in, to or from the
	// ranks owning its counters, whatever the number of ranks; no rank looks at an op it neither hashed nor owns a counter of.
	// If any room runs out anywhere (a destination's records, a bin), every rank falls back to the whole batch through the
	// rounds (the hashes are all-gathered for that): nothing has been applied by then.
	void insert_tiles_routed(const Batch& v, uint64_t T, uint64_t cmask, uint64_t kbase)
	{
		cnt_partial_ = true;
		const uint64_t R = (uint64_t)comm_.world, me = (uint64_t)comm_.rank;
		const uint64_t chunk = ((T + R - 1) / R + 7) & ~7ull;
		const uint64_t a = std::min(T, me * chunk), b = std::min(T, a + chunk), nown = b - a;
		uint32_t* rflag = rcur_ + MAX_RANKS; // [0] some room ran out on this rank
		TileEnv te{ p_, cnt_, own_lo_, own_lo_ + own_span_, h0_, bins_, tile_cap_, tcur_, lead_, opflag_, tgt_, pendf_, rflag, 0u, 16u, 127u }; // (127: FRouteReply's byte)
		be_.memset(lead_, 0, T * 4);
		be_.memset(opflag_, 0, (T + 3) & ~3ull);
		be_.memset(tcur_, 0, ntiles_ * 4);
		be_.memset(ccur_, 0, ncoarse_ * 4);
		be_.memset(rcur_, 0, (MAX_RANKS + 2) * 4);
		if (nown) {
			dispatch_nw([&](auto nw) { FHashOps<(decltype(nw)::value & 7)> f{ p_, v, h0_, b, kbase, a, hash_run_ }; be_.launch((nown + hash_run_ - 1) / hash_run_, f, "hash_ops"); });
			RouteEnv re{ p_, h0_, a, b, own_chunk_, (uint32_t)R, rsend_, rcap_, rcur_, rslot_, rflag };
			FRoutePack f{ re };
			be_.launch_tiles((nown + BIN_CHUNK_OPS - 1) / BIN_CHUNK_OPS, f, "route_pack");
		}
		// who sends how much to whom (and whether anybody ran out of room): one small all-reduce on the host's behalf
		std::vector<uint32_t> sc32(MAX_RANKS + 2, 0);
		be_.d2h(sc32.data(), rcur_, (MAX_RANKS + 2) * 4);
		std::vector<uint64_t> mat(R * R + 1, 0);
		for (uint64_t q = 0; q < R; q++) mat[me * R + q] = std::min<uint64_t>(sc32[q], rcap_);
		mat[R * R] = sc32[MAX_RANKS];
		host_all_reduce_sum(mat.data(), (uint32_t)(R * R + 1));
		bool fallback = mat[R * R] != 0;
		std::vector<uint64_t> sc(R), sd(R), rc(R), rd(R);
		uint64_t nrec = 0;
		for (uint64_t q = 0; q < R; q++) { sc[q] = mat[me * R + q]; sd[q] = q * rcap_; rc[q] = mat[q * R + me]; rd[q] = nrec; nrec += rc[q]; }
		if (!fallback && nrec > rrecv_cap_) fallback = true; // (cannot happen while every sender keeps within rcap_: R x rcap_ is the room)
		auto bytes = [&](const std::vector<uint64_t>& x, uint64_t w) { std::vector<uint64_t> y(x); for (auto& e : y) e *= w; return y; };
		if (!fallback) {
			c_all_to_all_v(rsend_, bytes(sc, sizeof(TilePair)).data(), bytes(sd, sizeof(TilePair)).data(), rrecv_, bytes(rc, sizeof(TilePair)).data(), bytes(rd, sizeof(TilePair)).data());
			if (nrec) {
				BinEnv bn{ te, T, coarse_, coarse_cap_, ccur_, cshift_, ncoarse_ };
				FBinCoarseRec f1{ bn, rrecv_, nrec };
				be_.launch_tiles((nrec + BIN_CHUNK_PAIRS - 1) / BIN_CHUNK_PAIRS, f1, "bin_coarse");
				const uint32_t cpb = (coarse_cap_ + BIN_CHUNK_PAIRS - 1) / BIN_CHUNK_PAIRS;
				FBinFine f2{ bn, cpb };
				be_.launch_tiles((uint64_t)ncoarse_ * cpb, f2, "bin_fine");
				FTilePurity f3{ te };
				be_.launch_tiles(ntiles_, f3, "tile_purity");
			}
			// a bin that ran over anywhere sends the whole batch through the rounds on every rank
			uint64_t ovf = 0;
			{ uint32_t fl = 0; be_.d2h(&fl, rflag, 4); ovf = fl; }
			host_all_reduce_sum(&ovf, 1);
			fallback = ovf != 0;
		}
		if (fallback) {
			stats_.tile_overflows++;
			if (getenv("ABG_ROUTE_DEBUG")) fprintf(stderr, "[route] rank %d fallback: T %llu pack flag %llu nrec %llu cap %u tile_cap %u coarse_cap %u ntiles %llu\n", comm_.rank, (unsigned long long)T, (unsigned long long)mat[R * R], (unsigned long long)nrec, rcap_, tile_cap_, coarse_cap_, (unsigned long long)ntiles_);
			std::vector<uint64_t> c(R, chunk * 8), d(R);
			for (uint64_t q = 0; q < R; q++) d[q] = q * chunk * 8;
			c_all_gather_v(h0_, c.data(), d.data());
			FClaimOwned fc{ p_, h0_, nullptr, claim_[0], cmask, epoch_, own_lo_, own_span_ };
			be_.launch(T, fc, "claim_list");
			insert_rounds_dist(T, cmask, nullptr, T);
			return;
		}
		if (nrec) { FRouteReply f{ te, rrecv_, rrep_out_ }; be_.launch(nrec, f, "route_reply"); }
		c_all_to_all_v(rrep_out_, bytes(rc, 2).data(), bytes(rd, 2).data(), rrep_in_, bytes(sc, 2).data(), bytes(sd, 2).data());
		if (nown) { FRouteCombine f{ p_, rslot_, rrep_in_, rtgt_out_, rpendf_, cfg_.benign_sharers ? 1u : 0u }; be_.launch(nown, f, "op_target"); }
		c_all_to_all_v(rtgt_out_, sc.data(), sd.data(), rtgt_in_, rc.data(), rd.data());
		if (nrec) {
			FRouteTgt f{ rrecv_, rtgt_in_, tgt_ };
			be_.launch(nrec, f, "route_tgt");
			FTileApply fa{ te };
			be_.launch_tiles(napply_, fa, "tile_apply");
		}
		// the ops left over: each rank's own (in op order), made known to all
		be_.memset(pend_n_, 0, 8);
		if (nown) be_.compact_flagged(nullptr, rpendf_, nown, pend_[0], pend_n_);
		uint32_t mine = 0;
		be_.d2h(&mine, pend_n_, 4);
		std::vector<uint64_t> pc(R, 0);
		pc[me] = mine;
		host_all_reduce_sum(pc.data(), (uint32_t)R);
		uint64_t npend = 0;
		std::vector<uint64_t> gc(R), gd(R);
		for (uint64_t q = 0; q < R; q++) { gd[q] = npend * sizeof(TilePair); gc[q] = pc[q] * sizeof(TilePair); npend += pc[q]; }
		stats_.tiled_ops += T; stats_.tiled_pending += npend;
		if (npend) {
			// (the send buffer is free again: the gathered records go there -- at most T of them, so make room if need be)
			TilePair* g = rsend_;
			TilePair* big = nullptr;
			if (npend > R * (uint64_t)rcap_) { big = (TilePair*)be_.alloc(npend * sizeof(TilePair)); g = big; }
			if (mine) { FRoutePendRec f{ h0_, pend_[0], a, (TilePair*)((char*)g + gd[me]) }; be_.launch(mine, f, "route_pend"); }
			c_all_gather_v(g, gc.data(), gd.data());
			FRoutePendTake f{ g, h0_, pend_[1] };
			be_.launch(npend, f, "route_pend");
			if (big) { be_.sync(); be_.free(big); }
			FClaimOwned fc{ p_, h0_, pend_[1], claim_[0], cmask, epoch_, own_lo_, own_span_ };
			be_.launch(npend, fc, "claim_list");
		}
		insert_rounds_dist(T, cmask, pend_[1], npend);
	}
	// The reservation rounds of insert_range over a range-partitioned filter: the same rounds,
	// every rank running every pending op against the counters it owns, one all_reduce(MIN) of a
	// byte per op between "who holds all claims / what is the minimum" and "apply".  The list of
	// losers is compacted in op order, so it is the same list on every rank.  The caller has made
	// the first claims (in claim_[0], under epoch_) for the ops of `pin` (NULL: all T ops).
	void insert_rounds_dist(uint64_t T, uint64_t cmask, const uint32_t* pin, uint64_t npend)
	{
		(void)T;
		cnt_partial_ = true;
		uint64_t* ccur = claim_[0];
		uint64_t* cnext = claim_[1];
		uint32_t* pout = pend_[0];
		while (npend) {
			if (pin && npend <= cfg_.drain_threshold) {
				uint8_t* val = dres_;
				{ FDrainVals f{ p_, h0_, cnt_, pin, own_lo_, own_span_, val }; be_.launch(npend, f, "drain_vals"); }
				c_all_reduce(val, npend * p_.nh, DT_U8, OP_MAX);
				{ FDrainLoad f{ p_, h0_, cnt_, pin, own_lo_, own_span_, val, ccur, cmask, epoch_ }; be_.launch(npend, f, "drain_load"); }
				uint32_t* other = (pin == pend_[0]) ? pend_[1] : pend_[0];
				InsertDrainEnv de{ p_, h0_, cnt_, casc_, const_cast<uint32_t*>(pin), other, (uint32_t)npend,
					ccur, cnext, cmask, epoch_, pend_n_ };
				be_.launch_drain(de);
				uint32_t r[2] = { 0, 0 };
				be_.d2h(r, pend_n_, 8);
				epoch_ += r[1];
				last_rounds_ += r[1];
				stats_.insert_rounds += r[1];
				break;
			}
			{ FEvalDist f{ p_, h0_, cnt_, pin, ccur, cmask, epoch_, own_lo_, own_span_, dres_ }; be_.launch(npend, f, pin ? "insert_retry" : "insert_round"); }
			c_all_reduce(dres_, npend, DT_U8, OP_MIN);
			{ FApplyDist f{ p_, h0_, cnt_, pin, cnext, cmask, epoch_, own_lo_, own_span_, dres_, dlost_ }; be_.launch(npend, f, "insert_apply"); }
			be_.compact_flagged(pin, dlost_, npend, pout, pend_n_);
			uint32_t nn = 0;
			be_.d2h(&nn, pend_n_, 4);
			npend = nn;
			std::swap(ccur, cnext);
			pin = pout;
			pout = (pout == pend_[0]) ? pend_[1] : pend_[0];
			epoch_++;
			last_rounds_++;
			stats_.insert_rounds++;
		}
		epoch_++;
	}

	void ensure_walk()
	{
		if (walk_ready_) return;
		if (!cend_.hmin) { // (shared by all contexts)
			p2_batch_ = cfg_.p2_first_batch;
			cslots_ = std::min<uint32_t>(be_.max_slots(), cfg_.classify_slots);
			alloc_tab(cend_, cfg_.cend_log2);
		}
		wslots_ = std::min<uint32_t>(be_.max_slots(), cfg_.walk_slots);
		wtab_log2_ = cfg_.wtab_log2;
		alloc_tab(wtab_, wtab_log2_);
		la_pool_ = (VKey*)be_.alloc((uint64_t)wslots_ * LA_MAX_VISITED * sizeof(VKey));
		bulk_pool_ = (BulkScratch*)be_.alloc((uint64_t)wslots_ * sizeof(BulkScratch));
		if (!wstats_) { wstats_ = (uint64_t*)be_.alloc(WSTAT_N * 8); clear_wstats(); }
		walk_tb_cap_ = cfg_.tb_cap;
		walk_buf_cap_ = cfg_.buf_cap;
		alloc_walk_scratch();
		pool_cap_ = cfg_.pool_cap;
		pool_ = (uint8_t*)be_.alloc(pool_cap_);
		kh_ = (uint64_t*)be_.alloc(pool_cap_ * 8);
		pool_used_ = (uint64_t*)be_.alloc(8);
		rec_cap_ = cfg_.rec_cap;
		recs_ = (ContigRec*)be_.alloc((uint64_t)rec_cap_ * sizeof(ContigRec));
		rec_used_ = (uint32_t*)be_.alloc(8);
		order_ = (uint32_t*)be_.alloc((uint64_t)rec_cap_ * 4);
		order_n_ = (uint32_t*)be_.alloc(8);
		walk_ready_ = true;
	}
	void alloc_walk_scratch()
	{
		tb_pool_ = be_.alloc((uint64_t)wslots_ * walk_tb_cap_ * sizeof(TBFrame<NW_MASKED + MAX_NW>));
		tbk_pool_ = (VKey*)be_.alloc((uint64_t)wslots_ * walk_tb_cap_ * sizeof(VKey));
		lbuf_ = (uint8_t*)be_.alloc((uint64_t)wslots_ * walk_buf_cap_);
		rbuf_ = (uint8_t*)be_.alloc((uint64_t)wslots_ * walk_buf_cap_);
	}
	void free_walk_scratch() { be_.free(tb_pool_); be_.free(tbk_pool_); be_.free(lbuf_); be_.free(rbuf_); }
	void alloc_tab(WalkTab& t, uint32_t log2)
	{
		uint64_t cap = 1ull << log2;
		t.hmin = (uint64_t*)be_.alloc(cap * 8);
		t.hmax = (uint64_t*)be_.alloc(cap * 8);
		t.meta = (uint64_t*)be_.alloc(cap * 8);
		t.mask = cap - 1;
		be_.memset(t.hmin, 0xFF, cap * 8);
		be_.memset(t.meta, 0xFF, cap * 8);
	}
	void free_tab(WalkTab& t) { be_.free(t.hmin); be_.free(t.hmax); be_.free(t.meta); }
	void free_walk()
	{
		if (!walk_ready_) return;
		free_tab(wtab_);
		be_.free(la_pool_); be_.free(bulk_pool_);
		free_walk_scratch();
		be_.free(pool_); be_.free(kh_); be_.free(pool_used_); be_.free(recs_); be_.free(rec_used_);
		be_.free(order_); be_.free(order_n_);
		walk_ready_ = false;
	}
	// A round made no progress because its first candidate ran out of some capacity:
	// enlarge everything a single walk can exhaust.
	void grow_walk_resources()
	{
		free_walk_scratch();
		walk_tb_cap_ *= 4;
		walk_buf_cap_ *= 8;
		if (walk_buf_cap_ > (1u << 28) || wtab_log2_ > 31) {
			fail_now(FAIL_INTERNAL, "a unitig exceeds the walker limits");
		}
		// fewer walkers when each needs a lot of scratch
		while ((uint64_t)wslots_ * walk_buf_cap_ > (8ull << 30) && wslots_ > 64) wslots_ /= 2;
		alloc_walk_scratch();
		free_tab(wtab_);
		wtab_log2_++;
		alloc_tab(wtab_, wtab_log2_);
		be_.free(pool_); be_.free(kh_);
		pool_cap_ *= 2;
		pool_ = (uint8_t*)be_.alloc(pool_cap_);
		kh_ = (uint64_t*)be_.alloc(pool_cap_ * 8);
		if (cfg_.verbose)
			fprintf(stderr, "abyss_amd: walker resources grown: %u frames, %u bases, 2^%u table, %llu pool\n",
			    walk_tb_cap_, walk_buf_cap_, wtab_log2_, (unsigned long long)pool_cap_);
	}

	template <int NW>
	WalkEnv<NW> make_env(const Batch& b, uint32_t* cand_d, uint32_t* status_d, uint32_t* first_d)
	{
		WalkEnv<NW> e;
		e.p = p2_; e.cnt = cnt2_; e.batch = b; e.cand_read = cand_d; e.status = status_d; e.first_rec = first_d;
		e.tab = wtab_; e.owner_base = 0;
		// scratch strides are sized for the largest TBFrame; a smaller NW fits more frames in them
		e.tb_pool = (TBFrame<NW>*)tb_pool_;
		e.tbk_pool = tbk_pool_;
		e.tb_cap = walk_tb_cap_;
		e.fast = nullptr; e.fast_bytes = 0; e.dbg = dbg_; e.coop = false;
		e.la_pool = la_pool_;
		e.guide = guide_; e.bulk_pool = bulk_pool_; e.wstats = wstats_; e.memo = memo_; e.mcache = nullptr;
		e.lbuf_pool = lbuf_; e.rbuf_pool = rbuf_; e.buf_cap = walk_buf_cap_;
		e.pool = pool_; e.pool_cap = pool_cap_; e.pool_used = pool_used_;
		e.recs = recs_; e.rec_cap = rec_cap_; e.rec_used = rec_used_;
		return e;
	}

	template <int NW>
	uint32_t commit(const Batch& b, uint32_t* cand_d, uint32_t* status_d, uint32_t* first_d,
	    uint8_t* result_d, const uint64_t* rkoff_d, uint32_t c_begin, uint32_t c_end)
	{
		// contigEndKmers grows ahead of demand: every record not yet committed may add two keys
		{
			uint32_t nrec = 0, nord = 0;
			be_.d2h(&nrec, rec_used_, 4);
			be_.d2h(&nord, order_n_, 4);
			uint64_t need = cend_count_ + 2ull * (std::min(nrec, rec_cap_) - std::min(nord, rec_cap_));
			while (need * 2 > cend_.mask + 1) grow_cend();
		}
		CommitState cs;
		cs.counters = counters_;
		cs.break_at = c_begin; cs.pad_ = 0; cs.cend_count = cend_count_;
		be_.h2d(cstate_, &cs, sizeof cs);
		{
			FPreCommit<NW> f{ p_, b, cand_d, status_d, first_d, recs_, vis_, kh_, rkh_, rkoff_d, read_flag_, c_begin, 0, ~0ULL, nullptr, nullptr };
			be_.launch_wave(c_end - c_begin, f, "precommit");
			FPreCommitCopies fc{ status_d, first_d, recs_, read_flag_, c_begin, p_.k };
			be_.launch(c_end - c_begin, fc, "precommit");
		}
		CommitEnv<NW> e;
		e.read_flag = read_flag_;
		e.p = p_; e.b = b; e.vis32 = (uint32_t*)vis_; e.both32 = (uint32_t*)both_;
		e.cand_read = cand_d; e.status = status_d; e.first_rec = first_d;
		e.recs = recs_; e.pool = pool_; e.result = result_d;
		e.kh = kh_; e.rkh = rkh_; e.rkoff = rkoff_d;
		e.cend = cend_; e.st = cstate_;
		e.order = order_; e.order_n = order_n_;
		be_.template launch_commit<NW>(e, c_begin, c_end);
		be_.d2h(&cs, cstate_, sizeof cs);
		counters_ = cs.counters;
		cend_count_ = cs.cend_count;
		if (cs.pad_) fail_now(FAIL_INTERNAL, "contigEndKmers table overflowed");
		return cs.break_at;
	}

	// The same commit as a parallel fixed-point computation (see ParCommit).
	template <int NW>
	uint32_t commit_par(const Batch& b, uint32_t* cand_d, uint32_t* status_d, uint32_t* first_d,
	    uint8_t* result_d, uint64_t* rkoff_d, uint32_t c_begin, uint32_t c_end)
	{
		uint32_t n = c_end - c_begin;
		const uint32_t n0 = n; // (n may shrink below: a commit holds 2^T_TIME_BITS records)
		uint32_t nrec = 0, nord = 0;
		be_.d2h(&nrec, rec_used_, 4);
		be_.d2h(&nord, order_n_, 4);
		nrec = std::min(nrec, rec_cap_);
		{
			uint64_t need = cend_count_ + 2ull * (nrec - std::min(nord, rec_cap_));
			while (need * 2 > cend_.mask + 1) grow_cend();
		}
		// partitioned run: every rank holds the visited filter and every contig, but tests and stamps
		// the bits of its own range only; a byte per candidate and per record goes through all_reduce
		const bool part = dist();
		uint8_t* part_buf = part ? (uint8_t*)be_.alloc((uint64_t)n + nrec + 64) : nullptr;
		uint8_t* part_c = part_buf; uint8_t* part_r = part ? part_buf + n : nullptr;
		{
			if (part) be_.memset(part_buf, 1, (uint64_t)n + nrec);
			FPreCommit<NW> f{ p_, b, cand_d, status_d, first_d, recs_, vis_, kh_, rkh_, rkoff_d, read_flag_, c_begin,
				part ? own_lo_ : 0, part ? own_span_ : ~0ULL, part_c, part_r };
			be_.launch_wave(n, f, "precommit");
			if (!part) { FPreCommitCopies fc{ status_d, first_d, recs_, read_flag_, c_begin, p_.k }; be_.launch(n, fc, "precommit"); }
			if (part) {
				c_all_reduce(part_buf, (uint64_t)n + nrec, DT_U8, OP_MIN);
				FPreCommitFin ff{ status_d, first_d, recs_, read_flag_, c_begin, p_.k, part_c, part_r };
				be_.launch(n, ff, "precommit");
			}
		}
		ParCommit e;
		e.own_lo = part ? own_lo_ : 0; e.own_span = part ? own_span_ : ~0ULL; e.part_c = part_c; e.part_r = part_r;
		e.p = p_; e.b = b; e.cand_read = cand_d; e.status = status_d; e.first_rec = first_d;
		e.recs = recs_; e.pool = pool_; e.result = result_d; e.kh = kh_; e.rkh = rkh_; e.rkoff = rkoff_d;
		e.read_flag = read_flag_; e.vis32 = (uint32_t*)vis_; e.both32 = (uint32_t*)both_; e.T = T_; e.cend = cend_; e.cnt8 = cnt_;
		if (!part) e.arc = arc_;
		e.Tk = nullptr; e.Tv = nullptr; e.Tmask = 0;
		if (t_hashed()) {
			// the bits this commit can stamp: (k-mers of the contigs in the pool) x H
			uint64_t bases = 0;
			be_.d2h(&bases, pool_used_, 8);
			const uint64_t need = std::min<uint64_t>(bases, pool_cap_) * p_.nh + 16;
			if (!Tk_ || (T_entries_ + need) * 2 > (1ull << Tlog2_)) {
				uint32_t log2 = 20;
				while ((1ull << log2) < 2 * need) log2++;
				if (!Tk_ || log2 > Tlog2_) {
					if (Tk_) { be_.free(Tk_); be_.free(Tv_); }
					Tlog2_ = log2;
					Tk_ = (uint64_t*)be_.alloc(8ull << Tlog2_);
					Tv_ = (uint32_t*)be_.alloc(4ull << Tlog2_);
				}
				be_.memset(Tk_, 0xFF, 8ull << Tlog2_);
				be_.memset(Tv_, 0xFF, 4ull << Tlog2_);
				T_entries_ = 0;
				t_tag_ = std::min<uint32_t>(cfg_.t_tags, T_TAGS) - 1;
			}
			T_entries_ += need;
			e.T = nullptr; e.Tk = Tk_; e.Tv = Tv_; e.Tmask = (1ull << Tlog2_) - 1;
		}
		e.c_begin = c_begin; e.c_end = c_end; e.brk = c_end;
		e.off = (uint32_t*)be_.alloc((n + 1ull) * 4); e.cnt = (uint32_t*)be_.alloc(n * 4ull + 4);
		e.cnt2 = (uint32_t*)be_.alloc(n * 4ull + 4); e.cnt3 = (uint64_t*)be_.alloc(n * 8ull + 8);
		e.active = (uint8_t*)be_.alloc(n + 8ull); e.short_list = (uint32_t*)be_.alloc(nrec * 4ull + 4);
		e.scal = (uint32_t*)be_.alloc(64);
		uint32_t scal_h[8] = { 0, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0 };
		be_.h2d(e.scal, scal_h, sizeof scal_h);
		e.tcend = WalkTab{ nullptr, nullptr, nullptr, 0 };
		// commit positions
		std::vector<uint32_t> c1(n), c2(n), off(n + 1ull);
		{ FPcCount f{ e }; be_.launch(n, f, "pc_count"); }
		be_.d2h(c1.data(), e.cnt, n * 4ull);
		off[0] = 0;
		for (uint32_t i = 0; i < n; i++) off[i + 1] = off[i] + c1[i];
		// (a stamp holds 2^T_TIME_BITS commit positions: a range with more contig records than that is
		// committed up to there, and the caller comes back for the rest)
		if (off[n] > (1u << T_TIME_BITS)) {
			uint32_t keep = 1;
			while (keep < n && off[keep + 1] <= (1u << T_TIME_BITS)) keep++;
			if (off[keep] > (1u << T_TIME_BITS)) fail_now(FAIL_INTERNAL, "one read produced more contigs than a commit can order");
			n = keep; c_end = c_begin + n; e.c_end = c_end; e.brk = c_end;
		}
		be_.h2d(e.off, off.data(), (n + 1ull) * 4);
		{ FPcStamp f{ e }; be_.launch(n, f, "pc_stamp"); }
		be_.d2h(scal_h, e.scal, sizeof scal_h);
		const uint32_t nshort = scal_h[2];
		e.short_keys = nullptr;
		if (nshort) {
			uint32_t log2 = 4;
			while ((1ull << log2) < 4ull * nshort) log2++;
			alloc_tab(e.tcend, log2);
			e.short_keys = (VKey*)be_.alloc(2ull * nshort * sizeof(VKey));
			FPcShortKeys f{ e };
			be_.launch(nshort, f, "pc_short_keys");
		}
		// the fixed point
		for (uint32_t round = 0;; round++) {
			// (tag T_TAGS - 1 is what the cleared array carries: never handed out; cfg_.t_tags < T_TAGS only
			// makes the clearing more frequent -- the tests use that to exercise it)
			if (t_tag_ == 0) {
				// the tags ran out: every stamp back to "never" (a hashed table keeps its keys)
				if (e.T) be_.memset(e.T, 0xFF, m_ * 4ull);
				else be_.memset(e.Tv, 0xFF, 4ull << Tlog2_);
				t_tag_ = std::min<uint32_t>(cfg_.t_tags, T_TAGS) - 1;
			}
			e.tag = --t_tag_;
			if (nshort) {
				be_.memset(e.tcend.hmin, 0xFF, (e.tcend.mask + 1) * 8);
				FPcShort f{ e, 0 };
				be_.launch(1, f, "pc_short");
			}
			be_.memset(e.scal, 0, 4);
			if (round && !part && cfg_.link_duplicates) { FPcSnapshot f{ e }; be_.launch(n, f, "pc_stamp"); }
			{ FPcTimeMin f{ e }; be_.launch_wave(n, f, "pc_timemin"); }
			if (part) {
				be_.memset(part_buf, 1, (uint64_t)n0 + nrec);
				{ FPcDecideA f{ e }; be_.launch_wave(n, f, "pc_decide"); }
				c_all_reduce(part_buf, (uint64_t)n0 + nrec, DT_U8, OP_MIN);
				{ FPcDecideB f{ e }; be_.launch_wave(n, f, "pc_decide"); }
				c_all_reduce(part_c, n, DT_U8, OP_MAX);
				{ FPcDecideC f{ e }; be_.launch(n, f, "pc_decide"); }
			} else { FPcDecide f{ e }; be_.launch_wave(n, f, "pc_decide"); }
			be_.d2h(scal_h, e.scal, 4);
			stats_.commit_rounds++;
			if (!scal_h[0]) break;
		}
		{ FPcBreak f{ e }; be_.launch(n, f, "pc_break"); }
		be_.d2h(scal_h, e.scal, sizeof scal_h);
		const uint32_t brk = std::min(scal_h[1], c_end);
		e.brk = brk;
		if (sliced_) {
			// (FPcApply still sets the visited bits; the counters of a k-mer are spread over the ranks)
			ParCommit ea = e; ea.cnt8 = nullptr;
			{ FPcApply f{ ea }; be_.launch_wave(n, f, "pc_apply"); }
			const uint64_t ext = g_pool_ + 64;
			uint8_t* kmin = (uint8_t*)be_.alloc(ext);
			be_.memset(kmin, 0xFF, ext);
			{ FPcCover f{ e, kmin, 0u }; be_.launch_wave(n, f, "pc_cover"); }
			c_all_reduce(kmin, g_pool_, DT_U8, OP_MIN);
			{ FPcCover f{ e, kmin, 1u }; be_.launch_wave(n, f, "pc_cover"); }
			be_.free(kmin);
		} else { FPcApply f{ e }; be_.launch_wave(n, f, "pc_apply"); }
		if (nshort) { FPcShort f{ e, 1 }; be_.launch(1, f, "pc_short"); }
		// commit order, contig ids, counters
		std::vector<uint64_t> c3(n);
		std::vector<uint8_t> act(n);
		be_.d2h(c1.data(), e.cnt, n * 4ull);
		be_.d2h(c2.data(), e.cnt2, n * 4ull);
		be_.d2h(c3.data(), e.cnt3, n * 8ull);
		be_.d2h(act.data(), e.active, n);
		be_.d2h(scal_h, e.scal, sizeof scal_h);
		if (scal_h[4]) fail_now(FAIL_INTERNAL, "contigEndKmers table overflowed");
		uint32_t orec = 0; uint64_t oid = 0, bases = 0, visited = 0;
		for (uint32_t i = 0; i < n && c_begin + i < brk; i++) {
			uint32_t r = c1[i], q = c2[i];
			c1[i] = orec; c2[i] = (uint32_t)oid;
			orec += r; oid += q; bases += c3[i];
			if (!act[i]) visited++;
		}
		be_.h2d(e.cnt, c1.data(), n * 4ull);
		be_.h2d(e.cnt2, c2.data(), n * 4ull);
		{ FPcWrite f{ e, order_, nord, counters_.contig_id }; be_.launch(n, f, "pc_write"); }
		nord += orec;
		be_.h2d(order_n_, &nord, 4);
		counters_.contig_id += oid;
		counters_.bases_assembled += bases;
		counters_.visited_reads += visited;
		stats_.generated += (brk - c_begin) - visited;
		cend_count_ += scal_h[3];
		if (nshort) { free_tab(e.tcend); be_.free(e.short_keys); }
		be_.free(e.off); be_.free(e.cnt); be_.free(e.cnt2); be_.free(e.cnt3); be_.free(e.active);
		be_.free(e.short_list); be_.free(e.scal);
		if (part_buf) be_.free(part_buf);
		return brk;
	}

	// Partitioned run: each rank walked its share of a launch's candidates and appended records and
	// sequences after the merged ones ([g_rec_, ..) / [g_pool_, ..)).  Exchange them so that every
	// rank holds every record, in rank order, and the per-candidate status / first record of all.
	void merge_walk_results(const uint32_t* list_d, uint32_t nlist, uint32_t* status_d, uint32_t* first_d, uint32_t nc)
	{
		const uint32_t R = (uint32_t)comm_.world, me = (uint32_t)comm_.rank;
		uint32_t lr = 0; uint64_t lp = 0;
		be_.d2h(&lr, rec_used_, 4);
		be_.d2h(&lp, pool_used_, 8);
		lr = std::min(lr, rec_cap_); lp = std::min<uint64_t>(lp, pool_cap_);
		std::vector<uint64_t> cnt(2 * R, 0);
		cnt[2 * me] = lr - g_rec_; cnt[2 * me + 1] = lp - g_pool_;
		host_all_reduce_sum(cnt.data(), 2 * R);
		FRecFix fx;
		fx.rbase[0] = 0; fx.pbase[0] = 0;
		for (uint32_t q = 0; q < R; q++) {
			fx.rbase[q + 1] = fx.rbase[q] + (uint32_t)cnt[2 * q];
			fx.pbase[q + 1] = fx.pbase[q] + cnt[2 * q + 1];
		}
		const uint64_t tot_rec = fx.rbase[R], tot_pool = fx.pbase[R];
		// room for everybody's results (contents are kept: earlier records are still to be committed)
		if (g_rec_ + tot_rec > rec_cap_) {
			uint32_t cap = rec_cap_;
			while (g_rec_ + tot_rec > cap) cap *= 2;
			recs_ = (ContigRec*)regrow(recs_, (uint64_t)rec_cap_ * sizeof(ContigRec), (uint64_t)cap * sizeof(ContigRec));
			order_ = (uint32_t*)regrow(order_, (uint64_t)rec_cap_ * 4, (uint64_t)cap * 4);
			rec_cap_ = cap;
		}
		if (g_pool_ + tot_pool > pool_cap_) {
			uint64_t cap = pool_cap_;
			while (g_pool_ + tot_pool > cap) cap *= 2;
			pool_ = (uint8_t*)regrow(pool_, pool_cap_, cap);
			kh_ = (uint64_t*)regrow(kh_, pool_cap_ * 8, cap * 8);
			pool_cap_ = cap;
		}
		// own block to its place in the merged numbering
		if (me && cnt[2 * me]) {
			void* tmp = be_.alloc(cnt[2 * me] * sizeof(ContigRec));
			be_.d2d(tmp, recs_ + g_rec_, cnt[2 * me] * sizeof(ContigRec));
			be_.d2d(recs_ + g_rec_ + fx.rbase[me], tmp, cnt[2 * me] * sizeof(ContigRec));
			be_.free(tmp);
		}
		if (me && cnt[2 * me + 1]) {
			void* tmp = be_.alloc(cnt[2 * me + 1] * 8);
			be_.d2d(tmp, pool_ + g_pool_, cnt[2 * me + 1]);
			be_.d2d(pool_ + g_pool_ + fx.pbase[me], tmp, cnt[2 * me + 1]);
			be_.d2d(tmp, kh_ + g_pool_, cnt[2 * me + 1] * 8); // (the k-mer hashes of prep_local_records, one per base)
			be_.d2d(kh_ + g_pool_ + fx.pbase[me], tmp, cnt[2 * me + 1] * 8);
			be_.free(tmp);
		}
		std::vector<uint64_t> c(R), d(R);
		for (uint32_t q = 0; q < R; q++) { c[q] = cnt[2 * q] * sizeof(ContigRec); d[q] = (uint64_t)(g_rec_ + fx.rbase[q]) * sizeof(ContigRec); }
		if (tot_rec) c_all_gather_v(recs_, c.data(), d.data());
		for (uint32_t q = 0; q < R; q++) { c[q] = cnt[2 * q + 1]; d[q] = g_pool_ + fx.pbase[q]; }
		if (tot_pool) c_all_gather_v(pool_, c.data(), d.data());
		for (uint32_t q = 0; q < R; q++) { c[q] *= 8; d[q] *= 8; }
		if (tot_pool) c_all_gather_v(kh_, c.data(), d.data());
		if (tot_rec) {
			fx.recs = recs_; fx.g_rec = g_rec_; fx.world = R;
			be_.launch(tot_rec, fx, "merge_fix");
		}
		if (nlist && fx.rbase[me]) { FFirstFix f{ list_d, first_d, fx.rbase[me] }; be_.launch(nlist, f, "merge_fix"); }
		c_all_reduce(status_d, nc, DT_U32, OP_MAX); // WS_NONE < WS_COMPLETE < WS_OVERFLOW; only the walking rank changed an entry
		c_all_reduce(first_d, nc, DT_U32, OP_MIN);  // REC_END unless walked
		g_rec_ += (uint32_t)tot_rec; g_pool_ += tot_pool;
		be_.h2d(rec_used_, &g_rec_, 4);
		be_.h2d(pool_used_, &g_pool_, 8);
	}
	void* regrow(void* old, uint64_t old_bytes, uint64_t new_bytes)
	{
		void* n = be_.alloc(new_bytes);
		be_.d2d(n, old, old_bytes);
		be_.free(old);
		return n;
	}

	// hashes + coverage of the contig records produced since the last call (parallel)
	template <int NW>
	void prep_new_records(uint32_t& prepped)
	{
		uint32_t nrec = 0;
		be_.d2h(&nrec, rec_used_, 4);
		nrec = std::min(nrec, rec_cap_);
		if (nrec > prepped) {
			FContigPrep<NW> f{ p_, cnt_, recs_, prepped, pool_, kh_, use_par_commit() ? 0u : 1u };
			be_.launch_wave(nrec - prepped, f, "contig_prep");
			if (use_par_commit()) link_duplicates(prepped, nrec);
			prepped = nrec;
		}
	}
	// Links every new long record (index >= first_new) to the lowest candidate's record with the same k-mers, if there is one
	// (FDupKeys / FDupLink / FDupVerify): a sort of the records' fingerprints, a look at each group of equal ones, and a comparison
	// of the two hash sequences for every link.  Plain (not partitioned) runs with the parallel commit only.
	void link_duplicates(uint32_t first_new, uint32_t nrec)
	{
		if (!cfg_.link_duplicates || dist() || nrec <= first_new || nrec < 2) return;
		uint64_t* key = (uint64_t*)be_.alloc(nrec * 16ull); uint64_t* key2 = key + nrec;
		uint32_t* val = (uint32_t*)be_.alloc(nrec * 8ull); uint32_t* val2 = val + nrec;
		{ FDupKeys f{ recs_, p_.k, key, val }; be_.launch(nrec, f, "dup_link"); }
		be_.sort_pairs_u64_u32(key, key2, val, val2, nrec);
		{ FDupLink f{ recs_, key2, val2, nrec, first_new }; be_.launch(nrec, f, "dup_link"); }
		{ FDupVerify f{ recs_, kh_, p_.k, first_new }; be_.launch_wave(nrec - first_new, f, "dup_link"); }
		be_.sync(); // (the sort's buffers go back)
		be_.free(key); be_.free(val);
	}
	// partitioned run: the records this rank's walkers appended after the merged ones
	template <int NW>
	void prep_local_records()
	{
		uint32_t lr = 0;
		be_.d2h(&lr, rec_used_, 4);
		lr = std::min(lr, rec_cap_);
		if (lr > g_rec_) {
			FContigPrep<NW> f{ p_, cnt_, recs_, g_rec_, pool_, kh_, use_par_commit() ? 0u : 1u };
			be_.launch_wave(lr - g_rec_, f, "contig_prep");
		}
	}
	void grow_cend()
	{
		uint32_t log2 = 1;
		while ((1ull << log2) < cend_.mask + 1) log2++;
		WalkTab bigger{};
		alloc_tab(bigger, log2 + 2);
		uint32_t* failed = (uint32_t*)be_.alloc(8);
		be_.memset(failed, 0, 4);
		FRehash f{ cend_, bigger, failed };
		be_.launch(cend_.mask + 1, f, "rehash");
		uint32_t bad = 0;
		be_.d2h(&bad, failed, 4);
		be_.free(failed);
		if (bad) fail_now(FAIL_INTERNAL, "contigEndKmers rehash failed");
		free_tab(cend_);
		cend_ = bigger;
	}
	// The vertex table holds every vertex of every walker of one launch.  It is sized ahead for
	// `nwalk` walkers at wtab_per_walker_ entries each (twice that, to keep probing short); a
	// launch that fills it anyway reports overflow and the estimate doubles (run_rounds).
	void ensure_wtab(uint64_t nwalk)
	{
		const uint64_t want = 2 * nwalk * wtab_per_walker_;
		uint32_t log2 = wtab_log2_;
		while ((1ull << log2) < want && log2 < wtab_log2_cap()) log2++;
		if (log2 == wtab_log2_) return;
		free_tab(wtab_);
		wtab_log2_ = log2;
		alloc_tab(wtab_, wtab_log2_);
		if (cfg_.verbose) fprintf(stderr, "abyss_amd: walker vertex table grown to 2^%u entries\n", wtab_log2_);
	}
	// the vertex table takes 24 bytes per entry: at most an eighth of the device's memory
	uint32_t wtab_log2_cap()
	{
		uint32_t cap = cfg_.wtab_log2_max;
		const uint64_t mem = be_.device_mem_bytes();
		while (mem && cap > cfg_.wtab_log2 && (24ull << cap) > mem / 8) cap--;
		return cap;
	}
	void clear_wtab()
	{
		be_.memset(wtab_.hmin, 0xFF, (wtab_.mask + 1) * 8);
		be_.memset(wtab_.meta, 0xFF, (wtab_.mask + 1) * 8);
	}

	// ---- one batch of reads in flight -----------------------------------------------------
	// Every candidate that is not entirely visited by now is walked; then the ordered commit runs as far as the results
	// allow.  Walk results are pure functions of the read and the solid filter, so they stay valid across iterations.
	struct BatchRun {
		Batch v{};                  // the batch's reads
		uint64_t first = 0, n = 0;  // ... which are reads [first, first + n) of the call
		uint8_t* res_d = nullptr;   // their verdicts
		std::vector<uint32_t> cand_h;
		uint32_t nc = 0;
		uint32_t* cand_d = nullptr; uint32_t* status_d = nullptr; uint32_t* first_d = nullptr;
		uint32_t* need_d = nullptr; uint32_t* need_n = nullptr; uint64_t* rkoff_d = nullptr;
		uint32_t base = 0;          // candidates [0, base) are accounted for
		uint32_t prepped = 0, owner_next = 0, committed = 0, force = 0xFFFFFFFFu, nneed = 0, nneed_all = 0;
		bool round_started = false; // the prologue of a round (see start_round) ran for `base`
		bool predicted = false;     // ... including the first prediction
		bool pending = false;       // ... and its walkers are running (or done) but not yet accounted for
		bool overflow = false, debug = false;
	};
	template <int NW>
	void assemble_nw(const Batch& b, uint8_t* result_d, uint8_t* results_host,
	    const std::function<void(const ContigOut&)>& sink)
	{
		uint64_t next = 0; // first read not yet classified
		while (next < b.n) {
			ensure_walk();
			BatchRun r;
			classify_batch<NW>(b, next, std::min<uint64_t>(p2_batch_, b.n - next), result_d, r);
			next += r.n;
			counters_.reads_processed += r.n;
			p2_batch_ = next_batch_size();
			prefetch_ = nullptr;
			if (next < b.n && !dist() && cfg_.prefetch_classify) {
				// Classification of the NEXT batch on the side stream, queued right before this batch's
				// walkers so that it fills the machine while they thin out (a launch ends with its
				// slowest walker).  It sees an older snapshot; FRefilter brings it up to date.
				const uint64_t nf = next, nn = std::min<uint64_t>(p2_batch_, b.n - next);
				prefetch_ = [this, &b, result_d, nf, nn]() {
					if (!la_pool_c2_) la_pool_c2_ = (VKey*)be_.alloc((uint64_t)cslots_ * LA_MAX_VISITED * sizeof(VKey));
					Batch vn = b;
					vn.woff = b.woff + nf; vn.len = b.len + nf; vn.koff = b.koff + nf; vn.n = nn;
					FClassify<NW> f{ p2_, vn, 0, cnt2_, vis_, result_d + nf, la_pool_c2_, both_, arc_, cfg_.cls_debug_skip };
					be_.launch_slots_side(nn, f, cslots_, "classify");
					pre_first_ = nf; pre_n_ = nn;
				};
			}
			if (r.nc) {
				setup_batch<NW>(r);
				finish_batch<NW>(r, sink);
			}
			if (results_host) {
				be_.d2h(results_host + r.first, r.res_d, r.n);
				if (const void* left = memchr(results_host + r.first, RES_CANDIDATE, r.n)) {
					fail_now(FAIL_INTERNAL, strf("read %llu left unprocessed", (unsigned long long)((const uint8_t*)left - results_host)));
				}
			}
		}
	}

	// Verdicts of reads [first, first + n) against the current visited snapshot, and the batch they
	// make: the longest prefix holding at most cfg_.p2_max_candidates candidates (what the walkers'
	// tables and the commit's positions are sized for); the rest is classified again later.
	template <int NW>
	void classify_batch(const Batch& b, uint64_t first, uint64_t n, uint8_t* result_d, BatchRun& r)
	{
		Batch v = b;
		v.woff = b.woff + first; v.len = b.len + first; v.koff = b.koff + first; v.n = n;
		uint8_t* res_d = result_d + first;
		if (!la_pool_c_) la_pool_c_ = (VKey*)be_.alloc((uint64_t)cslots_ * LA_MAX_VISITED * sizeof(VKey));
		if (pre_n_ == n && pre_first_ == first) {
			// classified ahead against an older snapshot (prefetch_classify): BLUNT_END / NOT_SOLID do not
			// depend on the snapshot and "all k-mers visited" is final once true, so only the candidates
			// are tested again
			be_.sync_side();
			FRefilter<NW> f{ p_, v, vis_, res_d };
			be_.launch(n, f, "reclassify");
		} else if (dist()) {
			// every rank classifies a slice of the batch; the verdicts are gathered
			const uint64_t R = (uint64_t)comm_.world;
			std::vector<uint64_t> c(R), d(R);
			for (uint64_t q = 0; q < R; q++) { d[q] = n * q / R; c[q] = n * (q + 1) / R - d[q]; }
			FClassify<NW> f{ p2_, v, d[comm_.rank], cnt2_, vis_, res_d, la_pool_c_ };
			be_.launch_slots(c[comm_.rank], f, cslots_, "classify");
			c_all_gather_v(res_d, c.data(), d.data());
		} else {
			be_.sync_side();
			FClassify<NW> f{ p2_, v, 0, cnt2_, vis_, res_d, la_pool_c_, both_, arc_, cfg_.cls_debug_skip };
			be_.launch_slots(n, f, cslots_, "classify");
		}
		pre_n_ = 0;
		std::vector<uint8_t> res(n);
		be_.d2h(res.data(), res_d, n);
		r = BatchRun();
		uint64_t used = n;
		// (no more candidates in a batch than the vertex table at its largest plans room for)
		const uint64_t max_cand = std::min<uint64_t>(cfg_.p2_max_candidates,
		    std::max<uint64_t>(1024, (1ull << wtab_log2_cap()) / (2 * wtab_per_walker_)));
		// (the candidates are few among millions of verdicts late in a read set: found with memchr, the
		// other verdicts counted in a loop the compiler vectorises)
		for (const uint8_t* q = res.data(), *end = q + n; (q = (const uint8_t*)memchr(q, RES_CANDIDATE, (size_t)(end - q))) != nullptr; q++) {
			if (r.cand_h.size() >= max_cand) { used = (uint64_t)(q - res.data()); break; }
			r.cand_h.push_back((uint32_t)(q - res.data()));
		}
		uint64_t n_visited = 0;
		for (uint64_t i = 0; i < used; i++) n_visited += res[i] == RR_ALL_KMERS_VISITED;
		counters_.solid_reads += n_visited + r.cand_h.size();
		counters_.visited_reads += n_visited;
		v.n = used;
		r.v = v; r.first = first; r.n = used; r.res_d = res_d;
		r.nc = (uint32_t)r.cand_h.size();
		if (used < n) stats_.batch_cuts++;
		stats_.candidates += r.nc;
		last_candidates_ = r.nc;
	}

	// per-batch arrays and the hashes of the candidates' read k-mers (predictor, commit)
	template <int NW>
	void setup_batch(BatchRun& r)
	{
		const uint32_t nc = r.nc;
		const Batch& b = r.v;
		r.cand_d = (uint32_t*)be_.alloc(nc * 4ull);
		r.status_d = (uint32_t*)be_.alloc(nc * 4ull);
		r.first_d = (uint32_t*)be_.alloc(nc * 4ull);
		r.need_d = (uint32_t*)be_.alloc(nc * 4ull);
		r.need_n = (uint32_t*)be_.alloc(8);
		be_.h2d(r.cand_d, r.cand_h.data(), nc * 4ull);
		// (the candidates' lengths only: the batch may hold millions of reads)
		std::vector<uint32_t> len_h(nc);
		{
			uint32_t* cl = (uint32_t*)be_.alloc(nc * 4ull + 4);
			FGatherU32 f{ b.len, r.cand_d, cl };
			be_.launch(nc, f, "gather_len");
			be_.d2h(len_h.data(), cl, nc * 4ull);
			be_.free(cl);
		}
		std::vector<uint64_t> rkoff(nc + 1, 0);
		for (uint32_t i = 0; i < nc; i++) rkoff[i + 1] = rkoff[i] + (len_h[i] - p_.k + 1);
		r.rkoff_d = (uint64_t*)be_.alloc((nc + 1) * 8ull);
		be_.h2d(r.rkoff_d, rkoff.data(), (nc + 1) * 8ull);
		rkh_ = (uint64_t*)be_.alloc(std::max<uint64_t>(rkoff[nc], 1) * 8);
		read_flag_ = (uint8_t*)be_.alloc(nc);
		{
			FReadPrep<NW> f{ p_, b, r.cand_d, r.rkoff_d, rkh_, 0 };
			be_.launch_wave(nc, f, "read_prep");
		}
		r.debug = getenv("ABG_WALK_DEBUG") != nullptr;
		if (r.debug) { dbg_ = (uint64_t*)be_.alloc(nc * 8ull * WALK_DBG_N); be_.memset(dbg_, 0, nc * 8ull * WALK_DBG_N); }
		r.base = 0; r.round_started = false;
	}
	void dump_walkers(BatchRun& r, const char* what, uint32_t nwalk)
	{
		if (!r.debug) return;
		const uint32_t nc = r.nc;
		std::vector<uint64_t> d(nc * (uint64_t)WALK_DBG_N);
		be_.d2h(d.data(), dbg_, nc * 8ull * WALK_DBG_N);
		uint64_t best = 0, bi = 0, nn = 0, sum[WALK_DBG_N] = { 0 };
		for (uint32_t i = 0; i < nc; i++) {
			const uint64_t* x = &d[i * (uint64_t)WALK_DBG_N];
			if (!x[0]) continue;
			nn++;
			for (int q = 0; q < (int)WALK_DBG_N; q++) sum[q] += x[q];
			if (x[0] > best) { best = x[0]; bi = i; }
		}
		auto line = [&](const char* tag, const uint64_t* x) {
			fprintf(stderr, "[walkdbg]   %s: t=%.2fms (search %.2f [chains %.2f] in %llu calls, %llu tbnodes; linear %.2f of which bulk %.2f in %llu tries / %llu hits / %llu steps; post %.2f) steps=%llu contigs=%llu\n",
			    tag, x[0] / 1e5, x[2] / 1e5, x[12] / 1e5, (unsigned long long)x[3], (unsigned long long)x[4], x[8] / 1e5, x[5] / 1e5,
			    (unsigned long long)x[9], (unsigned long long)x[10], (unsigned long long)x[11], x[7] / 1e5,
			    (unsigned long long)x[1], (unsigned long long)x[6]);
		};
		fprintf(stderr, "[walkdbg] %s n=%u ran=%llu\n", what, nwalk, (unsigned long long)nn);
		line("sum", sum);
		fprintf(stderr, "[walkdbg]   lookAhead inside trueBranch: %.2f ms in %llu calls; bulk examine phase %.2f ms; trueBranch waiting for neighbour masks %.2f ms in %llu probe rounds; call entries %.2f ms\n", sum[13] / 1e5, (unsigned long long)sum[14], sum[15] / 1e5,
		    sum[16] / 1e5, (unsigned long long)sum[17], sum[18] / 1e5);
		fprintf(stderr, "[walkdbg]   bulk examine: strand hashes of the chunk %.2f ms; k-mer, identity, home slot %.2f; eight first probes %.2f; the survivors' other probes %.2f; table and checks %.2f; barrier %.2f\n",
		    sum[22] / 1e5, sum[23] / 1e5, sum[24] / 1e5, sum[25] / 1e5, sum[26] / 1e5, sum[27] / 1e5);
		fprintf(stderr, "[walkdbg]   the read's k-mers looked up one by one %.2f ms; inside walk_extend %.2f ms (searches and linear runs included)\n", sum[19] / 1e5, sum[20] / 1e5);
		{
			const uint64_t* x = &d[bi * (uint64_t)WALK_DBG_N];
			fprintf(stderr, "[walkdbg]   slowest walker's lookAhead: %.2f ms in %llu calls; its trueBranch: %.2f ms waiting for neighbour masks (%llu probe rounds), %.2f ms in call entries (identity, on-stack scan, frame push)\n", x[13] / 1e5, (unsigned long long)x[14],
			    x[16] / 1e5, (unsigned long long)x[17], x[18] / 1e5);
		}
		line("slowest", &d[bi * (uint64_t)WALK_DBG_N]);
		be_.memset(dbg_, 0, nc * 8ull * WALK_DBG_N);
	}

	// Prologue of a round over the candidates [r.base, nc) -- nothing of them walked yet -- up to and
	// including the launch of the round's first walkers.
	template <int NW>
	void start_round(BatchRun& r)
	{
		const uint32_t nc = r.nc, base = r.base;
		stats_.rounds++;
		be_.memset(r.status_d + base, 0, (nc - base) * 4ull);
		be_.memset(r.first_d + base, 0xFF, (nc - base) * 4ull);
		be_.memset(pool_used_, 0, 8);
		be_.memset(rec_used_, 0, 4);
		be_.memset(order_n_, 0, 4);
		g_rec_ = 0; g_pool_ = 0;
		r.prepped = 0; r.owner_next = 0;
		prep_new_records<NW>(r.prepped);
		r.committed = base; r.force = 0xFFFFFFFFu; r.overflow = false;
		r.round_started = true;
		predict_and_walk<NW>(r);
	}
	// the candidates without a result whose reads are not entirely visited by now are walked
	template <int NW>
	void predict_and_walk(BatchRun& r)
	{
		const uint32_t nc = r.nc;
		be_.memset(r.need_n, 0, 8);
		{
			FPredict<NW> fp{ p_, r.v, r.cand_d, r.status_d, vis_, r.rkoff_d, rkh_,
				r.need_d, r.need_n, r.committed, r.force, (uint32_t)comm_.rank, (uint32_t)comm_.world };
			be_.launch(nc - r.committed, fp, "predict");
		}
		// partitioned run: nneed = what this rank walks (every rank's c % world == rank share of the
		// needed candidates), nneed_all = what all ranks walk together
		uint32_t nn2[2] = { 0, 0 };
		be_.d2h(nn2, r.need_n, 8);
		r.nneed = nn2[0];
		r.nneed_all = comm_.world > 1 ? nn2[1] : r.nneed; // (FPredict counts need_n[1] only when it filters)
		r.predicted = true; r.pending = false;
		if (!r.nneed_all) return;
		if (r.owner_next > 0xF0000000u - nc) { r.overflow = true; return; } // owner ids exhausted: restart
		ensure_wtab(std::max<uint32_t>(r.nneed, 1));
		clear_wtab();
		WalkEnv<NW> env = make_env<NW>(r.v, r.cand_d, r.status_d, r.first_d);
		env.owner_base = r.owner_next;
		r.owner_next += nc;
		FWalk<NW> fw{ env, r.need_d };
		if (prefetch_) { prefetch_(); prefetch_ = nullptr; }
		be_.launch_walkers(r.nneed, fw, wslots_, "rewalk");
		r.pending = true;
	}

	template <int NW>
	void finish_batch(BatchRun& r, const std::function<void(const ContigOut&)>& sink)
	{
		const uint32_t nc = r.nc;
		while (r.base < nc) {
			if (!r.round_started) start_round<NW>(r);
			while (r.committed < nc && !r.overflow) {
				if (!r.predicted) predict_and_walk<NW>(r);
				r.predicted = false;
				if (r.overflow) break;
				if (r.pending) {
					r.pending = false;
					stats_.rewalked += r.nneed_all;
					dump_walkers(r, "rewalk", r.nneed);
					if (dist()) {
						// (a rank prepares the records it walked; their hashes travel with the sequences)
						prep_local_records<NW>();
						merge_walk_results(r.need_d, r.nneed, r.status_d, r.first_d, nc);
						r.prepped = g_rec_;
					} else prep_new_records<NW>(r.prepped);
				}
				// stage 3: ordered commit as far as the results allow
				uint32_t next = use_par_commit() ? commit_par<NW>(r.v, r.cand_d, r.status_d, r.first_d, r.res_d, r.rkoff_d, r.committed, nc)
				                                 : commit<NW>(r.v, r.cand_d, r.status_d, r.first_d, r.res_d, r.rkoff_d, r.committed, nc);
				if (next < nc) {
					stats_.breaks++;
					uint32_t st = 0;
					be_.d2h(&st, r.status_d + next, 4);
					if (st == WS_OVERFLOW) { r.committed = next; r.overflow = true; break; }
					if (next == r.committed && (st == WS_COMPLETE || r.force == next)) {
						fail_now(FAIL_INTERNAL, strf("commit made no progress at candidate %llu (status %llu)", next, st));
					}
					r.force = next; // needed after all: walk it in the next iteration
				}
				r.committed = next;
			}
			deliver(r.cand_h, r.first, sink);
			if (r.overflow) {
				// the candidate at `committed` ran out of some capacity.  Results not yet committed
				// are dropped and the walk restarts from there; if nothing was committed in this
				// attempt the capacities themselves are too small for that read.
				stats_.overflows++;
				// what ran out?  This context's own allocation counters say whether it was its contig pool or its
				// records (a walker that finds no room leaves the counter past the capacity; the shared
				// WSTAT_OVF_* counters are statistics only: with several batches in flight another context's
				// walkers bump them too); anything else is the vertex table (plan for longer walks from now on)
				// or, when not even the first candidate got through, a walker's own stack or path buffer.
				uint64_t pool_now = 0; uint32_t recs_now = 0;
				be_.d2h(&pool_now, pool_used_, 8);
				be_.d2h(&recs_now, rec_used_, 4);
				bool pool_out = pool_now > pool_cap_, recs_out = recs_now > rec_cap_;
				if (dist()) {
					// (a partitioned run has merged the ranks' records by now and set the counters to the merged sizes;
					// it runs one batch at a time, so the shared counters are this batch's)
					uint64_t ovf[2] = { 0, 0 };
					be_.d2h(ovf, wstats_ + WSTAT_OVF_POOL, 16);
					pool_out = pool_out || ovf[0] != ovf_seen_[0]; recs_out = recs_out || ovf[1] != ovf_seen_[1];
					ovf_seen_[0] = ovf[0]; ovf_seen_[1] = ovf[1];
				}
				if (pool_out) {
					be_.free(pool_); be_.free(kh_);
					pool_cap_ *= 2;
					pool_ = (uint8_t*)be_.alloc(pool_cap_);
					kh_ = (uint64_t*)be_.alloc(pool_cap_ * 8);
					if (cfg_.verbose) fprintf(stderr, "abyss_amd: contig pool grown to %llu bases\n", (unsigned long long)pool_cap_);
				}
				if (recs_out) {
					be_.free(recs_); be_.free(order_);
					rec_cap_ *= 2;
					recs_ = (ContigRec*)be_.alloc((uint64_t)rec_cap_ * sizeof(ContigRec));
					order_ = (uint32_t*)be_.alloc((uint64_t)rec_cap_ * 4);
					if (cfg_.verbose) fprintf(stderr, "abyss_amd: contig records grown to %u\n", rec_cap_);
				}
				if (!pool_out && !recs_out) {
					if (r.committed == r.base) grow_walk_resources();
					else wtab_per_walker_ *= 2;
				}
			}
			r.base = r.committed;
			r.round_started = false; r.predicted = false; r.pending = false;
		}
		be_.free(rkh_); rkh_ = nullptr;
		be_.free(read_flag_); read_flag_ = nullptr;
		if (r.debug) { be_.free(dbg_); dbg_ = nullptr; }
		be_.free(r.cand_d); be_.free(r.status_d); be_.free(r.first_d);
		be_.free(r.need_d); be_.free(r.need_n); be_.free(r.rkoff_d);
		r.cand_h.clear();
	}

	void deliver(const std::vector<uint32_t>& cand_h, uint64_t read_base,
	    const std::function<void(const ContigOut&)>& sink)
	{
		if (!sink) return; // nobody wants the records: counters already reflect them
		uint32_t n = 0;
		be_.d2h(&n, order_n_, 4);
		if (!n) return;
		std::vector<uint32_t> order(n);
		be_.d2h(order.data(), order_, n * 4ull);
		uint32_t nrec = 0;
		be_.d2h(&nrec, rec_used_, 4);
		nrec = std::min(nrec, rec_cap_);
		std::vector<ContigRec> recs(nrec);
		be_.d2h(recs.data(), recs_, nrec * sizeof(ContigRec));
		uint64_t used = 0;
		be_.d2h(&used, pool_used_, 8);
		used = std::min<uint64_t>(used, pool_cap_);
		std::vector<uint8_t> pool(used);
		be_.d2h(pool.data(), pool_, used);
		// The copies are off the device; turning them into strings and handing them to the caller goes on
		// beside the next batch's device work, on a thread of its own -- one delivery at a time, in order;
		// assemble_packed waits for the last one before it returns.
		deliveries_wait();
		const std::function<void(const ContigOut&)>* sk = &sink;
		delivery_ = std::async(std::launch::async,
		    [sk, read_base, n, order = std::move(order), recs = std::move(recs), pool = std::move(pool), cand = cand_h]() {
			ContigOut o;
			for (uint32_t i = 0; i < n; i++) {
				const ContigRec& r = recs[order[i]];
				o.contig_id = r.redundant ? ~0ULL : r.contig_id;
				o.read_index = read_base + cand[r.cand];
				o.seq.resize(r.len);
				for (uint32_t j = 0; j < r.len; j++) o.seq[j] = "ACGTN"[pool[r.seq_off + j] <= 4 ? pool[r.seq_off + j] : 4];
				o.coverage = r.redundant ? 0 : r.coverage;
				o.redundant = r.redundant != 0;
				o.left_ext = r.left_ext; o.right_ext = r.right_ext;
				o.left_code = r.left_code; o.right_code = r.right_code;
				o.seed_pos = r.seed_pos;
				(*sk)(o);
			}
		});
	}
	std::future<void> delivery_;
	void deliveries_wait() { if (delivery_.valid()) delivery_.get(); }

};

} // namespace abg
