// abg_walk.h -- PASS 2 of abyss-bloom-dbg on the device: the per-read unitig walk
// (processRead, BloomDBG/bloom-dbg.h:781-882) as a pure function of the read and the
// read-only solid filter, and the per-k-mer pieces of the ordered commit
// (outputContig, bloom-dbg.h:538-620).
//
// A "walker" owns one candidate read.  Its vertex sets (extendPath's `visited`,
// processRead's `assembledKmers`) live in one device-wide open-addressing table keyed
// by (k-mer identity, owner), so walkers never see each other's entries.
#pragma once
#include "abg_core.h"

namespace abg {

// ---------------------------------------------------------- packed read batch
// Sequences are pure ACGT, 2 bits per base, 16 bases per 32-bit word, each sequence
// starting on a word boundary.
struct Batch {
	const uint32_t* words;
	const uint64_t* woff;   // [n + 1] word offset of each sequence
	const uint32_t* len;    // [n] length in bases
	const uint64_t* koff;   // [n + 1] prefix sum of k-mer counts (len - k + 1)
	uint64_t n;
};
ABG_HD unsigned batch_base(const Batch& b, uint64_t r, uint32_t i)
{
	uint32_t w = b.words[b.woff[r] + (i >> 4)];
	return (w >> (2 * (i & 15))) & 3u;
}
template <int NW>
ABG_HD Kmer<NW> batch_kmer(const Batch& b, uint64_t r, uint32_t pos, unsigned k)
{
	return window_kmer<NW>(b.words, b.woff[r], pos, k); // (a handful of word loads in flight together, not a load per base)
}
// index of the sequence holding k-mer op t (koff[r] <= t < koff[r+1])
ABG_HD uint64_t find_seq(const uint64_t* koff, uint64_t n, uint64_t t)
{
	uint64_t lo = 0, hi = n; // invariant: koff[lo] <= t < koff[hi]
	while (hi - lo > 1) {
		uint64_t mid = (lo + hi) >> 1;
		if (koff[mid] <= t) lo = mid; else hi = mid;
	}
	return lo;
}

// ---------------------------------------------------------------- vertex table
constexpr uint64_t WT_EMPTY = ~0ULL;
constexpr uint32_t WT_TOMB = 0xFFFFFFFEu;      // contig field of a tombstoned entry
struct WalkTab {
	uint64_t* hmin;   // [cap]  WT_EMPTY when free
	uint64_t* hmax;   // [cap]
	uint64_t* meta;   // [cap]  owner << 32 | contig
	uint64_t mask;    // cap - 1 (cap is a power of two)
};
// table key of a vertex: its identity (vtx_ident), kept clear of the "free slot" value
ABG_HD VKey wt_key(VKey key)
{
	if (key.fh == WT_EMPTY) key.fh = WT_EMPTY - 1;
	return key;
}
template <int NW>
ABG_HD VKey vtx_key(const Params& p, const Vtx<NW>& v) { return wt_key(vtx_ident(p, v)); }
ABG_HD uint64_t wt_slot(const WalkTab& t, const VKey& key, uint32_t owner)
{
	uint64_t x = key.fh ^ ((uint64_t)(owner + 1) * 0xD6E8FEB86659FD93ULL);
	x ^= x >> 32; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 29;
	return x & t.mask;
}
enum { WT_NEW = 0, WT_SAME_CONTIG = 1, WT_EARLIER = 2, WT_FULL = 3 };
// insert (key, owner) with contig number; returns WT_NEW (also when reviving a
// tombstone), WT_SAME_CONTIG (already inserted by this contig walk: a cycle),
// WT_EARLIER (inserted by an earlier contig of the same read; now re-tagged) or WT_FULL.
ABG_HD int wt_insert(const WalkTab& t, const VKey& key, uint32_t owner, uint32_t contig, bool coop = false)
{
	uint64_t s = wt_slot(t, key, owner);
	for (uint64_t probes = 0; probes <= t.mask; probes++, s = (s + 1) & t.mask) {
		// optimistic: most insertions find their home slot free, so try to take it first
		// (one round trip) instead of reading it and then taking it (two)
		uint64_t cur = wu_cas_u64(&t.hmin[s], WT_EMPTY, key.fh, coop);
		if (cur == WT_EMPTY) {
			wu_st_coherent(&t.hmax[s], key.rh, coop);
			wu_st_coherent(&t.meta[s], ((uint64_t)owner << 32) | contig, coop);
			return WT_NEW;
		}
		if (cur != key.fh) continue;
		uint64_t m = ld_coherent(&t.meta[s]);
		if ((uint32_t)(m >> 32) != owner) continue;
		if (ld_coherent(&t.hmax[s]) != key.rh) continue;
		uint32_t c = (uint32_t)m;
		if (c == contig) return WT_SAME_CONTIG;
		wu_st_coherent(&t.meta[s], ((uint64_t)owner << 32) | contig, coop);
		return c == WT_TOMB ? WT_NEW : WT_EARLIER;
	}
	return WT_FULL;
}
// returns the slot of (key, owner) or WT_EMPTY
ABG_HD uint64_t wt_find(const WalkTab& t, const VKey& key, uint32_t owner)
{
	uint64_t s = wt_slot(t, key, owner);
	for (uint64_t probes = 0; probes <= t.mask; probes++, s = (s + 1) & t.mask) {
		uint64_t cur = ld_coherent(&t.hmin[s]);
		if (cur == WT_EMPTY) return WT_EMPTY;
		if (cur != key.fh) continue;
		uint64_t m = ld_coherent(&t.meta[s]);
		if ((uint32_t)(m >> 32) != owner) continue;
		if (ld_coherent(&t.hmax[s]) != key.rh) continue;
		return s;
	}
	return WT_EMPTY;
}


// ------------------------------------------------------------- walker output
enum WalkStatus : uint32_t {
	WS_NONE = 0,      // not walked yet
	WS_COMPLETE = 1,  // all contigs of the read recorded
	WS_OVERFLOW = 3   // a capacity (stack, path buffer, pool, table, records) was exceeded
};
struct ContigRec {
	uint64_t seq_off;     // offset of the sequence in the contig pool (1 byte per base, 0..3)
	uint32_t len;         // bases
	uint32_t cand;        // candidate that produced it
	uint32_t next;        // next record of the same candidate (UINT32_MAX = end)
	uint32_t seed_pos;    // read k-mer index that seeded the walk
	uint32_t left_ext, right_ext;
	uint8_t left_code, right_code;
	uint8_t redundant;    // filled by the commit
	uint8_t pre_redundant; // settled ahead of the commit: every k-mer already visited
	uint32_t coverage;    // filled by the commit
	uint64_t contig_id;   // filled by the commit
	uint32_t time;        // parallel commit: position of this contig in the commit order of its range
	uint32_t ins;         // parallel commit: the contig is (assumed to be) inserted into the visited set
	uint32_t ins_prev;    // ... as the pass before left it (FPcSnapshot: what the copies of a contig look at)
	uint32_t dup_of;      // a record of a LOWER candidate holding exactly this contig's k-mers (REC_END: none; Engine::link_duplicates)
	uint64_t fp;          // sum of mixed k-mer hashes: equal for equal k-mer multisets (FContigPrep)
};
constexpr uint32_t REC_END = 0xFFFFFFFFu;

template <int NW>
struct WalkEnv {
	Params p;
	const uint8_t* cnt;       // solid filter (read-only in pass 2)
	Batch batch;
	const uint32_t* cand_read; // [ncand] read index of each candidate (ascending)
	uint32_t* status;          // [ncand] WalkStatus
	uint32_t* first_rec;       // [ncand]
	WalkTab tab;
	uint32_t owner_base;       // owner ids of this launch are owner_base + candidate index
	// per-slot scratch
	TBFrame<NW>* tb_pool; VKey* tbk_pool; uint32_t tb_cap;
	// fast memory private to the walker (LDS on the device): scratch, path state, trueBranch stack
	void* fast; uint32_t fast_bytes;
	VKey* la_pool;
	uint8_t* lbuf_pool; uint8_t* rbuf_pool; uint32_t buf_cap;
	uint64_t* dbg;             // optional [ncand][16] per-walker work counters (profiling aid)
	bool coop;                 // the walker is a whole wavefront in lock step
	Guide guide;               // read-guided bulk steps (tab == NULL: off)
	SuccMemo memo;             // shared answers of successor() (k0 == NULL: off)
	MaskCache* mcache;         // neighbour masks kept in the walker's fast memory (NULL: none)
	BulkScratch* bulk_pool;    // [slots] scratch of the bulk steps when the fast memory has no room for it
	uint64_t* wstats;          // [WSTAT_N] work counters summed over all walkers
	// contig output
	uint8_t* pool; uint64_t pool_cap; uint64_t* pool_used;
	ContigRec* recs; uint32_t rec_cap; uint32_t* rec_used;
};

// The path of one contig walk: reverse(lbuf[0..nl)) + seed + rbuf[0..nr), materialised
// into the pool (one slack base either side) once both extensions are done.
template <int NW>
struct WalkState {
	Vtx<NW> seed;
	uint8_t* lbuf; uint8_t* rbuf;
	uint32_t nl, nr;
	// hand-over between walk_extend and walk_linear
	Vtx<NW> head;     // last vertex of the path in the walking direction
	VKey prev_key;    // identity of the vertex before it
	uint32_t ext;     // vertices appended by this extendPath call so far
	int32_t ins;      // LIN_INS: what the insertion of `head` into the visited set returned
	// read-guided bulk steps (walk_bulk)
	BulkScratch* bulk;     // NULL: off
	uint32_t bulk_skip;    // the head is known not to start a bulk step (the last one stopped at it)
	uint32_t bulk_overflow; // the vertex table filled up during a bulk step
	uint32_t n_bulk_calls, n_bulk_steps, n_lin_steps; // work counters of this walker
	uint32_t n_bulk_tries;
	uint64_t t_bx[6];               // ... the examine phase of walk_bulk piece by piece
	uint64_t t_seed, t_ext, t_mat;  // ... the read's k-mers looked up one after the other / inside walk_extend / the path written out
	uint64_t t_bulk, t_lin, t_post; // profiling aid (ABG_WALK_DEBUG): clock ticks in walk_bulk / walk_linear / after the extensions
	uint64_t t_bp[4];               // ... and inside walk_bulk: verify the hint / examine the vertices / repeats / take the steps
};
template <int NW>
ABG_HD unsigned ws_base(const Params& p, const WalkState<NW>& w, uint32_t j)
{
	if (j < w.nl) return w.lbuf[w.nl - 1 - j];
	j -= w.nl;
	if (j < p.k) return kmer_get(w.seed.s, j);
	return w.rbuf[j - p.k];
}
template <int NW>
ABG_HDN Vtx<NW> ws_vertex(const Params& p, const WalkState<NW>& w, uint32_t i, bool coop = false)
{
	return gather_vertex<NW>(p, coop, [&](unsigned j) { return ws_base(p, w, i + j); });
}
template <int NW>
ABG_HDN Vtx<NW> pool_vertex(const Params& p, const uint8_t* seq, uint64_t i, bool coop = false)
{
	return gather_vertex<NW>(p, coop, [&](unsigned j) { return (unsigned)seq[i + j] & 3u; });
}

// profiling aid (ABG_WALK_DEBUG): the 100 MHz wall clock, only read when a debug buffer is attached
ABG_HD uint64_t dbg_clock(const uint64_t* dbg)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return dbg ? wall_clock64() : 0;
#else
	(void)dbg; return 0;
#endif
}
// One bulk step of walk_linear.  Entered like its loop: w.head is pushed but not yet entered into
// the visited set; `hint` is the guide entry of its canonical hash.  If the hint's read really
// holds the head, its following k-mers (up to the end of the read, at most 64) are examined one
// per lane exactly as walk_linear examines one head per iteration -- visited.insert(v) would be
// new (ExtendPath.h:650-658), v has one neighbour either side and the one behind is the previous
// vertex (extendPathBySingleVertex at successor()'s level 0, ExtendPath.h:314-362,403-459) -- plus
// "the one ahead is the read's next k-mer", and the longest prefix of vertices that all pass is
// taken: entered into the visited set, their bases appended, the head moved on.  Returns the
// number of steps taken (0: nothing done, nothing changed).  A vertex that repeats an earlier one
// of the same chunk ends the prefix (the step-by-step code then sees the cycle).
template <int NW, bool COOP>
ABG_HDX uint32_t walk_bulk(const WalkEnv<NW>& e, WalkState<NW>& w, const int dir_in, const uint32_t owner_in,
    const uint32_t contig_in, const uint64_t hint_in)
{
	{
	const uint64_t tq0 = dbg_clock(e.dbg);
	const Params p = uniform_params<COOP>(e.p);
	const unsigned k = p.k;
	const uint8_t* __restrict__ cnt = uniptr<COOP>(e.cnt);
	const uint32_t* __restrict__ gwords = uniptr<COOP>(e.guide.words);
	const uint64_t gnwords = uni64<COOP>(e.guide.nwords);
	uint8_t* const seen = uniptr<COOP>(e.guide.seen);
	WalkTab tab;
	tab.hmin = uniptr<COOP>(e.tab.hmin); tab.hmax = uniptr<COOP>(e.tab.hmax); tab.meta = uniptr<COOP>(e.tab.meta);
	tab.mask = uni64<COOP>(e.tab.mask);
	const uint64_t hint = uni64<COOP>(hint_in);
	const uint64_t woff = hint & GUIDE_MAX_WOFF;
	const uint32_t pos = (uint32_t)(hint >> 39) & 0xFFu, nk = ((uint32_t)(hint >> 47) & 0xFFu) + 1u;
	if (pos >= nk || woff + ((nk + k - 1 + 15) >> 4) > gnwords) return 0; // (a stale or mangled hint)
	const int dir = (int)uni32<COOP>((uint32_t)dir_in);
	const uint32_t owner = uni32<COOP>(owner_in), contig = uni32<COOP>(contig_in);
	const int fsense = (dir == FORWARD) ? SENSE : ANTISENSE, bsense = (dir == FORWARD) ? ANTISENSE : SENSE;
	Vtx<NW> head;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) head.s.w[j] = uni64<COOP>(w.head.s.w[j]);
	// does the read hold the head at `pos`, and on which strand?
	bool same = true, anti = true;
	{
		const Kmer<NW> rk = window_kmer<NW>(gwords, woff, pos, k);
		const Kmer<NW> hr = kmer_revcomp_fast(head.s, k);
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) {
			const uint64_t x = uni64<COOP>(rk.w[j]);
			same = same & (x == head.s.w[j]);
			anti = anti & (x == hr.w[j]);
		}
	}
	if (!same && !anti) return 0;
	const bool up = (fsense == SENSE) == same; // the walk runs towards higher k-mer indices of the read
	uint8_t* buf = uniptr<COOP>(dir == FORWARD ? w.rbuf : w.lbuf);
	const uint32_t nbuf = uni32<COOP>(dir == FORWARD ? w.nr : w.nl), buf_cap = uni32<COOP>(e.buf_cap);
	uint32_t n = up ? nk - pos : pos + 1u;
	if (n > BULK_LANES) n = BULK_LANES;
	if (nbuf >= buf_cap) return 0;
	if (n > buf_cap - nbuf) n = buf_cap - nbuf;
	if (n < BULK_MIN) return 0;
	BulkScratch& bs = *uniptr<COOP>(w.bulk);
	VKey prev_key;
	prev_key.fh = uni64<COOP>(w.prev_key.fh); prev_key.rh = uni64<COOP>(w.prev_key.rh);
	const uint64_t sk0 = p.seed_k[0], sk1 = p.seed_k[1], sk2 = p.seed_k[2], sk3 = p.seed_k[3];
	const uint64_t rk0 = p.seedrc_k[0], rk1 = p.seedrc_k[1], rk2 = p.seedrc_k[2], rk3 = p.seedrc_k[3];
	const uint64_t sm0 = p.seed_km1[0], sm1 = p.seed_km1[1], sm2 = p.seed_km1[2], sm3 = p.seed_km1[3];
	const uint64_t rm0 = p.seedrc_km1[0], rm1 = p.seedrc_km1[1], rm2 = p.seedrc_km1[2], rm3 = p.seedrc_km1[3];
	const uint32_t lane0 = COOP ? lane_id() : 0u, lstep = COOP ? BULK_LANES : 1u;
	// the k-mer of position l of the chunk, in the walker's orientation, and its strand hashes
	auto vertex_at = [&](uint32_t l, Kmer<NW>& s, uint64_t& fh, uint64_t& rh) {
		s = window_kmer<NW>(gwords, woff, up ? pos + l : pos - l, k);
		if (!same) s = kmer_revcomp_fast(s, k);
		kmer_hashes(s, k, fh, rh);
	};
	// hashes of the neighbour of (s, fh, rh) with base b in direction sense (neighbour_hashes)
	auto nbr = [&](const Kmer<NW>& s, uint64_t fh, uint64_t rh, int sense, unsigned b, uint64_t& nfh, uint64_t& nrh) {
		if (sense == SENSE) {
			const unsigned out = (unsigned)s.w[0] & 3u;
			nfh = srol1(fh) ^ pick4(out, sk0, sk1, sk2, sk3) ^ seed_of(b);
			nrh = sror1(rh ^ seed_of(3u - out)) ^ pick4(b, rm0, rm1, rm2, rm3);
		} else {
			const unsigned out = kmer_get(s, k - 1);
			nfh = sror1(fh ^ seed_of(out)) ^ pick4(b, sm0, sm1, sm2, sm3);
			nrh = (srol1(rh) ^ pick4(out, rk0, rk1, rk2, rk3)) ^ seed_of(3u - b);
		}
	};
	for (uint32_t l = lane0; l < 2 * BULK_LANES; l += lstep) bs.dup[l] = 0;
	if (lane0 == 0) { bs.dupstop = n; bs.full = 0; }
	const uint64_t tq1 = dbg_clock(e.dbg);
	// ---- every predicted vertex: identity, "new to the walker", "simple", "continues as predicted"
	Kmer<NW> my_s; uint64_t my_fh = 0, my_rh = 0;
	// spaced seed: the masked-out terms of the lane's vertex, of its neighbours behind and of those ahead
	uint64_t my_df = 0, my_dr = 0, my_bdf = 0, my_bdr = 0, my_fdf = 0, my_fdr = 0;
	auto terms = [&]() {
		if constexpr (MASKED_BUILD<NW>) {
			masked_terms(p, my_s, my_df, my_dr);
			masked_terms_shifted(p, my_s, my_df, my_dr, bsense, my_bdf, my_bdr);
			masked_terms_shifted(p, my_s, my_df, my_dr, fsense, my_fdf, my_fdr);
		}
	};
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) my_s.w[j] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	// the strand hashes of the chunk's k-mers, all lanes at once (stretch_hashes_wave: two prefix scans over the
	// read's bases instead of one k-round hash per lane); lane l examines position pos +/- l
	uint64_t st_fh = 0, st_rh = 0;
	if (COOP) {
		stretch_hashes_wave<NW>(gwords, woff, up ? pos : pos - (n - 1u), n, k, st_fh, st_rh);
		if (!up) { const unsigned src = lane0 < n ? n - 1u - lane0 : lane0; st_fh = shfl64(st_fh, src); st_rh = shfl64(st_rh, src); }
		if (!same) { const uint64_t t = st_fh; st_fh = st_rh; st_rh = t; } // (the reverse complement's hashes are the strand hashes swapped)
	}
#endif
	const uint64_t tx0 = dbg_clock(e.dbg);
	uint64_t tx1 = tx0, tx2 = tx0, tx3 = tx0, tx4 = tx0;
	for (uint32_t l = lane0; l < n; l += lstep) {
#if defined(__HIP_DEVICE_COMPILE__)
		if (COOP) {
			my_s = window_kmer<NW>(gwords, woff, up ? pos + l : pos - l, k);
			if (!same) my_s = kmer_revcomp_fast(my_s, k);
			my_fh = st_fh; my_rh = st_rh;
		} else
#endif
		vertex_at(l, my_s, my_fh, my_rh);
		terms();
		const VKey key = kmer_ident(p, my_s, my_fh, my_rh, my_df, my_dr);
		bs.key[l] = key;
		// "new to the walker": the home slot of (key, owner) is read while the probes below are in flight -- nearly always
		// free, which settles it (only this walker enters keys of its own, and it is not entering any now)
		const uint64_t home = ld_coherent(&tab.hmin[wt_slot(tab, wt_key(key), owner)]);
		// what an earlier walker left about this k-mer (Guide::seen), and the base the read goes on with: both loads beside the one above
		const uint32_t rpos = up ? pos + l : pos - l; // the vertex's k-mer in the read
		const uint64_t seen_at = woff * 16u + rpos;
		const unsigned known = seen ? (unsigned)seen[seen_at] : 0u;
		const uint32_t rb_pos = (l + 1 < n) ? (up ? pos + l + k : pos - l - 1u) : (up ? pos + l : pos - l); // (always inside the read)
		const uint32_t rb_word = gwords[woff + (rb_pos >> 4)];
		tx1 = dbg_clock(e.dbg);
		unsigned bad = 0; // bit q: neighbour q (q < 4 behind, q >= 4 ahead) is not in the solid filter
		// A cooperative caller skips the probes when EVERY vertex of the chunk is known (a wave goes round for its slowest lane
		// anyway); a serial one vertex by vertex.
		const bool skip_probes = COOP ? !wave_any(l < n && !(known & SV_VALID)) : (known & SV_VALID) != 0;
		if (skip_probes) {
			if (known & SV_SIMPLE) {
				// the bases as the walker holds the k-mer: its reverse complement swaps the sides and complements the bases
				const unsigned nx = (known >> 1) & 3u, pv = (known >> 3) & 3u;
				const unsigned sense_b = same ? nx : 3u - pv, anti_b = same ? pv : 3u - nx; // neighbour added at the end / at the front
				const unsigned fbk = fsense == SENSE ? sense_b : anti_b, bbk = fsense == SENSE ? anti_b : sense_b;
				bad = 0xFFu & ~((1u << bbk) | (1u << (4u + fbk)));
			} else bad = 0xFFu; // (not a simple vertex: no step through it)
		} else {
			// Two stages.  Six of the eight neighbours do not exist, and nearly all of those fail the first
			// hash function already: that one is probed for all eight, the other nh - 1 only for the neighbours
			// that passed it, two of them per round (the usual survivors: the one behind and the one ahead).
			// The same verdicts as probing everything at once for 8 + ~2 (nh - 1) positions instead of 8 nh.
			auto nbr_h = [&](unsigned q) -> uint64_t {
				uint64_t nfh, nrh;
				nbr(my_s, my_fh, my_rh, q < 4 ? bsense : fsense, q & 3u, nfh, nrh);
				nfh ^= q < 4 ? my_bdf : my_fdf; nrh ^= q < 4 ? my_bdr : my_fdr;
				return nrh < nfh ? nrh : nfh;
			};
			uint8_t c0[8];
#pragma unroll
			for (unsigned q = 0; q < 8; q++) c0[q] = (uint8_t)probe_c(p, cnt, pos_i(p, nbr_h(q), 0u));
#pragma unroll
			for (unsigned q = 0; q < 8; q++) bad |= (c0[q] < p.kc ? 1u : 0u) << q;
			unsigned surv = p.nh > 1 ? ~bad & 0xFFu : 0u;
			tx2 = dbg_clock(e.dbg);
			// (four survivors a round: a wave goes round until its slowest lane is done, and one lane in three has a false
			// positive among its six absent neighbours -- with two a round that was 2-3 rounds for nearly every wave.  Their
			// hashes are computed again: eight of them kept across the first probes went to scratch)
			while (COOP ? wave_any(surv != 0) : surv != 0) {
				if (surv) {
					unsigned qs[4];
#pragma unroll
					for (unsigned t = 0; t < 4; t++) { qs[t] = surv ? (unsigned)__builtin_ctz(surv) : qs[0]; surv &= surv - 1; }
					uint64_t hs[4];
#pragma unroll
					for (unsigned t = 0; t < 4; t++) hs[t] = nbr_h(qs[t]);
					for (unsigned base = 1; base < p.nh; base += 3) {
						uint8_t cc[4][3];
#pragma unroll
						for (unsigned i = 0; i < 3; i++) {
							const unsigned hi = base + i < p.nh ? base + i : 0u;
#pragma unroll
							for (unsigned t = 0; t < 4; t++) cc[t][i] = (uint8_t)probe_c(p, cnt, pos_i(p, hs[t], hi)); // (a lane with fewer survivors probes its first again: a load under a lane's own condition is waited for on the spot)
						}
#pragma unroll
						for (unsigned i = 0; i < 3; i++)
#pragma unroll
							for (unsigned t = 0; t < 4; t++) bad |= (cc[t][i] < p.kc ? 1u : 0u) << qs[t];
					}
				}
			}
			if (seen && !(known & SV_VALID)) {
				// for whoever comes next: simple or not, and the two neighbours as the read runs
				const unsigned bm = ~bad & 0xFu, fm = (~bad >> 4) & 0xFu;
				unsigned v = SV_VALID;
				if (bm != 0 && !(bm & (bm - 1)) && fm != 0 && !(fm & (fm - 1))) {
					const unsigned bbx = (bm & 1u) ? 0u : (bm & 2u) ? 1u : (bm & 4u) ? 2u : 3u, fbx = (fm & 1u) ? 0u : (fm & 2u) ? 1u : (fm & 4u) ? 2u : 3u;
					const unsigned sense_b = fsense == SENSE ? fbx : bbx, anti_b = fsense == SENSE ? bbx : fbx;
					const unsigned nx = same ? sense_b : 3u - anti_b, pv = same ? anti_b : 3u - sense_b;
					v |= SV_SIMPLE | (nx << 1) | (pv << 3);
				}
				seen[seen_at] = (uint8_t)v;
			}
		}
		tx3 = dbg_clock(e.dbg);
		bool ok = home == WT_EMPTY;
		if (!ok) {
			const uint64_t fs = wt_find(tab, wt_key(key), owner);
			ok = fs == WT_EMPTY || (uint32_t)ld_coherent(&tab.meta[fs]) == WT_TOMB;
		}
		const unsigned bmask = ~bad & 0xFu, fmask = (~bad >> 4) & 0xFu;
		ok = ok && bmask != 0 && !(bmask & (bmask - 1)) && fmask != 0 && !(fmask & (fmask - 1));
		const unsigned bb = (bmask & 1u) ? 0u : (bmask & 2u) ? 1u : (bmask & 4u) ? 2u : 3u;
		const unsigned fb = (fmask & 1u) ? 0u : (fmask & 2u) ? 1u : (fmask & 4u) ? 2u : 3u;
		if (l == 0) {
			// the one vertex behind the head must be the previous vertex of the path; for the later
			// positions it is: position l - 1 is a neighbour behind position l and is solid (it passed)
			uint64_t tfh, trh;
			nbr(my_s, my_fh, my_rh, bsense, bb, tfh, trh);
			Kmer<NW> ts = my_s;
			kmer_shift(ts, k, bsense, bb);
			ok = ok && key_equal(kmer_ident(p, ts, tfh, trh, my_bdf, my_bdr), prev_key);
		}
		if (l + 1 < n) {
			// the read's next k-mer adds this base at the walking end
			const unsigned rb = (rb_word >> (2u * (rb_pos & 15u))) & 3u;
			ok = ok && fb == (same ? rb : 3u - rb);
		}
		bs.good[l] = ok ? 1 : 0;
		bs.fbase[l] = (uint8_t)fb;
		tx4 = dbg_clock(e.dbg);
	}
	wave_sync();
	const uint64_t tq2 = dbg_clock(e.dbg);
	if (e.dbg && lane0 == 0) { w.t_bx[0] += tx0 - tq1; w.t_bx[1] += tx1 - tx0; w.t_bx[2] += tx2 - tx1; w.t_bx[3] += tx3 - tx2; w.t_bx[4] += tx4 - tx3; w.t_bx[5] += tq2 - tx4; }
	// ---- a vertex that repeats an earlier one of the chunk (a cycle within the read) stops the prefix
	for (uint32_t l = lane0; l < n; l += lstep) {
		const VKey key = bs.key[l];
		uint32_t slot = (uint32_t)((key.fh ^ (key.fh >> 32) ^ key.rh) * 0x9E3779B9u >> 16) & (2 * BULK_LANES - 1);
		for (;;) {
			const uint32_t old = cas_u32(&bs.dup[slot], 0u, l + 1u);
			if (old == 0) break;
			if (key_equal(bs.key[old - 1], key)) { atomic_min_u32(&bs.dupstop, old - 1 > l ? old - 1 : l); break; }
			slot = (slot + 1) & (2 * BULK_LANES - 1);
		}
	}
	wave_sync();
	uint32_t m = 0;
	if (COOP) {
		const uint64_t g = wave_ballot(lane0 < n && bs.good[lane0 < n ? lane0 : 0] != 0);
		const uint64_t ng = ~g;
		m = ng ? (uint32_t)__builtin_ctzll(ng) : 64u;
		if (m > n) m = n;
	} else {
		while (m < n && bs.good[m]) m++;
	}
	{
		const uint32_t ds = uni32<COOP>(ld_coherent(&bs.dupstop));
		if (ds < m) m = ds;
	}
	const uint64_t tq3 = dbg_clock(e.dbg);
	if (e.dbg) { w.t_bp[0] += tq1 - tq0; w.t_bp[1] += tq2 - tq1; w.t_bp[2] += tq3 - tq2; }
	if (m == 0) return 0;
	// ---- take the m steps
	for (uint32_t l = lane0; l < m; l += lstep) {
		const int ins = wt_insert(tab, wt_key(bs.key[l]), owner, contig, false);
		if (ins != WT_NEW) bs.full = 1; // (not new can only mean: no room)
		buf[nbuf + l] = bs.fbase[l];
		if (l == m - 1) {
			// the new head: the neighbour ahead of the last vertex taken
			if (!COOP) { vertex_at(l, my_s, my_fh, my_rh); terms(); }
			uint64_t nfh, nrh;
			const unsigned fb = bs.fbase[l];
			nbr(my_s, my_fh, my_rh, fsense, fb, nfh, nrh);
			kmer_shift(my_s, k, fsense, fb);
#pragma unroll
			for (int j = 0; j < KW<NW>; j++) bs.hw[j] = my_s.w[j];
			bs.hfh = nfh; bs.hrh = nrh; bs.hdf = my_fdf; bs.hdr = my_fdr;
		}
	}
	wave_sync(); // (write-through stores, acknowledged: the table entries are there for whoever looks next)
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) w.head.s.w[j] = uni64<COOP>(bs.hw[j]);
	w.head.fh = uni64<COOP>(bs.hfh); w.head.rh = uni64<COOP>(bs.hrh);
	vtx_set_d(w.head, uni64<COOP>(bs.hdf), uni64<COOP>(bs.hdr));
	{
		const VKey pk = bs.key[m - 1];
		w.prev_key.fh = uni64<COOP>(pk.fh); w.prev_key.rh = uni64<COOP>(pk.rh);
	}
	w.ext += m;
	if (dir == FORWARD) w.nr = nbuf + m; else w.nl = nbuf + m;
	w.bulk_skip = m < n ? 1u : 0u; // the vertex the prefix stopped at goes through the step-by-step code
	if (uni32<COOP>(ld_coherent(&bs.full))) w.bulk_overflow = 1;
	w.n_bulk_calls++; w.n_bulk_steps += m;
	if (e.dbg) w.t_bp[3] += dbg_clock(e.dbg) - tq3;
	return m;
	}
}

// The unbranched stretch of extendPath as a loop of its own.  A step is "simple" when the head,
// newly entered into the visited set, has exactly one neighbour on either side and the one
// behind it is the vertex the path came from: then successor() answers at its level 0 in both
// directions (ExtendPath.h:314-362) and the path grows by the one neighbour ahead.  The loop
// keeps that state -- and wave-uniform copies of every parameter it reads -- in registers and
// takes as many simple steps as it can; anything else (a branch, a dead end, a cycle, an entry of
// an earlier contig, a full buffer) is handed back to
// walk_extend, which runs the step as written in the reference.  Being out of line, the loop
// has a register allocation of its own: the searches the general code calls do not spill into it.
// Entered with w.head pushed but not yet entered into the visited set.
enum { LIN_GENERAL = 0, // w.head is in the visited set; its step is not simple
       LIN_INS = 1,     // inserting w.head returned w.ins (not WT_NEW); nothing else was done for it
       LIN_OVERFLOW = 3 }; // the vertex table filled up (during a bulk step)
template <int NW, bool COOP>
ABG_HDX uint32_t walk_linear(const WalkEnv<NW>& e, WalkState<NW>& w, const int dir_in, const uint32_t owner_in,
    const uint32_t contig_in)
{
	const Params p = uniform_params<COOP>(e.p);
	const uint8_t* __restrict__ cnt = uniptr<COOP>(e.cnt);
	WalkTab tab;
	tab.hmin = uniptr<COOP>(e.tab.hmin); tab.hmax = uniptr<COOP>(e.tab.hmax); tab.meta = uniptr<COOP>(e.tab.meta);
	tab.mask = uni64<COOP>(e.tab.mask);
	const uint32_t buf_cap = uni32<COOP>(e.buf_cap);
	const int dir = (int)uni32<COOP>((uint32_t)dir_in);
	const uint32_t owner = uni32<COOP>(owner_in), contig = uni32<COOP>(contig_in);
	const int fsense = (dir == FORWARD) ? SENSE : ANTISENSE, bsense = (dir == FORWARD) ? ANTISENSE : SENSE;
	uint8_t* buf = uniptr<COOP>(dir == FORWARD ? w.rbuf : w.lbuf);
	uint32_t nbuf = uni32<COOP>(dir == FORWARD ? w.nr : w.nl);
	uint32_t ext = uni32<COOP>(w.ext);
	Vtx<NW> head;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) head.s.w[j] = uni64<COOP>(w.head.s.w[j]);
	head.fh = uni64<COOP>(w.head.fh); head.rh = uni64<COOP>(w.head.rh);
	vtx_set_d(head, uni64<COOP>(vtx_df(w.head)), uni64<COOP>(vtx_dr(w.head)));
	VKey prev_key;
	prev_key.fh = uni64<COOP>(w.prev_key.fh); prev_key.rh = uni64<COOP>(w.prev_key.rh);
	// The rolling-hash tables as named scalars: an array in registers that is indexed at run time
	// (even through a chain of selects, which the optimiser folds back into an indexed load)
	// would be demoted to per-lane scratch.
	const uint64_t sk0 = p.seed_k[0], sk1 = p.seed_k[1], sk2 = p.seed_k[2], sk3 = p.seed_k[3];
	const uint64_t rk0 = p.seedrc_k[0], rk1 = p.seedrc_k[1], rk2 = p.seedrc_k[2], rk3 = p.seedrc_k[3];
	const uint64_t sm0 = p.seed_km1[0], sm1 = p.seed_km1[1], sm2 = p.seed_km1[2], sm3 = p.seed_km1[3];
	const uint64_t rm0 = p.seedrc_km1[0], rm1 = p.seedrc_km1[1], rm2 = p.seedrc_km1[2], rm3 = p.seedrc_km1[3];
	auto pick = [](unsigned i, uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3) -> uint64_t {
		return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3;
	};
	const unsigned k = p.k;
	uint32_t why;
	int32_t ins = WT_NEW;
	// read-guided bulk steps (walk_bulk)
	const uint64_t* __restrict__ gtab = uniptr<COOP>(e.guide.tab);
	const uint64_t gmask = uni64<COOP>(e.guide.mask);
	const bool bulk_on = gtab != nullptr && uniptr<COOP>(w.bulk) != nullptr && p.nh <= 8;
	uint32_t bulk_skip = 0, lin_steps = 0;
	for (;;) {
		if (bulk_on) {
			if (!bulk_skip) {
				const uint64_t hm = head.rh < head.fh ? head.rh : head.fh;
				const uint64_t hint = uni64<COOP>(gtab[guide_slot(hm, gmask)]);
				if ((hint >> 63) && ((uint32_t)(hint >> 55) & 0xFFu) == guide_tag(hm)) {
					w.head = head; w.prev_key = prev_key; w.ext = ext;
					if (dir == FORWARD) w.nr = nbuf; else w.nl = nbuf;
					const uint64_t tb0 = dbg_clock(e.dbg);
					const uint32_t took = uni32<COOP>(walk_bulk<NW, COOP>(e, w, dir, owner, contig, hint));
					if (e.dbg) { w.t_bulk += dbg_clock(e.dbg) - tb0; w.n_bulk_tries++; }
					if (took) {
#pragma unroll
						for (int j = 0; j < KW<NW>; j++) head.s.w[j] = uni64<COOP>(w.head.s.w[j]);
						head.fh = uni64<COOP>(w.head.fh); head.rh = uni64<COOP>(w.head.rh);
						vtx_set_d(head, uni64<COOP>(vtx_df(w.head)), uni64<COOP>(vtx_dr(w.head)));
						prev_key.fh = uni64<COOP>(w.prev_key.fh); prev_key.rh = uni64<COOP>(w.prev_key.rh);
						ext = uni32<COOP>(w.ext);
						nbuf = uni32<COOP>(dir == FORWARD ? w.nr : w.nl);
						bulk_skip = uni32<COOP>(w.bulk_skip);
						if (uni32<COOP>(w.bulk_overflow)) { why = LIN_OVERFLOW; break; }
						continue;
					}
				}
			}
			bulk_skip = 0;
		}
		// rolling states of the head shifted one base either way, before the incoming base is
		// added (neighbour_hashes; NTC64 / NTC64L, nthash.hpp:242-304)
		const unsigned out_s = kmer_get(head.s, 0), out_a = kmer_get(head.s, k - 1);
		const uint64_t fb_s = srol1(head.fh) ^ pick(out_s, sk0, sk1, sk2, sk3);
		const uint64_t rb_s = sror1(head.rh ^ seed_of(3u - out_s));
		const uint64_t fb_a = sror1(head.fh ^ seed_of(out_a));
		const uint64_t rb_a = srol1(head.rh) ^ pick(out_a, rk0, rk1, rk2, rk3);
		// hashes of neighbour `b` in direction `sense`
		auto nbr = [&](int sense, unsigned b, uint64_t& fh, uint64_t& rh) {
			if (sense == SENSE) { fh = fb_s ^ seed_of(b); rh = rb_s ^ pick(b, rm0, rm1, rm2, rm3); }
			else { fh = fb_a ^ pick(b, sm0, sm1, sm2, sm3); rh = rb_a ^ seed_of(3u - b); }
		};
		// spaced seed: the masked-out terms the four neighbours of either side share
		uint64_t df_s = 0, dr_s = 0, df_a = 0, dr_a = 0;
		if constexpr (MASKED_BUILD<NW>) {
			masked_terms_shifted(p, head.s, head.df, head.dr, SENSE, df_s, dr_s);
			masked_terms_shifted(p, head.s, head.df, head.dr, ANTISENSE, df_a, dr_a);
			df_s = uni64<COOP>(df_s); dr_s = uni64<COOP>(dr_s); df_a = uni64<COOP>(df_a); dr_a = uni64<COOP>(dr_a);
		}
		// canonical (masked) hash of that neighbour
		auto nbr_hash_c = [&](int sense, unsigned b) -> uint64_t {
			uint64_t fh, rh;
			nbr(sense, b, fh, rh);
			fh ^= (sense == SENSE) ? df_s : df_a;
			rh ^= (sense == SENSE) ? dr_s : dr_a;
			return rh < fh ? rh : fh;
		};
		// probe round over the 8 neighbours (q < 4: behind, q >= 4: ahead), in flight while the
		// head enters the visited set (ExtendPath.h:650-658)
		uint8_t my_c = 255; bool my_active = false;
		if (COOP) {
			const unsigned lane = lane_id(), q = lane >> 3, i = lane & 7;
			const uint64_t h = nbr_hash_c(q < 4 ? bsense : fsense, q & 3u);
			my_active = i < p.nh;
			if (my_active) my_c = (uint8_t)probe_c(p, cnt, pos_i(p, h, i));
		}
		const VKey hkey = vtx_ident(p, head);
		ins = (int32_t)uni32<COOP>((uint32_t)wt_insert(tab, wt_key(hkey), owner, contig, COOP));
		if (ins != WT_NEW) { why = LIN_INS; break; }
		unsigned m8 = 0xFFu;
		if (COOP) {
			const uint64_t bad = wave_ballot(my_active && my_c < p.kc);
#pragma unroll
			for (unsigned q = 0; q < 8; q++)
				if ((bad >> (8 * q)) & 0xFFu) m8 &= ~(1u << q);
		} else {
			for (unsigned q = 0; q < 8; q++)
				if (!solid_contains(p, cnt, nbr_hash_c(q < 4 ? bsense : fsense, q & 3u))) m8 &= ~(1u << q);
		}
		const unsigned bmask = m8 & 0xFu, fmask = m8 >> 4;
		why = LIN_GENERAL;
		if (bmask == 0 || (bmask & (bmask - 1)) || fmask == 0 || (fmask & (fmask - 1)) || nbuf >= buf_cap) break;
		const unsigned bb = (bmask & 1u) ? 0u : (bmask & 2u) ? 1u : (bmask & 4u) ? 2u : 3u;
		const unsigned fb = (fmask & 1u) ? 0u : (fmask & 2u) ? 1u : (fmask & 4u) ? 2u : 3u;
		// look behind (extendPathBySingleVertex, ExtendPath.h:417-437): the one vertex behind must be the previous one
		Vtx<NW> t = head;
		kmer_shift(t.s, k, bsense, bb);
		nbr(bsense, bb, t.fh, t.rh);
		vtx_set_d(t, (bsense == SENSE) ? df_s : df_a, (bsense == SENSE) ? dr_s : dr_a);
		if (!key_equal(vtx_ident(p, t), prev_key)) break;
		// path.push_back(v) / push_front(v)
		wu_st_u8(&buf[nbuf], (uint8_t)fb, COOP);
		nbuf++; ext++; lin_steps++;
		prev_key = hkey;
		uint64_t nfh, nrh;
		nbr(fsense, fb, nfh, nrh);
		kmer_shift(head.s, k, fsense, fb);
		head.fh = nfh; head.rh = nrh;
		vtx_set_d(head, (fsense == SENSE) ? df_s : df_a, (fsense == SENSE) ? dr_s : dr_a);
	}
	w.head = head; w.prev_key = prev_key; w.ext = ext; w.ins = ins;
	if (dir == FORWARD) w.nr = nbuf; else w.nl = nbuf;
	w.n_lin_steps += lin_steps;
	return why;
}

// extendPath (ExtendPath.h:620-706) with the ExtendPathParams of processRead
// (bloom-dbg.h:845-850): trimLen = trim, fpTrim = 5, no length limit, lookBehind = true,
// lookBehindStartVertex = false; extendPathBySingleVertex (:403-459) inlined.
// Returns the ExtCode, or -1 when the walker must stop (status written to *abort).
template <int NW>
ABG_HDN int walk_extend(WalkEnv<NW>& e, WalkState<NW>& w, int dir, uint32_t owner, uint32_t contig,
    SearchScratch<NW>& sc, uint32_t* ext_out, uint32_t* abort, bool* end_earlier)
{
	const Params& p = e.p;
	int other = (dir == FORWARD) ? REVERSE : FORWARD;
	uint32_t n = w.nl + 1 + w.nr;
	Vtx<NW> head = (dir == FORWARD) ? ws_vertex(p, w, n - 1, sc.coop) : ws_vertex(p, w, 0, sc.coop);
	VKey prev_key = vtx_ident(p, head); // identity of the vertex before the head (the head itself when n == 1)
	if (n > 1) prev_key = vtx_ident(p, (dir == FORWARD) ? ws_vertex(p, w, n - 2, sc.coop) : ws_vertex(p, w, 1, sc.coop));
	uint32_t ext = 0;
	bool look_behind = false;
	bool pending = false; // the head was pushed but not yet entered into `visited`
	const bool split = sc.coop && p.nh <= 8;
	// runs of simple steps go to walk_linear (cooperative probes hold up to 8 hash functions)
	const bool lean = p.nh <= 8 || !sc.coop;
	int ins_given = -1; // walk_linear already inserted the head: what the insertion returned
	int result;
	for (;;) {
		if (pending && lean) {
			w.head = head; w.prev_key = prev_key; w.ext = ext;
			const uint64_t tl0 = dbg_clock(e.dbg);
			const uint32_t why = sc.coop ? walk_linear<NW, true>(e, w, dir, owner, contig)
			                             : walk_linear<NW, false>(e, w, dir, owner, contig);
			if (e.dbg) w.t_lin += dbg_clock(e.dbg) - tl0;
			head = w.head; prev_key = w.prev_key; ext = w.ext;
			n = w.nl + 1 + w.nr;
			if (why == LIN_OVERFLOW) { *abort = WS_OVERFLOW; return -1; }
			if (why == LIN_INS) {
				ins_given = w.ins;
			} else {
				*end_earlier = false; // the head was new to the table
				look_behind = true;
				pending = false;
			}
		}
		Vtx<NW> t, v;
		// both neighbourhoods of the head in one probe round (8 k-mers x H counters in flight)
		uint64_t bfh[4], brh[4], ffh[4], frh[4];
		neighbour_hashes(p, head, (other == FORWARD) ? SENSE : ANTISENSE, bfh, brh);
		neighbour_hashes(p, head, (dir == FORWARD) ? SENSE : ANTISENSE, ffh, frh);
		uint64_t h8[8];
		{
			uint64_t bdf, bdr, fdf, fdr; // spaced seed: masked-out terms (zero without one)
			neighbour_mask_delta(p, head, (other == FORWARD) ? SENSE : ANTISENSE, bdf, bdr);
			neighbour_mask_delta(p, head, (dir == FORWARD) ? SENSE : ANTISENSE, fdf, fdr);
#pragma unroll
			for (unsigned q = 0; q < 4; q++) {
				uint64_t bf = bfh[q] ^ bdf, br = brh[q] ^ bdr, ff = ffh[q] ^ fdf, fr = frh[q] ^ fdr;
				h8[q] = br < bf ? br : bf;
				h8[4 + q] = fr < ff ? fr : ff;
			}
		}
		// start the probe loads, then enter the head into `visited` while they are in flight
		Probe8 pr;
		if (split) pr = probe8_issue(p, e.cnt, h8);
		if (pending) {
			// visited.insert(head), ExtendPath.h:650-658
			int ins = ins_given >= 0 ? ins_given : wt_insert(e.tab, vtx_key(p, head), owner, contig, sc.coop);
			ins_given = -1;
			if (ins == WT_FULL) { *abort = WS_OVERFLOW; return -1; }
			if (ins == WT_SAME_CONTIG) {
				// a cycle: path.pop_back() / pop_front()
				result = ER_CYCLE;
				if (dir == FORWARD) w.nr--; else w.nl--;
				n--; ext--;
				break;
			}
			*end_earlier = (ins == WT_EARLIER);
			look_behind = true; // params.lookBehind after the first extension
			pending = false;
		}
		unsigned m8 = split ? probe8_collect(p, pr) : solid_mask8(p, e.cnt, h8, sc.coop);
		unsigned bmask = m8 & 0xFu, fmask = m8 >> 4;
		// extendPathBySingleVertex (ExtendPath.h:403-459)
		if (look_behind) {
			result = successor_fast(p, head, other, bmask, bfh, brh, t);
			if (result < 0) {
				const uint64_t t0 = dbg_clock(e.dbg);
				result = successor_m(p, e.cnt, head, other, p.trim, bmask, t, sc);
				if (e.dbg) { sc.dbg_search += dbg_clock(e.dbg) - t0; sc.dbg_calls++; }
			}
			if (result == ER_AMBI_OUT) { result = ER_AMBI_IN; break; }
			if (n > 1) {
				if (result == ER_DEAD_END) { result = ER_AMBI_IN; break; }
				if (!key_equal(prev_key, vtx_ident(p, t))) { result = ER_AMBI_IN; break; }
			}
		}
		result = successor_fast(p, head, dir, fmask, ffh, frh, v);
		if (result < 0) {
			const uint64_t t0 = dbg_clock(e.dbg);
			result = successor_m(p, e.cnt, head, dir, p.trim, fmask, v, sc);
			if (e.dbg) { sc.dbg_search += dbg_clock(e.dbg) - t0; sc.dbg_calls++; }
		}
		if (sc.overflow) { *abort = WS_OVERFLOW; return -1; }
		if (result != ER_LENGTH_LIMIT) break;
		// path.push_back(v) / push_front(v)
		if (dir == FORWARD) {
			if (w.nr >= e.buf_cap) { *abort = WS_OVERFLOW; return -1; }
			w.rbuf[w.nr++] = (uint8_t)kmer_get(v.s, p.k - 1);
		} else {
			if (w.nl >= e.buf_cap) { *abort = WS_OVERFLOW; return -1; }
			w.lbuf[w.nl++] = (uint8_t)kmer_get(v.s, 0);
		}
		n++; ext++;
		prev_key = vtx_ident(p, head);
		head = v;
		pending = true;
	}
	if (sc.overflow) { *abort = WS_OVERFLOW; return -1; }
	*ext_out = ext;
	return result;
}

// isTip (bloom-dbg.h:759-776)
ABG_HD bool is_tip(unsigned length, int left, int right, unsigned trim)
{
	if (length > trim) return false;
	if (left == ER_DEAD_END && (right == ER_DEAD_END || right == ER_AMBI_IN)) return true;
	if (right == ER_DEAD_END && (left == ER_DEAD_END || left == ER_AMBI_IN)) return true;
	return false;
}

enum { CT_LINEAR = 0, CT_CIRCULAR = 1, CT_HAIRPIN = 2 };

// ambiguous(u, dir) (ExtendPath.h:368-374)
template <int NW>
ABG_HDN bool ambiguous1(const Params& p, const uint8_t* cnt, const Vtx<NW>& u, int dir,
    SearchScratch<NW>& sc)
{
	Vtx<NW> v;
	return successor(p, cnt, u, dir, p.trim, v, sc) == ER_AMBI_OUT;
}
// ambiguous(u, expected, dir) (ExtendPath.h:383-397)
template <int NW>
ABG_HDN bool ambiguous2(const Params& p, const uint8_t* cnt, const Vtx<NW>& u, const Vtx<NW>& expected,
    int dir, SearchScratch<NW>& sc)
{
	Vtx<NW> v;
	int r = successor(p, cnt, u, dir, p.trim, v, sc);
	return r == ER_AMBI_OUT || (r == ER_LENGTH_LIMIT && !vtx_equal(p, v, expected));
}

// A walker's scratch.  The search scratch and the path state are handed by reference to out-of-line
// functions, so they live in memory.  As locals that is per-lane scratch: 64 copies per cooperative
// wave and a 256-byte transaction per dword touched.  They are carved out of the walker's fast
// memory (LDS on the device) instead: one copy per wave, read by broadcast; the rest of it is the
// fast tier of the trueBranch stack.
template <int NW>
ABG_HD void walker_scratch(WalkEnv<NW>& e, uint32_t slot, SearchScratch<NW>*& scp, WalkState<NW>*& wp)
{
	char* fast = (char*)e.fast;
	uint32_t fast_bytes = e.fast_bytes;
	const uint32_t sc_bytes = (uint32_t)((sizeof(SearchScratch<NW>) + 15) & ~15ull);
	const uint32_t ws_bytes = (uint32_t)((sizeof(WalkState<NW>) + 15) & ~15ull);
	SearchScratch<NW>& sc = *(SearchScratch<NW>*)fast;
	WalkState<NW>& w = *(WalkState<NW>*)(fast + sc_bytes);
	scp = &sc; wp = &w;
	fast += sc_bytes + ws_bytes; fast_bytes -= sc_bytes + ws_bytes;
	sc.tb = e.tb_pool + (uint64_t)slot * e.tb_cap;
	sc.tb_keys = e.tbk_pool + (uint64_t)slot * e.tb_cap;
	sc.tb_cap = e.tb_cap;
	sc.tbf = nullptr; sc.tbf_keys = nullptr; sc.tbf_cap = 0; sc.tbk_cap = 0;
	sc.la = sc.la_local;
	// the scratch of the read-guided bulk steps (walk_bulk, chain_bulk): in fast memory when that
	// leaves the trueBranch stack a decent fast tier, else in the walker's global scratch
	w.bulk = nullptr;
	if (e.guide.tab) {
		const uint32_t bb = (uint32_t)((sizeof(BulkScratch) + 15) & ~15ull);
		if (fast_bytes >= bb + 64u * (uint32_t)(sizeof(TBFrame<NW>) + sizeof(VKey))) { w.bulk = (BulkScratch*)fast; fast += bb; fast_bytes -= bb; }
		else if (e.bulk_pool) w.bulk = e.bulk_pool + slot;
	}
	// lookAhead's visited set starts in fast memory too (LA_FAST entries)
	sc.la_fast = nullptr; sc.la_fast_cap = 0;
	if (fast_bytes >= LA_FAST * sizeof(VKey) + 64u * (uint32_t)(sizeof(TBFrame<NW>) + sizeof(VKey))) {
		sc.la_fast = (VKey*)fast; sc.la_fast_cap = LA_FAST;
		fast += LA_FAST * sizeof(VKey); fast_bytes -= LA_FAST * (uint32_t)sizeof(VKey);
	}
	sc.guide = e.guide; sc.bulk = w.bulk;
	sc.memo = e.memo; sc.n_memo_hits = 0; sc.n_memo_adds = 0; sc.wstats = e.wstats;
	sc.mcache = e.mcache;
	if (!w.bulk) sc.guide.tab = nullptr;
	{
		// trueBranch keys and frames side by side: three keys per frame (a deep search scans every key of
		// its stack at every call but touches only the top frame)
		uint32_t cap = fast_bytes / (uint32_t)(sizeof(TBFrame<NW>) + 3 * sizeof(VKey));
		sc.tbf_keys = (VKey*)fast;
		sc.tbf = (TBFrame<NW>*)(fast + (((uint64_t)3 * cap * sizeof(VKey) + 15) & ~15ull));
		sc.tbf_cap = cap - 1;
		sc.tbk_cap = 3 * cap;
	}
	w.bulk_skip = 0; w.bulk_overflow = 0; w.n_bulk_calls = 0; w.n_bulk_steps = 0; w.n_lin_steps = 0;
	w.n_bulk_tries = 0; w.t_bulk = 0; w.t_lin = 0; w.t_post = 0; w.t_seed = 0; w.t_ext = 0; w.t_mat = 0; for (int q = 0; q < 6; q++) w.t_bx[q] = 0; w.t_bp[0] = w.t_bp[1] = w.t_bp[2] = w.t_bp[3] = 0;
	sc.overflow = 0;
	sc.dbg_search = 0; sc.dbg_calls = 0; sc.dbg_nodes = 0; sc.dbg_chain = 0; sc.dbg_on = e.dbg ? 1u : 0u; sc.n_chain_steps = 0; sc.dbg_la = 0; sc.dbg_la_calls = 0; sc.dbg_mask = 0; sc.dbg_mask_n = 0; sc.dbg_memo = 0;
	sc.coop = e.coop;
	sc.la_visited = e.la_pool + (uint64_t)slot * LA_MAX_VISITED;
}

#if defined(__HIP_DEVICE_COMPILE__)
// assembledKmers.find(*it) for the read's k-mers from position `it` on, 64 of them per round (bloom-dbg.h:839-843 looks
// them up one after the other: three dependent loads each, 87 times a read -- a sixth of the walkers' time): the first
// position whose k-mer is not in the read's assembled set, nk if there is none.  Only this walker writes entries of its
// own, and it is not walking while it looks: the answers are those of the one-by-one loop.
template <int NW>
ABG_HDN uint32_t next_unassembled(const WalkEnv<NW>& e, uint64_t r, uint32_t it, uint32_t nk, uint32_t owner)
{
	const Params& p = e.p;
	const uint32_t* __restrict__ words = e.batch.words;
	const uint64_t woff = e.batch.woff[r];
	const uint32_t lane = lane_id();
	for (uint32_t base = it; base < nk; base += 64u) {
		const uint32_t n = nk - base < 64u ? nk - base : 64u;
		uint64_t fh, rh;
		stretch_hashes_wave<NW>(words, woff, base, n, p.k, fh, rh);
		bool missing = false;
		if (lane < n) {
			const Kmer<NW> s = window_kmer<NW>(words, woff, base + lane, p.k);
			uint64_t df = 0, dr = 0;
			if constexpr (MASKED_BUILD<NW>) masked_terms(p, s, df, dr);
			const uint64_t fs = wt_find(e.tab, wt_key(kmer_ident(p, s, fh, rh, df, dr)), owner);
			missing = fs == WT_EMPTY || (uint32_t)ld_coherent(&e.tab.meta[fs]) == WT_TOMB;
		}
		const uint64_t b = wave_ballot(missing);
		if (b) return base + (uint32_t)__builtin_ctzll(b);
	}
	return nk;
}
#endif
// One candidate read: the loop of processRead (bloom-dbg.h:839-879).
template <int NW>
ABG_HDN void walk_read(WalkEnv<NW>& e, uint32_t c, uint32_t slot)
{
	const Params& p = e.p;
	const unsigned k = p.k;
	SearchScratch<NW>* scp; WalkState<NW>* wp;
	walker_scratch(e, slot, scp, wp);
	SearchScratch<NW>& sc = *scp;
	WalkState<NW>& w = *wp;
#if defined(__HIP_DEVICE_COMPILE__)
	const uint64_t t_start = wall_clock64();
#else
	const uint64_t t_start = 0;
#endif
	uint64_t total_steps = 0;

	w.lbuf = e.lbuf_pool + (uint64_t)slot * e.buf_cap;
	w.rbuf = e.rbuf_pool + (uint64_t)slot * e.buf_cap;

	const uint64_t r = e.cand_read[c];
	const uint32_t L = e.batch.len[r];
	const uint32_t nk = L - k + 1;
	const uint32_t owner = e.owner_base + c;
	uint32_t first = REC_END, last = REC_END, contig = 0;
	uint32_t abort_status = 0;

	Vtx<NW> cur;
	cur.s = batch_kmer<NW>(e.batch, r, 0, k);
	vtx_rehash(p, cur);
	uint64_t ts0 = dbg_clock(e.dbg);
	for (uint32_t it = 0; it < nk; it++) {
		VKey ckey;
#if defined(__HIP_DEVICE_COMPILE__)
		if (sc.coop) {
			it = next_unassembled(e, r, it, nk, owner);
			if (it >= nk) break;
			cur.s = window_kmer<NW>(e.batch.words, e.batch.woff[r], it, k);
			vtx_rehash(p, cur);
			ckey = vtx_key(p, cur);
		} else
#endif
		{
			if (it > 0) vtx_shift(p, cur, SENSE, (uint32_t)batch_base(e.batch, r, it + k - 1));
			ckey = vtx_key(p, cur);
			// assembledKmers.find(*it), bloom-dbg.h:842
			uint64_t fs = wt_find(e.tab, ckey, owner);
			if (fs != WT_EMPTY && (uint32_t)ld_coherent(&e.tab.meta[fs]) != WT_TOMB) continue;
		}
		if (e.dbg) w.t_seed += dbg_clock(e.dbg) - ts0;
		const uint64_t te0 = dbg_clock(e.dbg);

		w.seed = cur; w.nl = 0; w.nr = 0;
		int ins = wt_insert(e.tab, ckey, owner, contig, sc.coop);
		if (ins == WT_FULL) { abort_status = WS_OVERFLOW; break; }
		bool seed_earlier = (ins == WT_EARLIER);
		bool left_earlier = seed_earlier, right_earlier = seed_earlier;
		uint32_t lext = 0, rext = 0;
		int lcode = walk_extend(e, w, REVERSE, owner, contig, sc, &lext, &abort_status, &left_earlier);
		if (lcode < 0) break;
		int rcode = walk_extend(e, w, FORWARD, owner, contig, sc, &rext, &abort_status, &right_earlier);
		if (rcode < 0) break;
		uint32_t n = w.nl + 1 + w.nr;
		total_steps += n;
		const uint64_t tp0 = dbg_clock(e.dbg);
		if (e.dbg) w.t_ext += tp0 - te0;

		const bool tip = is_tip(n, lcode, rcode, p.trim);
		if (!tip) {
			// materialise the path: S = reverse(lbuf) + seed + rbuf, one slack base each side
			uint64_t need = (uint64_t)n + k - 1 + 2;
			uint64_t off = wu_atomic_add_u64(e.pool_used, need, sc.coop);
			if (off + need > e.pool_cap) {
				wu_atomic_add_u64(&e.wstats[WSTAT_OVF_POOL], 1, sc.coop); // (never NULL for a walker: Engine::ensure_walk)
				abort_status = WS_OVERFLOW; break;
			}
			uint8_t* S = e.pool + off + 1;
			uint64_t slen = (uint64_t)n + k - 1;
			// (a cooperative caller's lanes copy 64 bases at a time)
			for (uint64_t j = sc.coop ? lane_id() : 0u; j < slen; j += sc.coop ? 64u : 1u) S[j] = (uint8_t)ws_base(p, w, (uint32_t)j);
			wave_sync();
			int64_t lo = 0, hi = (int64_t)n; // path = vertices [lo, hi) over S
			// ---- trimBranchKmers (bloom-dbg.h:723-757)
			Vtx<NW> popped[2]; bool popped_earlier[2]; int npopped = 0;
			Vtx<NW> pushed; int pushed_side = 0; // see preprocessCircularContig below
			if (n > 1) {
				Vtx<NW> front = pool_vertex<NW>(p, S, 0, sc.coop), back = pool_vertex<NW>(p, S, n - 1, sc.coop);
				// getContigType (bloom-dbg.h:629-645): edge(back, front) via adjacency (RollingBloomDBG.h:558-574)
				int type = CT_LINEAR;
				{
					uint64_t nfh[4], nrh[4];
					unsigned mask = neighbour_mask(p, e.cnt, back, SENSE, nfh, nrh, sc.coop);
					bool edge = false;
					for (unsigned b = 0; b < 4; b++) {
						if (!((mask >> b) & 1u)) continue;
						Vtx<NW> x = make_neighbour(p, back, SENSE, b, nfh[b], nrh[b]);
						if (vtx_equal(p, x, front)) { edge = true; break; }
					}
					if (edge) {
						Vtx<NW> x = front;
						vtx_shift(p, x, ANTISENSE, kmer_get(back.s, 0));
						type = kmer_equal(p, x.s, back.s) ? CT_CIRCULAR : CT_HAIRPIN;
					}
				}
				// preprocessCircularContig (bloom-dbg.h:648-702).  The vertex it appends is a copy of
				// the other end (or its reverse complement), which under a spaced seed may differ from
				// the window of S at the masked positions, so it is kept as a vertex of its own.
				// (pushed_side 1: appended at the back, -1: at the front)
				if (type != CT_LINEAR && n > 2) {
					bool bstart = ambiguous1(p, e.cnt, front, FORWARD, sc) || ambiguous1(p, e.cnt, front, REVERSE, sc);
					bool bend = ambiguous1(p, e.cnt, back, FORWARD, sc) || ambiguous1(p, e.cnt, back, REVERSE, sc);
					if (bstart && !bend) {
						// push_back(front) or push_back(rc(front)): one more base on the right
						unsigned nb = (type == CT_CIRCULAR) ? kmer_get(front.s, k - 1) : 3u - kmer_get(front.s, 0);
						S[hi + k - 1] = (uint8_t)nb;
						hi++;
						pushed = front; pushed_side = 1;
						if (type != CT_CIRCULAR) vtx_revcomp(p, pushed);
					} else if (!bstart && bend) {
						unsigned nb = (type == CT_CIRCULAR) ? kmer_get(back.s, 0) : 3u - kmer_get(back.s, k - 1);
						S[lo - 1] = (uint8_t)nb;
						lo--;
						pushed = back; pushed_side = -1;
						if (type != CT_CIRCULAR) vtx_revcomp(p, pushed);
					}
				}
				int64_t l = hi - lo;
				Vtx<NW> p0 = pool_vertex<NW>(p, S, lo, sc.coop), p1 = pool_vertex<NW>(p, S, lo + 1, sc.coop);
				Vtx<NW> q1 = pool_vertex<NW>(p, S, hi - 1, sc.coop), q2 = pool_vertex<NW>(p, S, hi - 2, sc.coop);
				if (pushed_side > 0) q1 = pushed;
				if (pushed_side < 0) p0 = pushed;
				(void)l;
				bool amb1 = ambiguous2(p, e.cnt, p0, p1, FORWARD, sc);
				bool amb2 = ambiguous2(p, e.cnt, q1, q2, REVERSE, sc);
				if (amb1) { popped[npopped] = p0; popped_earlier[npopped] = (lo == 0) ? (w.nl ? left_earlier : seed_earlier) : true; npopped++; lo++; }
				if (amb2) { popped[npopped] = q1; popped_earlier[npopped] = (hi == (int64_t)n) ? (w.nr ? right_earlier : seed_earlier) : true; npopped++; hi--; }
				if ((pushed_side < 0 && amb1) || (pushed_side > 0 && amb2)) pushed_side = 0; // trimmed off again
			}
			if (sc.overflow) { abort_status = WS_OVERFLOW; break; }
			// ---- record for outputContig
			uint32_t ri = wu_atomic_add_u32(e.rec_used, 1, sc.coop);
			if (ri >= e.rec_cap) {
				wu_atomic_add_u64(&e.wstats[WSTAT_OVF_RECS], 1, sc.coop);
				abort_status = WS_OVERFLOW; break;
			}
			ContigRec& rec = e.recs[ri];
			rec.seq_off = off + 1 + (uint64_t)lo;
			rec.len = (uint32_t)(hi - lo) + k - 1;
			rec.cand = c; rec.next = REC_END; rec.seed_pos = it;
			rec.left_ext = lext; rec.right_ext = rext;
			rec.left_code = (uint8_t)lcode; rec.right_code = (uint8_t)rcode;
			rec.redundant = 0; rec.pre_redundant = 0; rec.coverage = 0; rec.contig_id = ~0ULL;
			rec.ins_prev = 0; rec.dup_of = REC_END; rec.fp = 0;
			if (last == REC_END) first = ri; else e.recs[last].next = ri;
			last = ri;
			// ---- assembledKmers.insert(contigPath): vertices trimmed off the ends are not
			// part of the contig; forget them unless an earlier contig of this read holds them
			// or the same vertex is still an end of the path (circular / hairpin duplicates)
			if (npopped) {
				Vtx<NW> nf = pool_vertex<NW>(p, S, lo, sc.coop), nb = pool_vertex<NW>(p, S, hi - 1, sc.coop);
				if (pushed_side < 0) nf = pushed;
				if (pushed_side > 0) nb = pushed;
				for (int q = 0; q < npopped; q++) {
					if (popped_earlier[q]) continue;
					if (hi > lo && (vtx_equal(p, popped[q], nf) || vtx_equal(p, popped[q], nb))) continue;
					uint64_t s = wt_find(e.tab, vtx_key(p, popped[q]), owner);
					if (s != WT_EMPTY) wu_st_coherent(&e.tab.meta[s], ((uint64_t)owner << 32) | WT_TOMB, sc.coop);
				}
			}
			// ---- pathToSeq (bloom-dbg.h:130-158) under a spaced seed: a column keeps its 'N' unless
			// a '1' of some path k-mer lies over it.  Vertex i covers column c with mask[c - i];
			// mask[0] == mask[k-1] == '1', so only paths shorter than k can leave columns open.
			if (p.mask && hi > lo && (uint64_t)(hi - lo) < k) {
				const MaskTab& m = *p.mask;
				const uint32_t np = (uint32_t)(hi - lo);
				uint8_t* C = S + lo;
				for (uint32_t col = np; col + 1 < k; col++) {
					// offsets j = col - i for i in [0, np): [col - np + 1, col]
					uint32_t j0 = col - np + 1, j1 = col + 1;
					if (m.ones_prefix[j1] == m.ones_prefix[j0]) C[col] = 4;
				}
			}
		}
		contig++;
		if (e.dbg) { ts0 = dbg_clock(e.dbg); w.t_post += ts0 - tp0; }
	}
	e.first_rec[c] = first;
	e.status[c] = abort_status ? abort_status : (uint32_t)WS_COMPLETE;
	if (e.wstats) {
		wu_atomic_add_u64(&e.wstats[WSTAT_BULK_CALLS], w.n_bulk_calls, sc.coop);
		wu_atomic_add_u64(&e.wstats[WSTAT_BULK_STEPS], w.n_bulk_steps, sc.coop);
		wu_atomic_add_u64(&e.wstats[WSTAT_LIN_STEPS], w.n_lin_steps, sc.coop);
		wu_atomic_add_u64(&e.wstats[WSTAT_CHAIN_STEPS], sc.n_chain_steps, sc.coop);
		wu_atomic_add_u64(&e.wstats[WSTAT_MEMO_HITS], sc.n_memo_hits, sc.coop);
		wu_atomic_add_u64(&e.wstats[WSTAT_MEMO_ADDS], sc.n_memo_adds, sc.coop);
	}
	if (e.dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
		const uint64_t t_end = wall_clock64();
#else
		const uint64_t t_end = 0;
#endif
		uint64_t* d = e.dbg + (uint64_t)c * WALK_DBG_N;
		d[0] = t_end - t_start; d[1] = total_steps; d[2] = sc.dbg_search; d[3] = sc.dbg_calls;
		d[4] = sc.dbg_nodes; d[5] = w.t_bulk; d[6] = contig; d[7] = w.t_post;
		d[8] = w.t_lin; d[9] = w.n_bulk_tries; d[10] = w.n_bulk_calls; d[11] = w.n_bulk_steps; d[12] = sc.dbg_chain;
		d[13] = sc.dbg_la; d[14] = sc.dbg_la_calls; d[15] = w.t_bp[1]; d[16] = sc.dbg_mask; d[17] = sc.dbg_mask_n; d[18] = sc.dbg_memo;
		d[19] = w.t_seed + (dbg_clock(e.dbg) - ts0); d[20] = w.t_ext; d[21] = w.t_mat;
		for (int q = 0; q < 6; q++) d[22 + q] = w.t_bx[q];
	}
}

// ------------------------------------------------------------ commit helpers
// Contig sequences hold codes 0..3 = ACGT and 4 = 'N' (only in columns no '1' of a spaced seed
// covers).  Order and complement as the reference's characters have them: A < C < G < N < T,
// complementBaseChar('N') == 'N'.
ABG_HD unsigned code_rank(unsigned c) { return c == 4 ? 3u : (c == 3 ? 4u : c); }
ABG_HD unsigned code_comp(unsigned c) { return c == 4 ? 4u : 3u - c; }
// Key of an end k-mer of a contig in contigEndKmers (bloom-dbg.h:556-564,578-583):
// canonicalize(Sequence&) (Common/Sequence.h:39-44) turns the k-mer at seq[0..k) into whichever
// of it / its reverse complement is the smaller string; the Vertex built from that text is
// then identified as every vertex is (vtx_ident): its strand hashes ordered by ITS isCanonical().
ABG_HDN VKey canonical_end_key(const Params& p, const uint8_t* seq)
{
	const unsigned k = p.k;
	bool rc_less = false;
	for (unsigned i = 0; i < k; i++) {
		unsigned a = code_rank(code_comp(seq[k - 1 - i])), b = code_rank(seq[i]);
		if (a != b) { rc_less = a < b; break; }
	}
	auto text = [&](unsigned i) -> unsigned { return rc_less ? code_comp(seq[k - 1 - i]) : (unsigned)seq[i]; };
	uint64_t fs, rs;
	scratch_hashes(p, text, fs, rs);
	VKey key;
	if (p.ident_fast) {
		key.fh = rs < fs ? rs : fs;
		key.rh = rs < fs ? fs : rs;
		return wt_key(key);
	}
	bool canon = true; // LightweightKmer::isCanonical (LightweightKmer.h:88-101) of the text
	for (unsigned i = 0; i < k / 2; i++) {
		unsigned c1 = code_rank(text(i)), c2 = code_rank(code_comp(text(k - 1 - i)));
		if (c1 > c2) { canon = false; break; }
		if (c1 < c2) break;
	}
	key.fh = canon ? fs : rs;
	key.rh = canon ? rs : fs;
	return wt_key(key);
}

} // namespace abg
