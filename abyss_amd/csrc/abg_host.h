// abg_host.h -- host side of the C ABI (include/abyss_amd.h) above the engine:
// turns the reference's inputs (ASCII sequences as FastaReader hands them to loadSeq /
// processRead) into packed device batches and maps results back.  Templated over the
// backend like the engine; the product instantiates it for HIP only.
#pragma once
#include "abg_engine.h"
#include "../../include/abyss_amd.h"

#include <cctype>
#include <cmath>
#include <map>
#include <string>
#include <atomic>
#include <chrono>
#include <future>
#include <memory>
#include <thread>

namespace abg {
static_assert(FAIL_NOMEM == ABG_ENOMEM && FAIL_INTERNAL == ABG_EINTERNAL, "Failure codes are the C ABI's");


struct FContainsSolid { // CountingBloomFilter::contains (CountingBloomFilter.hpp:190-196) for every op
	Params p; const uint64_t* h0; const uint8_t* cnt; uint8_t* out;
	ABG_HD void operator()(uint64_t t, uint32_t) const { out[t] = solid_contains(p, cnt, h0[t]) ? 1 : 0; }
};
struct FExpand { // RollingHash::getHashes (RollingHash.h:141-146) for every op
	Params p; const uint64_t* h0; uint64_t* out;
	ABG_HD void operator()(uint64_t t, uint32_t) const
	{
		uint64_t h = h0[t];
		for (unsigned i = 0; i < p.nh; i++) out[t * p.nh + i] = hash_i(p, h, i);
	}
};

// roundUpToMultiple(round(B / 1.125 / sizeof(uint8_t)), 64): bloom-dbg.cc:365-367, BloomIO.h:14-24
inline uint64_t counters_for_budget(uint64_t bloom_bytes)
{
	double sz = (double)bloom_bytes / 1.125 / 1.0;
	uint64_t n = (uint64_t)std::round(sz);
	uint64_t rem = n % 64;
	return rem ? n + 64 - rem : n;
}

template <class BE>
class Session {
  public:
	std::string error;
	BE be;
	Engine<BE>* eng = nullptr;
	Config cfg;

	template <class... A>
	explicit Session(A&&... a) : be(std::forward<A>(a)...) {}
	~Session() { try { drain(); } catch (...) {} keep_drop(false); delete eng; }

	int create(const abg_params& p)
	{
		if (p.k < 2 || p.k > ABG_MAX_KMER) return fail(ABG_EINVAL, "k must be in 2.." + std::to_string(ABG_MAX_KMER));
		if (p.num_hashes < 1 || p.num_hashes > ABG_MAX_HASHES) return fail(ABG_EINVAL, "num_hashes must be in 1..32");
		if (p.spaced_seed && p.spaced_seed[0]) {
			// MaskedKmer::setMask (BloomDBG/MaskedKmer.h:25-48); RollingBloomDBGVertex::compare
			// additionally asserts a symmetric pattern (RollingBloomDBG.h:141-145)
			std::string m(p.spaced_seed);
			if (m.size() != p.k) return fail(ABG_EINVAL, "spaced seed must be exactly k bits long");
			if (m.find_first_not_of("01") != std::string::npos) return fail(ABG_EINVAL, "spaced seed must contain only '0's or '1's");
			if (m.front() != '1' || m.back() != '1') return fail(ABG_EINVAL, "spaced seed must begin and end with '1's");
			if (!std::equal(m.begin(), m.end(), m.rbegin())) return fail(ABG_EINVAL, "spaced seed must be symmetric");
			if (m.find('0') != std::string::npos) cfg.spaced_seed = m; // all '1's == no mask
		}
		cfg.k = p.k; cfg.nh = p.num_hashes; cfg.kc = p.min_cov;
		cfg.trim = (p.trim == 0xFFFFFFFFu) ? p.k : p.trim;
		cfg.counters = p.counters ? p.counters : counters_for_budget(p.bloom_bytes);
		cfg.cascade_levels = p.cascade_levels;
		cfg.slice_filter = p.slice_filter;
		if (const char* e = getenv("ABG_SLICE_FILTER")) cfg.slice_filter = (uint32_t)atoi(e); // partitioned run: each rank keeps its own range of the counters only (1), never (2)
		if (p.cascade_levels && (!p.counters || p.counters % 64)) return fail(ABG_EINVAL, "cascade mode needs `counters` = bits per level, a multiple of 64");
		if (!cfg.counters) return fail(ABG_EINVAL, "bloom_bytes / counters must be > 0");
		cfg.verbose = p.verbose;
		if (p.insert_batch_kmers) cfg.insert_batch_kmers = p.insert_batch_kmers;
		if (p.claim_log2) cfg.claim_log2 = p.claim_log2;
		if (p.walk_slots) cfg.walk_slots = p.walk_slots;
		if (p.wtab_log2) cfg.wtab_log2 = p.wtab_log2;
		if (const char* e = getenv("ABG_CLAIM_LOG2")) cfg.claim_log2 = (uint32_t)atoi(e);
		if (const char* e = getenv("ABG_INSERT_BATCH")) cfg.insert_batch_kmers = strtoull(e, 0, 10);
		if (const char* e = getenv("ABG_WALK_SLOTS")) cfg.walk_slots = (uint32_t)atoi(e);
		if (const char* e = getenv("ABG_WTAB_LOG2")) cfg.wtab_log2 = (uint32_t)atoi(e);
		if (const char* e = getenv("ABG_WTAB_LOG2_MAX")) cfg.wtab_log2_max = (uint32_t)std::max<int>(atoi(e), (int)cfg.wtab_log2); // (tests: forces overflow restarts)
		if (const char* e = getenv("ABG_POOL_CAP")) cfg.pool_cap = std::max<uint64_t>(1024, strtoull(e, 0, 10)); // (tests: the walkers run out of contig pool ...
		if (const char* e = getenv("ABG_REC_CAP")) cfg.rec_cap = (uint32_t)std::max(16, atoi(e));                  //  ... or of contig records, and the engine grows them)
		if (const char* e = getenv("ABG_P2_FIRST_BATCH")) cfg.p2_first_batch = strtoull(e, 0, 10);
		if (const char* e = getenv("ABG_P2_MAX_BATCH")) cfg.p2_max_batch = strtoull(e, 0, 10);
		if (const char* e = getenv("ABG_PAR_COMMIT")) cfg.par_commit = atoi(e) != 0;
		if (const char* e = getenv("ABG_T_TAGS")) cfg.t_tags = (uint32_t)std::max(2, atoi(e));
		if (const char* e = getenv("ABG_PAR_COMMIT_MAX_GB")) cfg.par_commit_max_bytes = strtoull(e, 0, 10) << 30; // e.g. 160 for B=40G on a 288 GB GPU
		if (const char* e = getenv("ABG_COMPACT_THRESHOLD")) cfg.compact_threshold = strtoull(e, 0, 10);
		if (const char* e = getenv("ABG_TILED")) cfg.tiled_insert = atoi(e) != 0; // PASS 1 through LDS tiles
		if (const char* e = getenv("ABG_DIST_ROUTE_MIN")) cfg.dist_route_min_ranks = (uint32_t)atoi(e); // partitioned run: pairs routed to their owners from this many ranks on (0: never)
		if (const char* e = getenv("ABG_BENIGN")) cfg.benign_sharers = atoi(e) != 0; // (diagnosis: 0 sends every k-mer with a shared counter to the rounds)
		if (const char* e = getenv("ABG_CLS_ARCHIVE")) cfg.cls_archive = atoi(e) != 0; // (0: the classification probes the filters for every k-mer)
		if (const char* e = getenv("ABG_CLS_DEBUG_SKIP")) cfg.cls_debug_skip = (uint32_t)atoi(e); // (diagnosis: WRONG verdicts -- FClassify without its look-aheads (1) / its sweep (2))
		if (const char* e = getenv("ABG_CLS_ARCHIVE_MAX_MB")) cfg.cls_archive_max_mb = (uint32_t)std::max(0, atoi(e));
		if (const char* e = getenv("ABG_SORTED_OVERFLOW")) cfg.sorted_overflow = atoi(e) != 0; // (0: a batch that runs a bin over takes the reservation rounds as a whole)
		if (const char* e = getenv("ABG_COSETTLE")) cfg.cosettle = atoi(e) != 0; // (0: round 4's rule -- a k-mer that may write a shared counter takes the rounds)
		if (const char* e = getenv("ABG_CLS_BOTH")) cfg.cls_both = atoi(e) != 0; // (0: the classification reads the plane and the visited filter where they are)
		if (const char* e = getenv("ABG_CLS_BOTH_MAX_MB")) cfg.cls_both_max_mb = (uint32_t)std::max(0, atoi(e));
		if (const char* e = getenv("ABG_COSETTLE_PASSES")) cfg.cosettle_passes = (uint32_t)std::max(1, atoi(e));
		if (const char* e = getenv("ABG_COSETTLE_LOG2")) cfg.cosettle_log2 = (uint32_t)std::min(25, std::max(5, atoi(e))); // (tests: a table in which nearly every counter looks marked)
		if (const char* e = getenv("ABG_LINK_DUPS")) cfg.link_duplicates = atoi(e) != 0; // the commit decides a contig's copies by their original (0: every record bit by bit)
		if (const char* e = getenv("ABG_OVERLAP_PURITY")) cfg.overlap_purity = atoi(e) != 0; // PASS 1: the next batch's tiles judged on the side stream too
		if (const char* e = getenv("ABG_OVERLAP_BINS")) cfg.overlap_bins = atoi(e) != 0; // the next batch hashed and binned beside this one
		if (const char* e = getenv("ABG_PREFETCH")) cfg.prefetch_classify = atoi(e) != 0;
		if (const char* e = getenv("ABG_ASYNC_LOAD")) cfg.async_load = atoi(e) != 0;
		if (const char* e = getenv("ABG_CLS_SLOTS")) cfg.classify_slots = (uint32_t)std::max(64, atoi(e));
		if (const char* e = getenv("ABG_SOLID_PLANE")) cfg.solid_plane = atoi(e) != 0;
		if (const char* e = getenv("ABG_MEMO")) cfg.memo = atoi(e) != 0; // shared answers of successor()
		if (const char* e = getenv("ABG_P2_MAX_CANDIDATES")) cfg.p2_max_candidates = (uint32_t)std::max(1, atoi(e));
		if (const char* e = getenv("ABG_P2_STARVED_GROWTH")) cfg.p2_starved_growth = (uint32_t)std::max(2, atoi(e));
		if (const char* e = getenv("ABG_P2_STARVED")) cfg.p2_starved = (uint32_t)atoi(e);
		if (const char* e = getenv("ABG_P2_GROWTH")) cfg.p2_growth = (uint32_t)std::max(2, atoi(e));
		if (const char* e = getenv("ABG_DRAIN_THRESHOLD")) cfg.drain_threshold = (uint32_t)strtoul(e, 0, 10);
		if (const char* e = getenv("ABG_GUIDE_STRIDE")) cfg.guide_stride = (uint32_t)strtoul(e, 0, 10); // 0: walk step by step
		if (const char* e = getenv("ABG_GUIDE_SEEN")) cfg.guide_seen = atoi(e) != 0; // the bulk steps' findings kept per read k-mer (0: every walker probes again)
		if (const char* e = getenv("ABG_GUIDE_LOG2_MAX")) cfg.guide_log2_max = (uint32_t)std::min(36, std::max(10, atoi(e)));
		if (!be.ok()) return fail(ABG_ENODEV, be.why());
		eng = new Engine<BE>(be, cfg);
		return ABG_OK;
	}

	// ------------------------------------------------------------------ PASS 1
	int load_seqs(const char* seqs, const uint64_t* off, uint64_t n) { return load_seqs_v(1, &seqs, &off, &n); }
	// the same over several buffers, taken one after the other (a reader's blocks as they lie: no
	// concatenation on the caller's side)
	int load_seqs_v(uint32_t nchunks, const char* const* seqs_v, const uint64_t* const* off_v, const uint64_t* n_v)
	{
		const uint32_t k = cfg.k;
		const double t_call = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
		// longest piece handed to the device as one sequence; longer ACGT runs are cut into
		// pieces overlapping by k-1 bases, which yields the same k-mers in the same order
		const uint32_t max_piece = (uint32_t)std::min<uint64_t>(1u << 20, cfg.insert_batch_kmers + k - 1);
		// packing is host work per base: the reads are split over threads, every thread packs its
		// range into a batch of its own, and the batches are joined in order
		struct Part { uint32_t c; uint64_t a, b; };
		std::vector<Part> plan;
		std::vector<uint64_t> base(nchunks + 1, 0);
		for (uint32_t c = 0; c < nchunks; c++) {
			base[c + 1] = base[c] + n_v[c];
			if (!n_v[c]) continue;
			const std::vector<uint64_t> cut = split_reads(off_v[c], n_v[c]);
			for (size_t t = 0; t + 1 < cut.size(); t++) plan.push_back(Part{ c, cut[t], cut[t + 1] });
		}
		const bool keeping = keep_on_ && !keep_failed_;
		// kept reads (keep_reads): what PASS 2 will want of this call's reads.  A read that is ACGT throughout
		// and at least k long is one piece of the batch (or, longer than a piece may be, packed once more on
		// the side); every other read has its verdict now (bloom-dbg.h:804,808).
		std::vector<KeptPart> kparts(keeping ? plan.size() : 0);
		std::vector<uint8_t> verdict(keeping ? base[nchunks] : 0, (uint8_t)RR_UNINITIALIZED);
		std::vector<HostBatch> parts(plan.size());
		const bool timing = getenv("ABG_HOST_TIMING") != nullptr;
		const auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		const double t0 = tnow();
		run_parts(parts.size(), [&](size_t t) {
			HostBatch& hb = parts[t];
			const Part& pt = plan[t];
			const char* seqs = seqs_v[pt.c];
			const uint64_t* off = off_v[pt.c];
			{
				const uint64_t nr = pt.b - pt.a, bases = off[pt.b] - off[pt.a];
				hb.words.reserve(bases / 16 + nr + 1); hb.woff.reserve(nr + 1); hb.len.reserve(nr); hb.koff.reserve(nr + 1);
				if (keeping) { kparts[t].read.reserve(nr); kparts[t].piece.reserve(nr); kparts[t].len.reserve(nr); }
			}
			std::string up;
			std::vector<std::pair<uint64_t, uint64_t>> runs;
			std::vector<uint8_t> bad;
			for (uint64_t i = pt.a; i < pt.b; i++) {
				const char* s = seqs + off[i];
				uint64_t L = off[i + 1] - off[i];
				if (L < k) { if (keeping) verdict[base[pt.c] + i] = (uint8_t)RR_SHORTER_THAN_K; continue; } // RollingHashIterator.h:37-40
				// (case is folded by the code table; the upper-cased copy is only needed under a spaced seed)
				const char* text = s;
				if (!cfg.spaced_seed.empty()) {
					up.assign(s, L);
					for (auto& ch : up) ch = (char)toupper((unsigned char)ch); // RollingHashIterator.h:132
					text = up.data();
				}
				bool clean = false;
				if (keeping) {
					uint8_t any = 0;
					for (uint64_t q = 0; q < L; q++) any |= codes_.t[(unsigned char)s[q]];
					clean = !(any & 0x80);
					if (!clean) verdict[base[pt.c] + i] = (uint8_t)RR_NON_ACGT;
				}
				if (clean && L <= max_piece) {
					// (all of it is one run of k-mers, whatever the seed)
					kparts[t].read.push_back(base[pt.c] + i); kparts[t].piece.push_back(hb.n()); kparts[t].len.push_back((uint32_t)L);
					hb.add_ascii(text, (uint32_t)L, k);
					continue;
				}
				if (clean) { kparts[t].extra.add_ascii(text, (uint32_t)L, k); kparts[t].extra_read.push_back(base[pt.c] + i); }
				valid_runs(text, L, runs, bad);
				for (auto& run : runs) {
					// k-mers run.first .. run.second - 1 start in this piece
					uint64_t pa = run.first, pb = run.second - 1 + k;
					for (uint64_t q = pa; q + k <= pb;) {
						uint64_t e = std::min<uint64_t>(pb, q + max_piece);
						hb.add_ascii(text + q, (uint32_t)(e - q), k);
						if (e == pb) break;
						q = e - (k - 1);
					}
				}
			}
		});
		// The device's share -- upload, the ordered insert, the kept store's book-keeping -- runs on a thread of
		// its own while the caller goes on (to parse and pack the next chunk: packing is host work as long
		// as a chunk's PASS 1).  One at a time, in call order; every other entry point waits for it first
		// (drain), and what it throws surfaces there.
		auto st = std::make_shared<LoadStage>();
		st->rbase.assign(parts.size() + 1, 0);
		for (size_t t = 0; t < parts.size(); t++) st->rbase[t + 1] = st->rbase[t] + parts[t].n();
		const double t1 = tnow();
		if (!parts.empty()) {
			HostBatch joined;
			const HostBatch& hb = join_parts(parts, joined);
			st->hb = (&hb == &joined) ? std::move(joined) : std::move(parts[0]);
		}
		st->keeping = keeping; st->kparts = std::move(kparts); st->verdict = std::move(verdict); st->n_reads = base[nchunks];
		st->t_pack = t1 - t0; st->t_join = tnow() - t1; st->nparts = parts.size();
		// The pipeline's chunks go up AHEAD: this thread uploads the batch's four arrays (Backend::upload_ahead: a stream, pinned
		// buffers and device blocks of its own) while the library's thread is still at the chunk before -- an upload standing in
		// front of every chunk's kernels with the device idle was 0.07 of the 0.41 s configs[1]'s load phase took.
		const bool piped = cfg.async_load && !comm_attached_ && keeping;
		double t_up = tnow();
		if (piped && !keep_failed_ && st->hb.n()) {
			const HostBatch& hb = st->hb;
			const void* src[4] = { hb.words.data(), hb.woff.data(), hb.len.data(), hb.koff.data() };
			const size_t bytes[4] = { hb.words.size() * 4, hb.woff.size() * 8, hb.len.size() * 4, hb.koff.size() * 8 };
			if (be.upload_ahead(src, bytes, 4, st->up)) st->uploaded = true;
		}
		const double t_d0 = tnow();
		t_up = t_d0 - t_up;
		drain();
		if (timing) fprintf(stderr, "[host] load call: plan %.3f s, pack %.3f s, join + hand-over %.3f s, upload ahead %.3f s, waited %.3f s for the device's share of the call before\n", t0 - t_call, t1 - t0, t_d0 - t1 - t_up, t_up, tnow() - t_d0);
		// Only a caller that asked for the pipeline (abg_keep_reads) gets it: otherwise the call does its
		// device work itself and returns with it done and its errors its own.  A partitioned run's
		// collectives are the caller's code: they stay on the caller's thread.
		if (!piped) { load_stage(*st); return ABG_OK; }
		pending_ = std::async(std::launch::async, [this, st]() { be.bind_thread(); load_stage(*st); });
		return ABG_OK;
	}
	// waits for the device's share of the last load call (and rethrows what it threw)
	void drain() { if (pending_.valid()) pending_.get(); }
  private:
	struct KeptPart { std::vector<uint64_t> read, piece; std::vector<uint32_t> len; HostBatch extra; std::vector<uint64_t> extra_read; };
	struct LoadStage {
		HostBatch hb; std::vector<uint64_t> rbase; bool keeping = false; std::vector<KeptPart> kparts; std::vector<uint8_t> verdict;
		uint64_t n_reads = 0; double t_pack = 0, t_join = 0; size_t nparts = 0;
		bool uploaded = false; void* up[4] = { nullptr, nullptr, nullptr, nullptr }; // words, woff, len, koff where upload_ahead put them
	};
	std::future<void> pending_;
	bool comm_attached_ = false;
	void load_stage(LoadStage& st)
	{
		const HostBatch& hb = st.hb;
		const bool timing = getenv("ABG_HOST_TIMING") != nullptr;
		const auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		const double t2 = tnow();
		if (!st.keeping || keep_failed_) {
			if (hb.n()) flush_load(hb); // (the engine cuts it into ordered-insert batches of insert_batch_kmers)
			if (timing) fprintf(stderr, "[host] load: %zu parts, pack %.3f s, join %.3f s, upload + device %.3f s\n", st.nparts, st.t_pack, st.t_join, tnow() - t2);
			return;
		}
		std::vector<KeptPart>& kparts = st.kparts;
		const std::vector<uint64_t>& rbase = st.rbase;
		// the batch's words go to the kept store and stay there; PASS 1 reads them where they are
		uint64_t extra_words = 0;
		for (auto& kp : kparts) extra_words += kp.extra.words.size();
		const uint64_t at = keep_.used;
		if (!keep_room(hb.words.size() + extra_words + 16)) {
			// no room on the device: PASS 1 goes on as if nothing were kept; assemble_kept will say so
			keep_drop(true);
			if (hb.n()) flush_load(hb);
			return;
		}
		if (hb.n()) {
			DevBatch d;
			d.words = nullptr;
			uint32_t* w = (uint32_t*)keep_.words + at;
			if (st.uploaded) {
				// (the arrays are on the device already -- the call put them there while the chunk before was at work: the words move
				// into the store, device to device; the others are read where they lie, in a block that is this chunk's until the
				// call after the next)
				be.d2d(w, st.up[0], hb.words.size() * 4);
				d.woff = st.up[1]; d.len = st.up[2]; d.koff = st.up[3];
			} else {
				be.h2d(w, hb.words.data(), hb.words.size() * 4);
				d.woff = be.alloc(hb.woff.size() * 8);
				d.len = be.alloc(std::max<size_t>(hb.len.size(), 1) * 4);
				d.koff = be.alloc(hb.koff.size() * 8);
				be.h2d(d.woff, hb.woff.data(), hb.woff.size() * 8);
				be.h2d(d.len, hb.len.data(), hb.len.size() * 4);
				be.h2d(d.koff, hb.koff.data(), hb.koff.size() * 8);
			}
			d.b = Batch{ w, (const uint64_t*)d.woff, (const uint32_t*)d.len, (const uint64_t*)d.koff, hb.n() };
			const double t3 = tnow();
			// (the kept store's book-keeping for these reads -- three numbers a read, tens of milliseconds a chunk -- beside the device's work)
			std::future<void> book = std::async(std::launch::async, [&]() { keep_book(st, at); });
			try { eng->load_packed(d.b, hb.koff.data()); } catch (...) { book.wait(); throw; }
			book.get();
			if (!st.uploaded) { be.free(d.woff); be.free(d.len); be.free(d.koff); }
			if (timing) fprintf(stderr, "[host] load: %zu parts, pack %.3f s, join %.3f s, upload %.3f s, device %.3f s\n", st.nparts, st.t_pack, st.t_join, t3 - t2, tnow() - t3);
		} else keep_book(st, at);
		keep_.used = at + hb.words.size();
		const uint64_t r0 = keep_.n_reads;
		for (size_t t = 0; t < kparts.size(); t++) {
			KeptPart& kp = kparts[t];
			if (kp.extra.n()) { // (reads longer than a piece: a copy of their own behind the batch; assemble_kept puts them back into read order)
				be.h2d((uint32_t*)keep_.words + keep_.used, kp.extra.words.data(), kp.extra.words.size() * 4);
				for (size_t j = 0; j < kp.extra.n(); j++) {
					keep_.orig.push_back(r0 + kp.extra_read[j]);
					keep_.woff.push_back(keep_.used + kp.extra.woff[j]);
					keep_.len.push_back(kp.extra.len[j]);
				}
				keep_.used += kp.extra.words.size();
				keep_.unordered = true;
			}
		}
		keep_.n_reads += st.n_reads;
		if (timing) fprintf(stderr, "[host] load: the device's share of the call took %.3f s in all (the kept store's book-keeping included)\n", tnow() - t2);
	}
	// the kept store's entries for a call's one-piece reads (words at `at`), and the call's verdicts so far
	void keep_book(LoadStage& st, uint64_t at)
	{
		const HostBatch& hb = st.hb;
		const uint64_t r0 = keep_.n_reads;
		uint64_t more = 0;
		for (const KeptPart& kp : st.kparts) more += kp.read.size();
		const size_t n0 = keep_.orig.size();
		keep_.orig.resize(n0 + more); keep_.woff.resize(n0 + more); keep_.len.resize(n0 + more);
		size_t o = n0;
		for (size_t t = 0; t < st.kparts.size(); t++) {
			const KeptPart& kp = st.kparts[t];
			const uint64_t* w = hb.woff.data() + st.rbase[t];
			for (size_t j = 0; j < kp.read.size(); j++, o++) {
				keep_.orig[o] = r0 + kp.read[j];
				keep_.woff[o] = at + w[kp.piece[j]];
				keep_.len[o] = kp.len[j];
			}
		}
		keep_.res.insert(keep_.res.end(), st.verdict.begin(), st.verdict.end());
	}
  public:
	// ---- reads kept on the device between the passes (include/abyss_amd.h: abg_keep_reads)
	int keep_reads(int on, uint64_t expected_bases)
	{
		drain();
		keep_drop(false);
		keep_on_ = on != 0;
		if (!keep_on_) return ABG_OK;
		const uint64_t mem = be.device_mem_bytes();
		if (mem && expected_bases / 4 > mem / 8) { keep_on_ = false; return fail(ABG_ENOMEM, "the reads would take more than an eighth of the device's memory: not kept"); }
		keep_.hint_words = expected_bases / 16 + expected_bases / 1024 + 1024;
		return ABG_OK;
	}
	uint64_t kept_reads() { drain(); return keep_on_ ? keep_.n_reads : 0; }
	int assemble_kept(uint8_t* results, abg_contig_cb cb, void* user)
	{
		drain();
		const double t_in = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
		if (!keep_on_) return fail(ABG_EINVAL, "no reads are kept (abg_keep_reads)");
		if (keep_failed_) return fail(ABG_EAGAIN, "the kept reads were dropped (no device memory for them): read the input again");
		if (eng->cascade_mode()) return fail(ABG_EINVAL, "assembly is not available on a cascading filter");
		const uint64_t n = keep_.n_reads, n2 = keep_.orig.size();
		if (keep_.unordered) {
			// (long reads were appended behind their part's batch: back into read order)
			std::vector<uint64_t> idx(n2);
			for (uint64_t j = 0; j < n2; j++) idx[j] = j;
			std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return keep_.orig[a] < keep_.orig[b]; });
			std::vector<uint64_t> o2(n2), w2(n2); std::vector<uint32_t> l2(n2);
			for (uint64_t j = 0; j < n2; j++) { o2[j] = keep_.orig[idx[j]]; w2[j] = keep_.woff[idx[j]]; l2[j] = keep_.len[idx[j]]; }
			keep_.orig.swap(o2); keep_.woff.swap(w2); keep_.len.swap(l2);
			keep_.unordered = false;
		}
		// the reference counts every read in readsProcessed (bloom-dbg.h:1045), also the
		// ones rejected at loading time; the engine counts the ones it sees
		Counters c0 = eng->counters();
		c0.reads_processed += n - n2;
		eng->set_counters(c0);
		std::vector<uint8_t>& res = keep_.res;
		if (n2) {
			keep_.woff.push_back(keep_.used); // (one past the last: the store's end)
			void* woff_d = be.alloc((n2 + 1) * 8);
			void* len_d = be.alloc(n2 * 4);
			be.h2d(woff_d, keep_.woff.data(), (n2 + 1) * 8);
			be.h2d(len_d, keep_.len.data(), n2 * 4);
			keep_.woff.pop_back();
			Batch b{ (const uint32_t*)keep_.words, (const uint64_t*)woff_d, (const uint32_t*)len_d, (const uint64_t*)woff_d /* unused in pass 2 */, n2 };
			std::vector<uint8_t> pres(n2);
			std::function<void(const ContigOut&)> sink;
			if (cb) sink = [&](const ContigOut& o) {
				abg_contig c;
				c.contig_id = o.contig_id; c.read_index = keep_.orig[o.read_index];
				c.seq = o.seq.c_str(); c.length = (uint32_t)o.seq.size(); c.coverage = o.coverage;
				c.redundant = o.redundant; c.left_ext = o.left_ext; c.right_ext = o.right_ext;
				c.left_code = o.left_code; c.right_code = o.right_code; c.seed_pos = o.seed_pos;
				cb(user, &c);
			};
			const bool timing = getenv("ABG_HOST_TIMING") != nullptr;
			const auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
			const double ta = tnow();
			eng->assemble_packed(b, pres.data(), sink);
			const double tb = tnow();
			be.free(woff_d); be.free(len_d);
			for (uint64_t j = 0; j < n2; j++) res[keep_.orig[j]] = pres[j];
			if (timing) fprintf(stderr, "[host] assemble (kept reads): set-up and upload %.3f s, device passes + callbacks %.3f s, verdicts %.3f s\n", ta - t_in, tb - ta, tnow() - tb);
		}
		if (results && n) memcpy(results, res.data(), n);
		keep_drop(false);
		keep_on_ = false;
		return ABG_OK;
	}
  private:
	// (on / failed live outside Keep: a load call reads them on the caller's thread while the previous call's
	// device share -- which may drop the store -- still runs on the library's)
	std::atomic<bool> keep_on_{ false }, keep_failed_{ false };
	struct Keep {
		bool unordered = false;
		void* words = nullptr; uint64_t cap = 0, used = 0, hint_words = 0; // device store of 2-bit words
		std::vector<uint64_t> woff, orig; std::vector<uint32_t> len;      // per kept (clean) read: where, which read
		std::vector<uint8_t> res;                                         // per read loaded: verdict known at loading time, or 0
		uint64_t n_reads = 0;
	} keep_;
	bool keep_room(uint64_t more)
	{
		if (keep_.used && getenv("ABG_KEEP_FAIL")) return false; // (tests: no room for anything after the first call's reads)
		if (keep_.used + more <= keep_.cap) return true;
		uint64_t cap = std::max<uint64_t>({ keep_.hint_words, keep_.cap * 2, keep_.used + more, 1ull << 20 });
		void* w = be.try_alloc(cap * 4);
		if (!w && cap > keep_.used + more) { cap = keep_.used + more; w = be.try_alloc(cap * 4); }
		if (!w) return false;
		if (keep_.used) be.d2d(w, keep_.words, keep_.used * 4);
		be.sync();
		if (keep_.words) be.free(keep_.words);
		keep_.words = w; keep_.cap = cap;
		return true;
	}
	void keep_drop(bool failed)
	{
		if (keep_.words) { be.sync(); be.free(keep_.words); }
		const uint64_t hint = keep_.hint_words;
		keep_ = Keep();
		keep_failed_ = failed; keep_.hint_words = hint;
	}
  public:
	int load_packed(const uint32_t* d_words, const uint64_t* d_woff, const uint32_t* d_len, uint64_t n)
	{
		drain();
		if (!n) return ABG_OK;
		// (the k-mer prefix sums and the batches' op ranges are made on the device: the reads are there,
		// and a host loop over tens of millions of lengths per call is time the device would idle)
		uint64_t* koff_d = (uint64_t*)be.alloc((n + 1) * 8);
		const bool any_short = eng->device_koff(d_len, n, koff_d);
		if (any_short) { be.free(koff_d); return fail(ABG_EINVAL, "packed sequence shorter than k"); }
		Batch b{ d_words, d_woff, d_len, koff_d, n };
		eng->load_packed(b);
		be.free(koff_d);
		return ABG_OK;
	}

	// ------------------------------------------------------------------ partitioned run
	int attach_comm(const abg_comm& c)
	{
		drain();
		if (!c.all_gather_v || !c.all_reduce) return fail(ABG_EINVAL, "communicator lacks a collective");
		typename Engine<BE>::Comm ec;
		ec.rank = c.rank; ec.world = c.world; ec.stream_ordered = c.stream_ordered != 0; ec.user = c.user;
		ec.all_gather_v = c.all_gather_v; ec.all_reduce = c.all_reduce;
		// (a caller compiled against the header without all_to_all_v passes a shorter struct: the member is read only when the caller says it is there)
		// struct_size is 0 (a caller of the round-4 header: the field was reserved there) or the size of a struct that reaches at
		// least to all_to_all_v; anything else is a struct this library does not know -- refused, not guessed at
		if (c.struct_size != 0 && (size_t)c.struct_size < offsetof(abg_comm, all_to_all_v))
			return fail(ABG_EINVAL, "abg_comm.struct_size is neither 0 nor the size of a struct this library knows (set it to sizeof(abg_comm))");
		ec.all_to_all_v = (size_t)c.struct_size >= offsetof(abg_comm, all_to_all_v) + sizeof c.all_to_all_v ? c.all_to_all_v : nullptr;
		if (c.struct_size == 0 && cfg.verbose && c.world >= 4)
			fprintf(stderr, "abyss_amd: abg_comm.struct_size is 0 (a caller built against the older header): all_to_all_v is not read, the partitioned PASS 1 runs without routed pairs\n");
		comm_attached_ = true;
		if (!eng->attach_comm(ec)) return fail(ABG_EINVAL, "bad rank / world (at most " + std::to_string(MAX_RANKS) + " ranks; not available on a cascading filter)");
		return ABG_OK;
	}
	int share_reads(const uint32_t* d_words, const uint64_t* d_woff, const uint32_t* d_len, uint64_t n,
	    const uint32_t** g_words, const uint64_t** g_woff, const uint32_t** g_len, uint64_t* n_total)
	{
		Batch loc{ d_words, d_woff, d_len, d_woff, n };
		Batch g = eng->share_reads(loc);
		*g_words = g.words; *g_woff = g.woff; *g_len = g.len; *n_total = g.n;
		return ABG_OK;
	}

	// ------------------------------------------------------------------ PASS 2
	int assemble_seqs(const char* seqs, const uint64_t* off, uint64_t n, uint8_t* results,
	    abg_contig_cb cb, void* user)
	{
		return assemble_seqs_v(1, &seqs, &off, &n, results, cb, user);
	}
	// The same over a read set held in several buffers (the host binary keeps the records of PASS 1 in
	// the chunks it parsed them into): ONE assemble call, so one guide and one batch schedule for all
	// of them.  Read indices (results, abg_contig.read_index) count through the chunks in order.
	int assemble_seqs_v(uint32_t nchunks, const char* const* seqs_v, const uint64_t* const* off_v, const uint64_t* n_v,
	    uint8_t* results, abg_contig_cb cb, void* user)
	{
		if (eng->cascade_mode()) return fail(ABG_EINVAL, "assembly is not available on a cascading filter");
		const uint32_t k = cfg.k;
		uint64_t n = 0;
		std::vector<uint64_t> base(nchunks + 1, 0);
		for (uint32_t c = 0; c < nchunks; c++) { base[c + 1] = base[c] + n_v[c]; }
		n = base[nchunks];
		std::vector<uint64_t> orig; // packed index -> caller index
		std::vector<uint8_t> res(n, (uint8_t)RR_UNINITIALIZED);
		// parts: (chunk, [first, last)) in read order
		struct Part { uint32_t c; uint64_t a, b; };
		std::vector<Part> plan;
		for (uint32_t c = 0; c < nchunks; c++) {
			if (!n_v[c]) continue;
			const std::vector<uint64_t> cut = split_reads(off_v[c], n_v[c]);
			for (size_t t = 0; t + 1 < cut.size(); t++) plan.push_back(Part{ c, cut[t], cut[t + 1] });
		}
		std::vector<HostBatch> parts(plan.size());
		std::vector<std::vector<uint64_t>> origs(parts.size());
		const bool timing = getenv("ABG_HOST_TIMING") != nullptr;
		const auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		const double t0 = tnow();
		run_parts(parts.size(), [&](size_t t) {
			const Part& pt = plan[t];
			const char* seqs = seqs_v[pt.c];
			const uint64_t* off = off_v[pt.c];
			{
				const uint64_t nr = pt.b - pt.a, bases = off[pt.b] - off[pt.a];
				parts[t].words.reserve(bases / 16 + nr + 1); parts[t].woff.reserve(nr + 1); parts[t].len.reserve(nr); parts[t].koff.reserve(nr + 1);
				origs[t].reserve(nr);
			}
			for (uint64_t i = pt.a; i < pt.b; i++) {
				const char* s = seqs + off[i];
				uint64_t L = off[i + 1] - off[i];
				const uint64_t gi = base[pt.c] + i;
				if (L < k) { res[gi] = RR_SHORTER_THAN_K; continue; }  // bloom-dbg.h:804
				uint8_t bad = 0;
				for (uint64_t q = 0; q < L; q++) bad |= codes_.t[(unsigned char)s[q]];
				if (bad & 0x80) { res[gi] = RR_NON_ACGT; continue; }     // allACGT, bloom-dbg.h:808 (either case: the reader folds it)
				parts[t].add_ascii(s, (uint32_t)L, k);
				origs[t].push_back(gi);
			}
		});
		HostBatch joined;
		HostBatch empty;
		const double t1 = tnow();
		const HostBatch& hb = parts.empty() ? empty : join_parts(parts, joined);
		{
			size_t total = 0;
			for (auto& o : origs) total += o.size();
			orig.reserve(total);
			for (auto& o : origs) { orig.insert(orig.end(), o.begin(), o.end()); std::vector<uint64_t>().swap(o); }
		}
		const double t2 = tnow();
		// the reference counts every read in readsProcessed (bloom-dbg.h:1045), also the
		// ones rejected above; the engine counts the ones it sees
		Counters c0 = eng->counters();
		c0.reads_processed += n - hb.n();
		eng->set_counters(c0);
		if (hb.n()) {
			DevBatch d = upload(hb);
			const double t3 = tnow();
			std::vector<uint8_t> pres(hb.n());
			std::function<void(const ContigOut&)> sink;
			if (cb) sink = [&](const ContigOut& o) {
				abg_contig c;
				c.contig_id = o.contig_id; c.read_index = orig[o.read_index];
				c.seq = o.seq.c_str(); c.length = (uint32_t)o.seq.size(); c.coverage = o.coverage;
				c.redundant = o.redundant; c.left_ext = o.left_ext; c.right_ext = o.right_ext;
				c.left_code = o.left_code; c.right_code = o.right_code; c.seed_pos = o.seed_pos;
				cb(user, &c);
			};
			eng->assemble_packed(d.b, pres.data(), sink);
			const double t4 = tnow();
			release(d);
			for (uint64_t j = 0; j < hb.n(); j++) res[orig[j]] = pres[j];
			if (timing) fprintf(stderr, "[host] assemble: pack %.3f s, join %.3f s, upload %.3f s, device passes + callbacks %.3f s, results %.3f s\n",
			    t1 - t0, t2 - t1, t3 - t2, t4 - t3, tnow() - t4);
		}
		if (results) memcpy(results, res.data(), n);
		return ABG_OK;
	}
	int assemble_packed(const uint32_t* d_words, const uint64_t* d_woff, const uint32_t* d_len,
	    uint64_t n, uint8_t* results, abg_contig_cb cb, void* user)
	{
		if (eng->cascade_mode()) return fail(ABG_EINVAL, "assembly is not available on a cascading filter");
		if (!n) return ABG_OK;
		Batch b{ d_words, d_woff, d_len, d_woff /* unused in pass 2 */, n };
		std::function<void(const ContigOut&)> sink;
		if (cb) sink = [&](const ContigOut& o) {
			abg_contig c;
			c.contig_id = o.contig_id; c.read_index = o.read_index;
			c.seq = o.seq.c_str(); c.length = (uint32_t)o.seq.size(); c.coverage = o.coverage;
			c.redundant = o.redundant; c.left_ext = o.left_ext; c.right_ext = o.right_ext;
			c.left_code = o.left_code; c.right_code = o.right_code; c.seed_pos = o.seed_pos;
			cb(user, &c);
		};
		eng->assemble_packed(b, results, sink);
		return ABG_OK;
	}

	// ------------------------------------------------------------------ -g
	// outputGraph (bloom-dbg.h:1171-1242) over sequences as FastaReader(FOLD_CASE) hands them over:
	// the GraphViz lines between "digraph g {" and "}", delivered in chunks.  trimSeq and the
	// searches run on the device (FTrimRun, FGraphBfs); the host turns the recorded visiting order
	// back into k-mer strings: a successor's k-mer is its parent's shifted by one base.
	// (Under a spaced seed, a non-ACGT character beneath a '0' of a START k-mer would be printed by
	// the reference as it stands in the read; here it comes out as 'N' -- the one known deviation.)
	int output_graph_seqs(const char* seqs, const uint64_t* off, uint64_t n, abg_text_cb cb, void* user,
	    uint64_t* nodes_out, uint64_t* edges_out)
	{
		if (eng->cascade_mode()) return fail(ABG_EINVAL, "not available on a cascading filter");
		const uint32_t k = cfg.k;
		HostBatch hb;
		struct Seg { uint64_t read; uint64_t start; };
		std::vector<Seg> segs;
		std::string up;
		for (uint64_t i = 0; i < n; i++) {
			const char* s = seqs + off[i];
			const uint64_t L = off[i + 1] - off[i];
			if (L < k) continue; // trimSeq :406-409
			const char* text = s;
			if (!cfg.spaced_seed.empty()) {
				up.assign(s, L);
				for (auto& ch : up) ch = (char)toupper((unsigned char)ch);
				text = up.data();
			}
			valid_runs(text, L, runs_);
			for (auto& run : runs_) { // k-mers run.first .. run.second - 1; a gap between runs ends a match (:422)
				hb.add_ascii(text + run.first, (uint32_t)(run.second - run.first + k - 1), k);
				segs.push_back(Seg{ i, run.first });
			}
		}
		std::vector<uint32_t> bs(segs.size()), bl(segs.size());
		if (!segs.empty()) {
			DevBatch d = upload(hb);
			eng->trim_runs(d.b, bs.data(), bl.data());
			release(d);
		}
		// per read: the first longest run over its segments; start vertices = its first k-mer and the
		// reverse complement of its last k-mer (:1217-1226)
		HostBatch starts;
		std::vector<std::string> start_kmers;
		auto fold = [](char c) { c = (char)toupper((unsigned char)c); return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N'; };
		auto comp = [](char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; };
		for (size_t a = 0; a < segs.size();) {
			size_t b = a;
			uint32_t best = 0; uint64_t best_pos = 0;
			for (; b < segs.size() && segs[b].read == segs[a].read; b++)
				if (bl[b] > best) { best = bl[b]; best_pos = segs[b].start + bs[b]; }
			if (best) {
				const char* s = seqs + off[segs[a].read];
				std::string first(k, 'N'), rc(k, 'N');
				for (uint32_t j = 0; j < k; j++) {
					first[j] = fold(s[best_pos + j]);
					rc[j] = comp(fold(s[best_pos + best - 1 + (k - 1 - j)]));
				}
				starts.add_ascii(first.data(), k, k);
				starts.add_ascii(rc.data(), k, k);
				start_kmers.push_back(first);
				start_kmers.push_back(rc);
			}
			a = b;
		}
		std::vector<uint8_t> ev, used;
		uint64_t edges = 0;
		if (starts.n()) {
			DevBatch d = upload(starts);
			eng->graph_bfs(d.b, ev, used, &edges);
			release(d);
		}
		// replay
		std::string out;
		out.reserve(1 << 20);
		auto flush = [&]() { if (cb && !out.empty()) cb(user, out.data(), out.size()); out.clear(); };
		std::vector<std::string> queue; // discovered vertices in order; `head` walks it
		size_t head = 0;
		uint64_t node = 0;
		static const char BASES[] = "ACGT";
		for (size_t si = 0; si < start_kmers.size(); si++) {
			if (!used[si]) continue;
			out += '\t'; out += start_kmers[si]; out += ";\n";
			queue.clear(); head = 0;
			queue.push_back(start_kmers[si]);
			while (head < queue.size()) {
				const std::string u = queue[head++];
				if (node >= ev.size()) return fail(ABG_EINTERNAL, "graph replay ran past the device's record");
				const unsigned e = ev[node++];
				for (unsigned b2 = 0; b2 < 4; b2++) {
					if (!((e >> b2) & 1)) continue;
					std::string v = u.substr(1);
					v += BASES[b2];
					out += '\t'; out += u; out += " -> "; out += v; out += ";\n";
					if ((e >> (4 + b2)) & 1) {
						out += '\t'; out += v; out += ";\n";
						queue.push_back(std::move(v));
					}
				}
				if (out.size() >= (1u << 20)) flush();
				if (head >= (1u << 16) && head * 2 > queue.size()) { queue.erase(queue.begin(), queue.begin() + head); head = 0; }
			}
		}
		if (node != ev.size()) return fail(ABG_EINTERNAL, "graph replay does not match the device's record");
		flush();
		if (nodes_out) *nodes_out = ev.size();
		if (edges_out) *edges_out = edges;
		return ABG_OK;
	}

	// ------------------------------------------------------------------ probes
	// valid k-mers of one sequence: positions and whether the solid filter contains them
	// (writeCovTrack's loop, bloom-dbg.h:1297-1312)
	int contains_seq(const char* seq, uint64_t len, uint32_t* pos_out, uint8_t* contains_out,
	    uint64_t cap, uint64_t* n_out)
	{
		if (eng->cascade_mode()) return fail(ABG_EINVAL, "not available on a cascading filter");
		const uint32_t k = cfg.k;
		HostBatch hb;
		std::vector<uint32_t> start;
		std::string up(seq, len);
		for (auto& ch : up) ch = (char)toupper((unsigned char)ch);
		valid_runs(up.data(), up.size(), runs_);
		const uint32_t max_piece = 1u << 20;
		for (auto& run : runs_) {
			uint64_t pa = run.first, pb = run.second - 1 + k;
			for (uint64_t q = pa; q + k <= pb;) {
				uint64_t e = std::min<uint64_t>(pb, q + max_piece);
				hb.add_ascii(up.data() + q, (uint32_t)(e - q), k);
				start.push_back((uint32_t)q);
				if (e == pb) break;
				q = e - (k - 1);
			}
		}
		uint64_t T = hb.koff.back();
		*n_out = T;
		if (!T) return ABG_OK;
		DevBatch d = upload(hb);
		uint64_t* h0 = (uint64_t*)be.alloc(T * 8);
		uint8_t* c = (uint8_t*)be.alloc(T);
		FHash fh{ eng->params(), d.b, h0 };
		be.launch(T, fh, "hash");
		FContainsSolid fc{ eng->params(), h0, eng->counters_dev(), c };
		be.launch(T, fc, "contains");
		std::vector<uint8_t> host(T);
		be.d2h(host.data(), c, T);
		uint64_t t = 0;
		for (uint64_t s2 = 0; s2 < hb.n(); s2++)
			for (uint32_t j = 0; j + k <= hb.len[s2]; j++, t++) {
				if (t >= cap) continue;
				if (pos_out) pos_out[t] = start[s2] + j;
				if (contains_out) contains_out[t] = host[t];
			}
		be.free(h0); be.free(c);
		release(d);
		return ABG_OK;
	}
	int hash_seq(const char* seq, uint64_t len, uint32_t* pos_out, uint64_t* hashes_out,
	    uint64_t cap, uint64_t* n_out)
	{
		const uint32_t k = cfg.k;
		HostBatch hb;
		std::vector<uint32_t> start; // start position of each packed piece
		std::string up(seq, len);
		for (auto& ch : up) ch = (char)toupper((unsigned char)ch);
		valid_runs(up.data(), up.size(), runs_);
		for (auto& run : runs_) {
			hb.add_ascii(up.data() + run.first, (uint32_t)(run.second - run.first + k - 1), k);
			start.push_back((uint32_t)run.first);
		}
		uint64_t T = hb.koff.back();
		*n_out = T;
		if (!T) return ABG_OK;
		DevBatch d = upload(hb);
		uint64_t* h0 = (uint64_t*)be.alloc(T * 8);
		uint64_t* all = (uint64_t*)be.alloc(T * 8 * cfg.nh);
		FHash fh{ eng->params(), d.b, h0 };
		be.launch(T, fh, "hash");
		FExpand fe{ eng->params(), h0, all };
		be.launch(T, fe, "expand");
		std::vector<uint64_t> host(T * cfg.nh);
		be.d2h(host.data(), all, T * 8 * cfg.nh);
		uint64_t t = 0;
		for (uint64_t s = 0; s < hb.n(); s++)
			for (uint32_t j = 0; j + k <= hb.len[s]; j++, t++) {
				if (t >= cap) continue;
				if (pos_out) pos_out[t] = start[s] + j;
				if (hashes_out) memcpy(hashes_out + t * cfg.nh, host.data() + t * cfg.nh, 8 * cfg.nh);
			}
		be.free(h0); be.free(all);
		release(d);
		return ABG_OK;
	}

	int fail(int code, const std::string& msg) { error = msg; return code; }

  private:
	static bool is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
	std::vector<std::pair<uint64_t, uint64_t>> runs_;
	BaseCodes codes_;
	std::vector<uint8_t> bad_;
	// Maximal runs [first, second) of consecutive start positions of the k-mers
	// RollingHashIterator yields for an upper-cased sequence (RollingHashIterator.h:35-97):
	// those with an ACGT character at every position the spaced seed looks at (all of them
	// without one).  A run of k-mers a..b-1 is the text [a, b - 1 + k); characters under a
	// '0' may be anything (they are packed as some base and never contribute to a hash).
	// (without a spaced seed `up` may be in either case; with one it must be upper case)
	void valid_runs(const char* up, uint64_t L, std::vector<std::pair<uint64_t, uint64_t>>& runs) { valid_runs(up, L, runs, bad_); }
	void valid_runs(const char* up, uint64_t L, std::vector<std::pair<uint64_t, uint64_t>>& runs, std::vector<uint8_t>& bad_) const
	{
		runs.clear();
		const uint64_t k = cfg.k;
		if (L < k) return;
		const uint64_t nk = L - k + 1;
		if (cfg.spaced_seed.empty()) {
			const uint8_t* t = codes_.t;
			uint64_t a = 0;
			while (a < L) {
				while (a < L && (t[(unsigned char)up[a]] & 0x80)) a++;
				uint64_t b = a;
				while (b < L && !(t[(unsigned char)up[b]] & 0x80)) b++;
				if (b - a >= k) runs.emplace_back(a, b - k + 1);
				a = b;
			}
			return;
		}
		bad_.assign(nk, 0);
		const std::string& m = cfg.spaced_seed;
		for (uint64_t x = 0; x < L; x++) {
			if (is_acgt(up[x])) continue;
			for (uint64_t i = 0; i < k; i++)
				if (m[i] == '1' && x >= i && x - i < nk) bad_[x - i] = 1;
		}
		for (uint64_t j = 0; j < nk;) {
			if (bad_[j]) { j++; continue; }
			uint64_t e = j;
			while (e < nk && !bad_[e]) e++;
			runs.emplace_back(j, e);
			j = e;
		}
	}
	// ---- host threads (packing): reads [0, n) cut into ranges of about equal size in bases
	static unsigned host_threads()
	{
		if (const char* e = getenv("ABG_HOST_THREADS")) return (unsigned)std::max(1, atoi(e));
		return std::min(48u, std::max(1u, std::thread::hardware_concurrency()));
	}
	static std::vector<uint64_t> split_reads(const uint64_t* off, uint64_t n)
	{
		const uint64_t bases = n ? off[n] - off[0] : 0;
		unsigned T = host_threads();
		if (bases < (8u << 20)) T = 1; // not worth a thread
		if (const char* e = getenv("ABG_HOST_SPLIT")) T = (unsigned)std::max(1, atoi(e)); // tests: split small inputs too
		std::vector<uint64_t> cut{ 0 };
		for (unsigned t = 1; t < T; t++) {
			const uint64_t want = off[0] + bases / T * t;
			uint64_t i = (uint64_t)(std::lower_bound(off, off + n + 1, want) - off);
			if (i > cut.back() && i < n) cut.push_back(i);
		}
		cut.push_back(n);
		return cut;
	}
	template <class F>
	static void run_parts(size_t nparts, F f)
	{
		if (nparts <= 1) { if (nparts) f(0); return; }
		// (at most host_threads() workers, whatever the number of parts)
		const size_t T = std::min<size_t>(nparts, std::max(1u, host_threads()));
		std::atomic<size_t> next{ 0 };
		std::vector<std::thread> pool;
		for (size_t w = 0; w < T; w++)
			pool.emplace_back([&f, &next, nparts]() { for (size_t t; (t = next.fetch_add(1)) < nparts;) f(t); });
		for (auto& th : pool) th.join();
	}
	// the parts' batches one after the other (a single part is used as it is)
	static const HostBatch& join_parts(std::vector<HostBatch>& parts, HostBatch& joined)
	{
		if (parts.size() == 1) return parts[0];
		// where each part goes, then every part copied by a thread of its own
		std::vector<uint64_t> wbase(parts.size() + 1, 0), rbase(parts.size() + 1, 0), kbase(parts.size() + 1, 0);
		for (size_t t = 0; t < parts.size(); t++) {
			wbase[t + 1] = wbase[t] + parts[t].words.size();
			rbase[t + 1] = rbase[t] + parts[t].n();
			kbase[t + 1] = kbase[t] + parts[t].koff.back();
		}
		const uint64_t reads = rbase[parts.size()];
		joined.words.resize(wbase[parts.size()]);
		joined.woff.resize(reads + 1); joined.len.resize(reads); joined.koff.resize(reads + 1);
		joined.woff[0] = 0; joined.koff[0] = 0;
		run_parts(parts.size(), [&](size_t t) {
			HostBatch& p = parts[t];
			if (!p.words.empty()) memcpy(joined.words.data() + wbase[t], p.words.data(), p.words.size() * 4);
			if (p.n()) memcpy(joined.len.data() + rbase[t], p.len.data(), p.n() * 4);
			for (size_t i = 1; i < p.woff.size(); i++) joined.woff[rbase[t] + i] = p.woff[i] + wbase[t];
			for (size_t i = 1; i < p.koff.size(); i++) joined.koff[rbase[t] + i] = p.koff[i] + kbase[t];
			p = HostBatch(); // give the memory back as we go
		});
		return joined;
	}
	struct DevBatch { Batch b; void* words; void* woff; void* len; void* koff; };
	DevBatch upload(const HostBatch& hb)
	{
		DevBatch d;
		d.words = be.alloc(std::max<size_t>(hb.words.size(), 1) * 4 + 64);
		d.woff = be.alloc(hb.woff.size() * 8);
		d.len = be.alloc(std::max<size_t>(hb.len.size(), 1) * 4);
		d.koff = be.alloc(hb.koff.size() * 8);
		be.h2d(d.words, hb.words.data(), hb.words.size() * 4);
		be.h2d(d.woff, hb.woff.data(), hb.woff.size() * 8);
		be.h2d(d.len, hb.len.data(), hb.len.size() * 4);
		be.h2d(d.koff, hb.koff.data(), hb.koff.size() * 8);
		d.b = Batch{ (const uint32_t*)d.words, (const uint64_t*)d.woff, (const uint32_t*)d.len,
			(const uint64_t*)d.koff, hb.n() };
		return d;
	}
	void release(DevBatch& d) { be.free(d.words); be.free(d.woff); be.free(d.len); be.free(d.koff); }
	void flush_load(const HostBatch& hb)
	{
		DevBatch d = upload(hb);
		eng->load_packed(d.b, hb.koff.data());
		release(d);
	}
};

} // namespace abg
