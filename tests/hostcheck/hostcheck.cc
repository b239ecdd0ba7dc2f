// hostcheck.cc -- TEST INFRASTRUCTURE ONLY.
// Runs the product's device logic (abyss_amd/csrc/abg_core.h, abg_walk.h, abg_engine.h,
// abg_host.h) on the CPU, one item at a time, through a serial backend, so that the
// "-m 'not gpu'" test-suite can compare the kernels' logic with the oracle on machines
// without a GPU.  It is built only by tests/ (g++), never linked into libabyss_amd.so,
// and is not a fallback: the product library refuses to run without a HIP device.
#include "../../abyss_amd/csrc/abg_host.h"
#include "../../abyss_amd/csrc/abg_overlap.h"

#include <cstdlib>
#include <cstring>
#include <sys/mman.h>

namespace {

struct SerialSync {
	uint32_t tid() const { return 0; }
	uint32_t nthreads() const { return 1; }
	void barrier() {}
	bool all(bool v) { return v; }
	uint32_t sum(uint32_t v) { return v; }
	uint32_t bcast(uint32_t v) { return v; }
	bool any(bool v) { return v; }
	void sort_u32(uint32_t* keys, uint32_t n) { std::sort(keys, keys + n); }
	abg::CommitDesc buf[abg::COMMIT_CHUNK];
	abg::CommitDesc* descs() { return buf; }
};

struct SerialBackend {
	static constexpr uint32_t ITEM_GROUP_LOG2 = 0; // launch() runs its items one at a time
	bool ok() const { return true; }
	std::string why() const { return ""; }
	void bind_thread() {}
	void* alloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) abort(); return p; }
	void* try_alloc(size_t n) { return malloc(n ? n : 1); }
	void free(void* p) { ::free(p); }
	// The window of a sliced filter: the whole array is reserved as addresses nothing may touch and only the pages of
	// [lo, lo + span) are given memory -- a kernel that reads or writes another rank's counters dies on the spot.
	void* alloc_window(uint64_t total, uint64_t lo, uint64_t span)
	{
		const uint64_t pg = 4096, len = (total + pg - 1) / pg * pg + pg;
		char* base = (char*)mmap(nullptr, len, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (base == (char*)MAP_FAILED) abort();
		const uint64_t a = lo / pg * pg, b = std::min<uint64_t>(len, (lo + span + pg - 1) / pg * pg);
		if (mprotect(base + a, b - a, PROT_READ | PROT_WRITE)) abort();
		return base;
	}
	void free_window(void* base, uint64_t total, uint64_t, uint64_t) { const uint64_t pg = 4096; munmap(base, (total + pg - 1) / pg * pg + pg); }
	void memset(void* p, int v, size_t n) { ::memset(p, v, n); }
	void side_scope_begin(const char*) {} // (serial: everything runs at once, in call order)
	void side_scope_end() {}
	void wait_side_scope() {}
	void h2d(void* d, const void* s, size_t n) { memcpy(d, s, n); }
	// (the serial backend has one thread's worth of everything: two malloc'ed blocks stand in for the device's)
	std::vector<char> ahead[2]; int ahead_next = 0;
	void* upload_ahead(const void* const* src, const size_t* bytes, int n, void** at)
	{
		if (getenv("ABG_NO_UPLOAD_AHEAD")) return nullptr;
		size_t total = 0;
		for (int i = 0; i < n; i++) total += (bytes[i] + 255) & ~(size_t)255;
		std::vector<char>& a = ahead[ahead_next];
		ahead_next ^= 1;
		if (a.size() < total + 1) a.resize(total + 1);
		size_t off = 0;
		for (int i = 0; i < n; i++) { at[i] = a.data() + off; memcpy(at[i], src[i], bytes[i]); off += (bytes[i] + 255) & ~(size_t)255; }
		return a.data();
	}
	void d2h(void* d, const void* s, size_t n) { memcpy(d, s, n); }
	void inclusive_sum_u64(uint64_t* d, uint64_t n) { for (uint64_t i = 1; i < n; i++) d[i] += d[i - 1]; }
	void sort_pairs_u64_u32(const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n)
	{
		std::vector<uint64_t> o(n);
		for (uint64_t i = 0; i < n; i++) o[i] = i;
		std::stable_sort(o.begin(), o.end(), [&](uint64_t a, uint64_t b) { return kin[a] < kin[b]; });
		for (uint64_t i = 0; i < n; i++) { kout[i] = kin[o[i]]; vout[i] = vin[o[i]]; }
	}
	uint32_t max_slots() const { return 1; }
	uint64_t device_mem_bytes() const { return 0; } // (unknown: no cap)
	void d2d(void* d, const void* s, size_t n) { memmove(d, s, n); }
	void sync() {}
	void begin(const char*) {}
	void end(const char*) {}
	void* stream_handle() const { return nullptr; }
	void compact_flagged(const uint32_t* in, const uint8_t* flags, uint64_t n, uint32_t* out, uint32_t* count)
	{
		uint32_t m = 0;
		for (uint64_t i = 0; i < n; i++) if (flags[i] == 1) out[m++] = in ? in[i] : (uint32_t)i; // (1 exactly: abg::PEND_ROUNDS)
		*count = m;
	}
	template <class F> void launch(uint64_t n, F f, const char*) { for (uint64_t i = 0; i < n; i++) f(i, 0); }
	template <class F> void launch_slots(uint64_t n, F f, uint32_t, const char*) { for (uint64_t i = 0; i < n; i++) f(i, 0); }
	template <class F> void launch_wave(uint64_t n, F f, const char*) { for (uint64_t i = 0; i < n; i++) f(i, 0, 1); }
	template <class F> void launch_slots_side(uint64_t n, F f, uint32_t slots, const char* name) { launch_slots(n, f, slots, name); }
	void sync_side() {}
	bool side_done() { return true; }
	// a small "fast" buffer so that the two-tier trueBranch stack crosses tiers in the tests;
	// HC_FAST_BYTES=16384 gives the walkers what the device gives them (the chain searches and the
	// bulk scratch then live in it as they do in LDS)
	alignas(16) unsigned char fastbuf[20480];
	uint32_t fast_bytes = 2048;
	SerialBackend() { if (const char* e = getenv("HC_FAST_BYTES")) fast_bytes = (uint32_t)std::min<long>(sizeof fastbuf, std::max<long>(1024, atol(e))); }
	template <class F> void launch_walkers(uint64_t n, F f, uint32_t, const char*)
	{
		if (fast_bytes >= sizeof(abg::MaskCache)) memset(fastbuf + fast_bytes - sizeof(abg::MaskCache), 0, sizeof(abg::MaskCache)); // (as k_walkers does)
		for (uint64_t i = 0; i < n; i++) f(i, 0, (void*)fastbuf, fast_bytes, false);
	}
	void launch_drain(abg::InsertDrainEnv e) { SerialSync sy; abg::insert_drain(e, sy); }
	std::vector<unsigned char> tilebuf;
	template <class F> void launch_tiles(uint64_t n, F f, const char*)
	{
		tilebuf.resize(F::FAST + 16);
		SerialSync sy;
		for (uint64_t i = 0; i < n; i++) f(i, (void*)tilebuf.data(), sy);
	}
	template <int NW> void launch_commit(abg::CommitEnv<NW> e, uint32_t b, uint32_t c)
	{
		SerialSync sy;
		abg::commit_candidates<NW>(e, b, c, sy);
	}
};

typedef abg::Session<SerialBackend> Sess;
// (every entry waits for the device stage of the last load call first, as the C ABI does: Session::load_seqs_v)
static Sess* S(void* h) { Sess* s = (Sess*)h; s->drain(); return s; }

// k-mer helpers of the bulk steps against the per-base forms: window_kmer vs batch_kmer,
// kmer_revcomp_fast vs kmer_revcomp, kmer_hashes vs vtx_rehash.  `words` holds one packed sequence
// of `len` bases.  Returns the number of mismatches.
template <int NW>
static uint64_t selftest_kmer_nw(unsigned k, const uint32_t* words, uint32_t len)
{
	abg::Params p = abg::make_params(k, 4, 2, k, 1024);
	const uint64_t woff[2] = { 0, (len + 15) / 16 };
	const uint64_t koff[2] = { 0, len - k + 1 };
	abg::Batch b{ words, woff, &len, koff, 1 };
	uint64_t bad = 0;
	for (uint32_t j = 0; j + k <= len; j++) {
		abg::Vtx<NW> v;
		// (base by base: the product's batch_kmer is window_kmer these days, so the per-base form lives here as its check)
		for (int q = 0; q < abg::KW<NW>; q++) v.s.w[q] = 0;
		for (unsigned i = 0; i < k; i++) abg::kmer_set(v.s, i, abg::batch_base(b, 0, j + i));
		{
			const abg::Kmer<NW> viaBatch = abg::batch_kmer<NW>(b, 0, j, k);
			for (int q = 0; q < abg::KW<NW>; q++) bad += viaBatch.w[q] != v.s.w[q];
		}
		abg::vtx_rehash(p, v);
		const abg::Kmer<NW> w = abg::window_kmer<NW>(words, 0, j, k);
		uint64_t fh, rh;
		abg::kmer_hashes(w, k, fh, rh);
		const abg::Kmer<NW> r0 = abg::kmer_revcomp(v.s, k), r1 = abg::kmer_revcomp_fast(v.s, k);
		for (int q = 0; q < abg::KW<NW>; q++) bad += (w.w[q] != v.s.w[q]) + (r0.w[q] != r1.w[q]);
		bad += (fh != v.fh) + (rh != v.rh);
	}
	// the rotated seeds worked out on the fly (SeedTabs) against the tables make_params holds
	{
		const abg::SeedTabs t = abg::seed_tabs(p);
		for (unsigned bb = 0; bb < 4; bb++)
			bad += (t.sk(bb) != p.seed_k[bb]) + (t.rk(bb) != p.seedrc_k[bb]) + (t.sm(bb) != p.seed_km1[bb]) + (t.rm(bb) != p.seedrc_km1[bb]);
	}
	// the prefix-XOR form of the same hashes over stretches of consecutive k-mers (stretch_hashes_serial: the
	// arithmetic of the device's stretch_hashes_wave), from every start and for several lengths
	for (uint32_t q = 0; q + k <= len; q += 3) {
		for (uint32_t n : { 1u, 2u, 33u, 64u }) {
			if (q + n + k - 1 > len) continue;
			uint64_t fh[64], rh[64];
			abg::stretch_hashes_serial<NW>(words, 0, q, n, k, fh, rh);
			for (uint32_t j = 0; j < n; j++) {
				uint64_t f0, r0;
				abg::kmer_hashes(abg::window_kmer<NW>(words, 0, q + j, k), k, f0, r0);
				bad += (fh[j] != f0) + (rh[j] != r0);
			}
		}
	}
	return bad;
}

} // namespace

extern "C" {

void* hc_create(unsigned k, unsigned nh, unsigned kc, unsigned trim, uint64_t counters,
    uint64_t insert_batch, unsigned claim_log2, uint64_t p2_first_batch, const char* spaced_seed)
{
	Sess* s = new Sess();
	abg_params p;
	memset(&p, 0, sizeof p);
	p.spaced_seed = spaced_seed;
	p.k = k; p.num_hashes = nh; p.min_cov = kc; p.trim = trim; p.counters = counters;
	s->cfg.claim_log2 = claim_log2 ? claim_log2 : 16;
	s->cfg.insert_batch_kmers = insert_batch ? insert_batch : (1u << 16);
	s->cfg.walk_slots = 1; s->cfg.tb_cap = 4096; s->cfg.buf_cap = 1u << 20;
	s->cfg.pool_cap = 1ull << 26; s->cfg.rec_cap = 1u << 18; s->cfg.wtab_log2 = 22;
	s->cfg.cend_log2 = 16; s->cfg.drain_threshold = 64; s->cfg.buf_cap = 1u << 20;
	if (p2_first_batch) s->cfg.p2_first_batch = p2_first_batch;
	p.insert_batch_kmers = 0; p.claim_log2 = 0; p.walk_slots = 0; p.wtab_log2 = 0;
	if (s->create(p) != ABG_OK) { fprintf(stderr, "hostcheck: %s\n", s->error.c_str()); delete s; return nullptr; }
	return s;
}
void* hc_create_cascade(unsigned k, unsigned nh, unsigned levels, uint64_t level_bits, uint64_t insert_batch, unsigned claim_log2)
{
	Sess* s = new Sess();
	abg_params p;
	memset(&p, 0, sizeof p);
	p.k = k; p.num_hashes = nh; p.min_cov = 0; p.trim = k; p.counters = level_bits; p.cascade_levels = levels;
	s->cfg.claim_log2 = claim_log2 ? claim_log2 : 14;
	s->cfg.insert_batch_kmers = insert_batch ? insert_batch : (1u << 16);
	s->cfg.drain_threshold = 64;
	if (s->create(p) != ABG_OK) { fprintf(stderr, "hostcheck: %s\n", s->error.c_str()); delete s; return nullptr; }
	return s;
}
uint8_t* hc_cascade_level(void* h, unsigned l) { return S(h)->eng->cascade_level_dev(l); }
void hc_destroy(void* h) { delete (Sess*)h; }
void hc_reset(void* h) { S(h)->eng->reset(); }
uint64_t hc_size(void* h) { return S(h)->eng->size(); }
uint8_t* hc_counters(void* h)
{
	try { return S(h)->eng->counters_dev(); } catch (const abg::Failure& f) { S(h)->error = f.msg; return nullptr; }
}
int hc_counters_export(void* h, uint8_t* out)
{
	try { S(h)->eng->counters_to_host(out); return 0; } catch (const abg::Failure& f) { S(h)->error = f.msg; return f.code; }
}
int hc_counters_import(void* h, const uint8_t* in)
{
	try { S(h)->eng->counters_from_host(in); return 0; } catch (const abg::Failure& f) { S(h)->error = f.msg; return f.code; }
}
const char* hc_last_error(void* h) { return S(h)->error.c_str(); }
uint8_t* hc_visited(void* h) { return S(h)->eng->visited_dev(); }
int hc_load_seqs(void* h, const char* seqs, const uint64_t* off, uint64_t n)
{
	try { return S(h)->load_seqs(seqs, off, n); } catch (const abg::Failure& f) { S(h)->error = f.msg; return f.code; }
}
uint64_t hc_insert_rounds(void* h) { return S(h)->eng->stats().insert_rounds; }
int hc_popcounts(void* h, uint64_t* a, uint64_t* b) { S(h)->eng->popcounts(a, b); return 0; }
int hc_assemble_seqs(void* h, const char* seqs, const uint64_t* off, uint64_t n, uint8_t* results,
    abg_contig_cb cb, void* user)
{
	return S(h)->assemble_seqs(seqs, off, n, results, cb, user);
}
int hc_assemble_seqs_v(void* h, uint32_t nchunks, const char* const* seqs, const uint64_t* const* off, const uint64_t* n,
    uint8_t* results, abg_contig_cb cb, void* user)
{
	return S(h)->assemble_seqs_v(nchunks, seqs, off, n, results, cb, user);
}
int hc_load_seqs_v(void* h, uint32_t nchunks, const char* const* seqs, const uint64_t* const* off, const uint64_t* n)
{
	return S(h)->load_seqs_v(nchunks, seqs, off, n);
}
int hc_keep_reads(void* h, int on, uint64_t expected_bases) { return S(h)->keep_reads(on, expected_bases); }
int hc_assemble_kept(void* h, uint8_t* results, abg_contig_cb cb, void* user) { return S(h)->assemble_kept(results, cb, user); }
int hc_contains_seq(void* h, const char* seq, uint64_t len, uint32_t* pos, uint8_t* val, uint64_t cap, uint64_t* n)
{
	return S(h)->contains_seq(seq, len, pos, val, cap, n);
}
int hc_hash_seq(void* h, const char* seq, uint64_t len, uint32_t* pos, uint64_t* hashes, uint64_t cap, uint64_t* n)
{
	return S(h)->hash_seq(seq, len, pos, hashes, cap, n);
}
// partitioned run (two or more hostcheck sessions, one per process, joined by a communicator)
int hc_attach_comm(void* h, const abg_comm* c) { return S(h)->attach_comm(*c); }
int hc_share_reads(void* h, const uint32_t* w, const uint64_t* woff, const uint32_t* len, uint64_t n,
    const uint32_t** gw, const uint64_t** gwoff, const uint32_t** glen, uint64_t* ntot)
{
	return S(h)->share_reads(w, woff, len, n, gw, gwoff, glen, ntot);
}
int hc_load_packed(void* h, const uint32_t* w, const uint64_t* woff, const uint32_t* len, uint64_t n)
{
	return S(h)->load_packed(w, woff, len, n);
}
int hc_assemble_packed(void* h, const uint32_t* w, const uint64_t* woff, const uint32_t* len, uint64_t n,
    uint8_t* results, abg_contig_cb cb, void* user)
{
	return S(h)->assemble_packed(w, woff, len, n, results, cb, user);
}
int hc_output_graph_seqs(void* h, const char* seqs, const uint64_t* off, uint64_t n, abg_text_cb cb, void* user,
    uint64_t* nodes, uint64_t* edges)
{
	return S(h)->output_graph_seqs(seqs, off, n, cb, user, nodes, edges);
}
void hc_get_counters(void* h, abg_counters* out)
{
	abg::Counters c = S(h)->eng->counters();
	out->solid_reads = c.solid_reads; out->visited_reads = c.visited_reads;
	out->reads_processed = c.reads_processed; out->bases_assembled = c.bases_assembled;
	out->next_contig_id = c.contig_id;
}
void hc_get_stats(void* h, abg_stats* out)
{
	auto s = S(h)->eng->stats();
	out->insert_rounds = s.insert_rounds; out->walk_rounds = s.rounds; out->candidates = s.candidates;
	out->walked = s.walked; out->rewalked = s.rewalked; out->commit_breaks = s.breaks; out->commit_rounds = s.commit_rounds; out->generated = s.generated;
	out->bulk_calls = s.bulk_calls; out->bulk_steps = s.bulk_steps; out->lin_steps = s.lin_steps; out->guide_slots = s.guide_slots; out->chain_steps = s.chain_steps; out->batch_cuts = s.batch_cuts; out->overflows = s.overflows;
	out->memo_hits = s.memo_hits; out->memo_adds = s.memo_adds;
	out->tiled_ops = s.tiled_ops; out->tiled_pending = s.tiled_pending; out->tile_overflows = s.tile_overflows;
	out->cls_covered_reads = s.cls_covered_reads; out->archive_bases = s.archive_bases; out->cls_decided_reads = s.cls_decided_reads;
	out->counter_bytes_held = S(h)->eng->counter_bytes_held();
}
uint64_t hc_selftest_kmer(unsigned k, const uint32_t* words, uint32_t len)
{
	switch ((k + 31) / 32) {
	case 1: return selftest_kmer_nw<1>(k, words, len);
	case 2: return selftest_kmer_nw<2>(k, words, len);
	case 3: return selftest_kmer_nw<3>(k, words, len);
	case 4: return selftest_kmer_nw<4>(k, words, len);
	default: return selftest_kmer_nw<6>(k, words, len);
	}
}
// exact modulo check: returns the number of mismatches between mod64 and the hardware %
uint64_t hc_mod_check(uint64_t m, const uint64_t* hs, uint64_t n)
{
	abg::Mod64 d = abg::make_mod64(m);
	uint64_t bad = 0;
	for (uint64_t i = 0; i < n; i++) bad += abg::mod64(d, hs[i]) != hs[i] % m;
	return bad;
}

// AdjList's k-1 overlap join (abg_overlap.h) through the serial backend; off: 2n+1 entries, tgt: *ne on return
// (call with tgt == NULL first to learn *ne)
int hc_overlap_join(uint32_t overlap, uint64_t n, const uint64_t* head, const uint64_t* tail, int ss, uint64_t* off, uint32_t* tgt, uint64_t* ne)
{
	SerialBackend be;
	abg::OverlapJoin<SerialBackend> j(be);
	j.run(overlap, n, head, tail, ss != 0);
	*ne = j.edges();
	j.fetch(off, tgt);
	return 0;
}

} // extern "C"
