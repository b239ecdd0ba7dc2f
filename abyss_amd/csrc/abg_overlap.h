// abg_overlap.h -- the step after the unitig stage: AdjList's join of contig ends that overlap
// by exactly k-1 bases.
//
// Reference behaviour restated here (ABySS 2.3.10):
//   AdjList/AdjList.cpp:192-230  readContigs: per contig i the vertices 2i (i+) and 2i+1 (i-);
//                                prefixes[2i] = first k-1 bases, prefixes[2i+1] = rc(last k-1 bases);
//                                suffixMap[last k-1 bases] gets 2i, suffixMap[rc(first k-1 bases)] gets 2i+1
//   AdjList/AdjList.cpp:233-263  buildOverlapGraph: for v = 0..2n-1, for u in suffixMap[prefixes[v]] in
//                                insertion order (ascending u): edge (v^1) -> (u^1), skipped under --SS
//                                when the two senses differ
// The unordered_map of the reference becomes a sort: every vertex's suffix key is hashed, the
// (hash, vertex) pairs are radix-sorted (stable, so equal keys stay in ascending vertex order), and
// every vertex looks its prefix key up by binary search, comparing whole keys on a hash match.  The
// edges come out as a CSR over the SOURCE vertex, each list in the reference's order.
//
// Like abg_core.h, everything here is ABG_HD and written against the backend interface, so
// tests/hostcheck runs the same code serially; the product runs it on the GPU only.
#pragma once
#include <vector>

#include "abg_core.h"

namespace abg {

constexpr uint32_t OV_MAX_WORDS = 8; // a key is (k-1) <= 256 bases, 2 bits each, base j at bits 2(j%32) of word j/32

ABG_HD uint32_t ov_base(const uint64_t* w, uint32_t j) { return (uint32_t)(w[j >> 5] >> (2 * (j & 31))) & 3u; }
ABG_HD uint64_t ov_mix(uint64_t h, uint64_t x)
{
	h ^= x;
	h *= 0xff51afd7ed558ccdULL;
	h ^= h >> 33;
	h *= 0xc4ceb9fe1a85ec53ULL;
	h ^= h >> 29;
	return h;
}

// One item per contig: the prefix and suffix keys of its two vertices and their hashes.
//   vertex 2i   (i+): prefix key = head,      suffix key = tail
//   vertex 2i+1 (i-): prefix key = rc(tail),  suffix key = rc(head)
struct FOverlapKeys {
	const uint64_t* head; const uint64_t* tail; // [n][W] as handed over by the caller
	uint32_t km1, W;
	uint64_t* pk; uint64_t* sk;                 // [2n][W]
	uint64_t* hp; uint64_t* hs; uint32_t* id;   // [2n]
	ABG_HD void operator()(uint64_t i, uint32_t) const
	{
		const uint64_t* H = head + i * W;
		const uint64_t* T = tail + i * W;
		uint64_t hh = km1, ht = km1, hrh = km1, hrt = km1;
		for (uint32_t w = 0; w < W; w++) {
			const uint32_t nb = km1 - 32 * w < 32 ? km1 - 32 * w : 32;
			const uint64_t mask = nb == 32 ? ~0ull : (1ull << (2 * nb)) - 1;
			uint64_t rh = 0, rt = 0;
			for (uint32_t b = 0; b < nb; b++) {
				const uint32_t j = km1 - 1 - (32 * w + b);
				rh |= (uint64_t)(3u - ov_base(H, j)) << (2 * b);
				rt |= (uint64_t)(3u - ov_base(T, j)) << (2 * b);
			}
			const uint64_t h = H[w] & mask, t = T[w] & mask;
			pk[(2 * i) * W + w] = h;
			sk[(2 * i) * W + w] = t;
			pk[(2 * i + 1) * W + w] = rt;
			sk[(2 * i + 1) * W + w] = rh;
			hh = ov_mix(hh, h); ht = ov_mix(ht, t); hrh = ov_mix(hrh, rh); hrt = ov_mix(hrt, rt);
		}
		hp[2 * i] = hh; hs[2 * i] = ht;
		hp[2 * i + 1] = hrt; hs[2 * i + 1] = hrh;
		id[2 * i] = (uint32_t)(2 * i);
		id[2 * i + 1] = (uint32_t)(2 * i + 1);
	}
};

struct OverlapEnv {
	const uint64_t* pk; const uint64_t* sk; const uint64_t* hp;
	const uint64_t* hs_sorted; const uint32_t* id_sorted; // suffix hashes in ascending order and whose they are
	uint64_t nv; uint32_t W; int ss;
};
// the vertices u whose suffix key equals the prefix key of v, in ascending order (AdjList.cpp:247-255)
template <class E>
ABG_HD void overlap_matches(const OverlapEnv& e, uint64_t v, E&& emit)
{
	const uint64_t h = e.hp[v];
	uint64_t lo = 0, hi = e.nv;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (e.hs_sorted[mid] < h) lo = mid + 1; else hi = mid;
	}
	for (uint64_t r = lo; r < e.nv && e.hs_sorted[r] == h; r++) {
		const uint32_t u = e.id_sorted[r];
		if (e.ss && ((u ^ (uint32_t)v) & 1u)) continue; // uc.sense() != vc.sense()
		bool eq = true;
		for (uint32_t w = 0; w < e.W; w++) eq &= e.sk[(uint64_t)u * e.W + w] == e.pk[v * e.W + w];
		if (eq) emit(u);
	}
}
struct FOverlapCount { // off[s + 1] = out-degree of s = v^1 (an inclusive scan turns them into offsets)
	OverlapEnv e; uint64_t* off;
	ABG_HD void operator()(uint64_t v, uint32_t) const
	{
		uint64_t n = 0;
		overlap_matches(e, v, [&](uint32_t) { n++; });
		off[(v ^ 1) + 1] = n;
	}
};
struct FOverlapFill {
	OverlapEnv e; const uint64_t* off; uint32_t* tgt;
	ABG_HD void operator()(uint64_t v, uint32_t) const
	{
		uint64_t at = off[v ^ 1];
		overlap_matches(e, v, [&](uint32_t u) { tgt[at++] = u ^ 1u; });
	}
};

// Host driver over a backend BE (HipBackend in the product, SerialBackend in tests/hostcheck).
template <class BE>
class OverlapJoin {
  public:
	explicit OverlapJoin(BE& be) : be_(be) {}
	~OverlapJoin() { drop(); }
	// head/tail: HOST arrays [n][W], W = ceil(km1 / 32)
	void run(uint32_t km1, uint64_t n, const uint64_t* head, const uint64_t* tail, bool ss)
	{
		drop();
		const uint32_t W = (km1 + 31) / 32;
		nv_ = 2 * n;
		off_ = (uint64_t*)be_.alloc((nv_ + 1) * 8);
		be_.memset(off_, 0, (nv_ + 1) * 8);
		if (!n) { ne_ = 0; return; }
		const size_t kb = (size_t)n * W * 8;
		// (temporaries of this call: given back whichever way it ends -- an allocation or a launch that fails throws)
		struct Temps {
			BE& be; std::vector<void*> v;
			void* get(size_t bytes) { void* p = be.alloc(bytes); v.push_back(p); return p; }
			~Temps() { for (void* p : v) be.free(p); }
		} tmp{ be_, {} };
		uint64_t* dh = (uint64_t*)tmp.get(kb);
		uint64_t* dt = (uint64_t*)tmp.get(kb);
		be_.h2d(dh, head, kb);
		be_.h2d(dt, tail, kb);
		uint64_t* pk = (uint64_t*)tmp.get(2 * kb);
		uint64_t* sk = (uint64_t*)tmp.get(2 * kb);
		uint64_t* hp = (uint64_t*)tmp.get(nv_ * 8);
		uint64_t* hs = (uint64_t*)tmp.get(nv_ * 8);
		uint64_t* hs2 = (uint64_t*)tmp.get(nv_ * 8);
		uint32_t* id = (uint32_t*)tmp.get(nv_ * 4);
		uint32_t* id2 = (uint32_t*)tmp.get(nv_ * 4);
		be_.launch(n, FOverlapKeys{ dh, dt, km1, W, pk, sk, hp, hs, id }, "overlap_keys");
		be_.sort_pairs_u64_u32(hs, hs2, id, id2, nv_);
		const OverlapEnv e{ pk, sk, hp, hs2, id2, nv_, W, ss ? 1 : 0 };
		be_.launch(nv_, FOverlapCount{ e, off_ }, "overlap_count");
		be_.inclusive_sum_u64(off_ + 1, nv_);
		be_.d2h(&ne_, off_ + nv_, 8);
		tgt_ = (uint32_t*)be_.alloc(ne_ ? ne_ * 4 : 4);
		be_.launch(nv_, FOverlapFill{ e, off_, tgt_ }, "overlap_fill");
		be_.sync();
	}
	uint64_t vertices() const { return nv_; }
	uint64_t edges() const { return ne_; }
	// offsets [2n+1] and targets [edges()] to HOST arrays
	void fetch(uint64_t* off, uint32_t* tgt)
	{
		if (off && off_) be_.d2h(off, off_, (nv_ + 1) * 8);
		if (tgt && ne_) be_.d2h(tgt, tgt_, ne_ * 4);
	}
  private:
	void drop()
	{
		if (off_) be_.free(off_);
		if (tgt_) be_.free(tgt_);
		off_ = nullptr; tgt_ = nullptr; nv_ = ne_ = 0;
	}
	BE& be_;
	uint64_t nv_ = 0, ne_ = 0;
	uint64_t* off_ = nullptr;
	uint32_t* tgt_ = nullptr;
};

} // namespace abg
