// abg_rr.h -- the read filter of the stage after AdjList: abyss-rresolver-short (SURVEY.md section 8 f4,
// bin/abyss-pe:581-585).  RResolver streams every read once per r value, hashes the r-mers of its first
// r + extract - 1 bases into a plain Bloom filter (7 hash functions) and later asks, for a few thousand candidate
// path sequences, how many of their r-mers the filter holds.
//
// Reference behaviour restated here (ABySS 2.3.10 over btllib, which is NOT under /root/reference -- configure.ac:268-276
// asks for an installed copy; its published algorithm is what oracle/shim/btllib restates and what this file follows):
//   RResolver/BloomFilters.h:12            HASH_NUM = 7
//   RResolver/BloomFilters.cpp:168-197     loadReads: insert(seq.substr(0, r + extract - 1)) for the reads of the current size
//   RResolver/BloomFilters.cpp:246         KmerBloomFilter(bytes, HASH_NUM, r)
//   RResolver/RAlgorithmsShort.cpp:310-366 testSequence: found = contains(sequence), tests = size - r + 1
//   btllib NtHash (ntHash2)                per-base seeds and the split rotation of vendor/nthash/nthash.hpp:18-64,186-217;
//                                          canonical value = forward + reverse strand hash; extra hashes
//                                          h_i = h_0 * (i ^ k * MULTISEED), h_i ^= h_i >> MULTISHIFT; r-mers holding a
//                                          character other than ACGT (either case) are skipped
//   btllib BloomFilter                     `bytes` rounded up to a multiple of 8; bit (h % bits) % 8 of byte (h % bits) / 8
//
// Like abg_core.h everything is ABG_HD: the kernels of abg_rr.hip are thin wrappers, and tests/hostcheck runs the same
// code serially for the CPU suite.  The product runs it on the GPU only.
#pragma once
#include "abg_core.h"

namespace abg {

constexpr uint32_t RR_MAX_HASHES = 16;
constexpr uint32_t RR_MAX_SPAN = 4096; // longest read prefix a record of the insert kernel holds

struct RRParams {
	uint32_t r = 0, hash_num = 0;
	uint64_t bits = 0;            // 8 * bytes, bytes a multiple of 8
	Mod64 mod;                    // ... as a divisor
	uint64_t out_f[4];            // srol^r(seed(c)): what a base leaving the window takes out of the forward hash
	uint64_t in_r[4];             // srol^r(seed(3 - c)): what a base entering the window puts into the reverse hash
	uint64_t mult[RR_MAX_HASHES]; // i ^ r * MULTISEED
};

inline RRParams make_rr_params(uint32_t r, uint32_t hash_num, uint64_t bytes)
{
	RRParams p;
	p.r = r;
	p.hash_num = hash_num;
	p.bits = bytes * 8;
	p.mod = make_mod64(p.bits);
	for (unsigned c = 0; c < 4; c++) {
		p.out_f[c] = srol_n(seed_of(c), r);
		p.in_r[c] = srol_n(seed_of(3 - c), r);
	}
	for (unsigned i = 0; i < RR_MAX_HASHES; i++) p.mult[i] = (uint64_t)i ^ ((uint64_t)r * MULTISEED);
	return p;
}

// A C G T of either case -> 0..3, anything else -> -1
ABG_HD int rr_code(unsigned c)
{
	const unsigned u = c & 0xDFu;
	const unsigned x = (u >> 1) & 3u; // A 0, C 1, T 2, G 3
	const bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
	return ok ? (int)(x ^ (x >> 1)) : -1;
}

// The r-mers of s[0, n) that hold only ACGT, in order: emit(position, forward hash, reverse hash).  One rolling step
// per base; a window that is still filling takes the same step with nothing leaving it (the reverse hash of a full
// window comes out as XOR_j srol^j(seed(3 - s_j)), its definition), and a bad character empties the window.
template <class Get, class Emit>
ABG_HD void rr_scan(const RRParams& p, Get&& get, uint32_t n, Emit&& emit)
{
	uint64_t f = 0, rv = 0;
	uint32_t run = 0;
	for (uint32_t i = 0; i < n; i++) {
		const int c = rr_code(get(i));
		if (c < 0) { f = 0; rv = 0; run = 0; continue; }
		uint64_t of = 0, orv = 0;
		if (run >= p.r) {
			const int o = rr_code(get(i - p.r)); // (inside the run: a base)
			of = p.out_f[o];
			orv = seed_of(3u - (unsigned)o);
		}
		f = srol1(f) ^ seed_of((unsigned)c) ^ of;
		rv = sror1(rv ^ p.in_r[c] ^ orv);
		run++;
		if (run >= p.r) emit(i + 1 - p.r, f, rv);
	}
}

// bit position of hash function i for the strand hashes (f, rv)
ABG_HD uint64_t rr_pos(const RRParams& p, uint64_t h0, unsigned i)
{
	uint64_t h = h0;
	if (i) {
		h = h0 * p.mult[i];
		h ^= h >> MULTISHIFT;
	}
	return mod64(p.mod, h);
}

ABG_HD void rr_set_bit(uint32_t* bits, uint64_t pos)
{
	const uint32_t m = 1u << (pos & 31);
#if defined(__HIP_DEVICE_COMPILE__)
	// (no value wanted back: a fire-and-forget global_atomic_or)
	(void)__hip_atomic_fetch_or(bits + (pos >> 5), m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
	bits[pos >> 5] |= m;
#endif
}
ABG_HD bool rr_get_bit(const uint32_t* bits, uint64_t pos) { return (bits[pos >> 5] >> (pos & 31)) & 1u; }

// KmerBloomFilter::insert over one record of `span` characters
template <class Get>
ABG_HD void rr_insert_record(const RRParams& p, Get&& get, uint32_t span, uint32_t* bits)
{
	rr_scan(p, get, span, [&](uint32_t, uint64_t f, uint64_t rv) {
		const uint64_t h0 = f + rv;
		for (unsigned i = 0; i < p.hash_num; i++) rr_set_bit(bits, rr_pos(p, h0, i));
	});
}

// does the filter hold the r-mer at s[at, at + r)?  -1: it holds a character other than ACGT (NtHash skips it)
template <class Get>
ABG_HD int rr_contains_at(const RRParams& p, Get&& get, uint32_t at, const uint32_t* bits)
{
	uint64_t f = 0, rv = 0;
	bool bad = false;
	for (uint32_t j = 0; j < p.r; j++) {
		const int c = rr_code(get(at + j));
		bad |= c < 0;
		const unsigned cc = (unsigned)c & 3u;
		f = srol1(f) ^ seed_of(cc);
		rv = sror1(rv ^ p.in_r[cc]);
	}
	if (bad) return -1;
	const uint64_t h0 = f + rv;
	bool all = true;
	for (unsigned i = 0; i < p.hash_num; i++) all &= rr_get_bit(bits, rr_pos(p, h0, i));
	return all ? 1 : 0;
}

// KmerBloomFilter::contains(seq) with the r-mers dealt out to `nlanes` lanes: this lane's share of the count
ABG_HD uint32_t rr_contains_share(const RRParams& p, const unsigned char* s, uint64_t len, uint32_t lane, uint32_t nlanes,
    const uint32_t* bits)
{
	if (len < p.r) return 0;
	const uint64_t n = len - p.r + 1;
	uint32_t found = 0;
	for (uint64_t at = lane; at < n; at += nlanes)
		found += rr_contains_at(p, [&](uint32_t i) { return (unsigned)s[i]; }, (uint32_t)at, bits) == 1;
	return found;
}

} // namespace abg
