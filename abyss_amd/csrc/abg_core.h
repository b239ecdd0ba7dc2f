// abg_core.h -- device-side building blocks of the MI355X Bloom-filter de Bruijn
// graph unitig stage: ntHash, 2-bit k-mers, counting/bit Bloom probes and the
// bounded graph searches (lookAhead / trueBranch / successor) that drive unitig
// extension.  Every function is ABG_HD (__host__ __device__) and free of global
// state, so the kernels in abg_kernels.hip are thin grid-stride wrappers around
// them, and tests/hostcheck can execute the very same code serially on a CPU.
// The product library only ever runs them on the GPU.
//
// Reference behaviour restated here (ABySS 2.3.10, paths relative to the repo):
//   vendor/nthash/nthash.hpp                 ntHash v1 (NTF64/NTR64/NTC64/NTC64L/NTE64)
//   BloomDBG/RollingHash.h                   canonical hash + H derived hashes
//   vendor/btl_bloomfilter/CountingBloomFilter.hpp   uint8 counters, minCount/contains
//   vendor/btl_bloomfilter/BloomFilter.hpp           bit filter (visited k-mers)
//   BloomDBG/RollingBloomDBG.h               implicit graph: neighbours in A,C,G,T order
//   Graph/ExtendPath.h                       lookAhead, trueBranch, successor, extendPath
#pragma once

#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ABG_HD __host__ __device__ __forceinline__
#define ABG_HDN __host__ __device__
// kept out of line on purpose: the rarely taken graph searches must not inflate the
// register pressure of the unbranched walking loop that calls them
#define ABG_HDX __host__ __device__ __attribute__((noinline))
#else
#define ABG_HD inline
#define ABG_HDN
#define ABG_HDX
#endif

namespace abg {

// ----------------------------------------------------------------- constants
constexpr int MAX_NW = 6;             // 6 x 32 bases = MAX_KMER 192 (configure.ac:151)
constexpr int MAX_HASHES = 32;        // configure.ac:156
constexpr unsigned FP_TRIM = 5;       // fpTrim / fpLookAhead, bloom-dbg.h:494,548,848

enum Dir : int { FORWARD = 0, REVERSE = 1 };       // Graph/Path.h:37
enum Sense : int { SENSE = 0, ANTISENSE = 1 };     // Common/Sense.h

// PathExtensionResultCode, Graph/ExtendPath.h:46-57
enum ExtCode : int { ER_AMBI_IN = 0, ER_AMBI_OUT, ER_DEAD_END, ER_CYCLE, ER_LENGTH_LIMIT };

// ReadResult, BloomDBG/bloom-dbg.h:256-266
enum ReadResult : int {
	RR_UNINITIALIZED = 0, RR_SHORTER_THAN_K, RR_NON_ACGT, RR_BLUNT_END, RR_NOT_SOLID,
	RR_ALL_KMERS_VISITED, RR_ALL_BRANCH_KMERS_VISITED, RR_GENERATED_CONTIGS
};

// nthash.hpp:18-29
constexpr uint64_t MULTISEED = 0x90b45d39fb6da1faULL;
constexpr int MULTISHIFT = 27;
constexpr uint64_t SEED_A = 0x3c8bfbb395c60474ULL;
constexpr uint64_t SEED_C = 0x3193c18562a02b4cULL;
constexpr uint64_t SEED_G = 0x20323ed082572324ULL;
constexpr uint64_t SEED_T = 0x295549f54be24456ULL;

// ------------------------------------------------------------------ ntHash
// rol1 + swapbits033 (nthash.hpp:186-207): rotate low 33 and high 31 bits left by 1.
ABG_HD uint64_t srol1(uint64_t v)
{
	uint64_t h = (v << 1) | (v >> 63);
	uint64_t x = (h ^ (h >> 33)) & 1;
	return h ^ (x | (x << 33));
}
// ror1 + swapbits3263 (nthash.hpp:191-217)
ABG_HD uint64_t sror1(uint64_t v)
{
	uint64_t h = (v >> 1) | (v << 63);
	uint64_t x = ((h >> 32) ^ (h >> 63)) & 1;
	return h ^ ((x << 32) | (x << 63));
}
// msTab31l[c][n%31] | msTab33r[c][n%33] (nthash.hpp:66-183)
ABG_HD uint64_t srol_n(uint64_t v, unsigned n)
{
	uint64_t lo = v & 0x1FFFFFFFFULL, hi = v >> 33;
	unsigned a = n % 33, b = n % 31;
	if (a) lo = ((lo << a) | (lo >> (33 - a))) & 0x1FFFFFFFFULL;
	if (b) hi = ((hi << b) | (hi >> (31 - b))) & 0x7FFFFFFFULL;
	return (hi << 33) | lo;
}
// seed of base code 0..3 = A,C,G,T (seedTab, nthash.hpp:31-64)
ABG_HD uint64_t seed_of(unsigned b)
{
	return b == 0 ? SEED_A : b == 1 ? SEED_C : b == 2 ? SEED_G : SEED_T;
}

// Exact h % m for a 64-bit h and runtime divisor m (CountingBloomFilter.hpp:56-58,
// BloomFilter.hpp:187,252).  Round-up magic-number division (65-bit magic, the
// "branch-free" scheme of Granlund-Montgomery / libdivide): q = floor(h / m) for
// every 64-bit h; verified against the hardware % in tests for edge values.
struct Mod64 {
	uint64_t m;
	uint64_t magic;
	uint32_t shift;
	uint32_t pow2; // m is a power of two
};
ABG_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul64hi(a, b);
#else
	return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
ABG_HD uint64_t mod64(const Mod64& d, uint64_t h)
{
	if (d.pow2)
		return h & (d.m - 1);
	uint64_t q = mulhi64(d.magic, h);
	uint64_t t = ((h - q) >> 1) + q;
	q = t >> d.shift;
	return h - q * d.m;
}
inline Mod64 make_mod64(uint64_t m)
{
	Mod64 d;
	d.m = m;
	d.magic = 0;
	d.shift = 0;
	d.pow2 = (m & (m - 1)) == 0;
	if (d.pow2)
		return d;
	unsigned fl = 63 - (unsigned)__builtin_clzll(m);
	// proposed = floor(2^(64+fl) / m), rem = 2^(64+fl) mod m
	unsigned __int128 num = (unsigned __int128)1 << (64 + fl);
	uint64_t proposed = (uint64_t)(num / m);
	uint64_t rem = (uint64_t)(num % m);
	proposed += proposed;
	uint64_t twice_rem = rem + rem;
	if (twice_rem >= m || twice_rem < rem)
		proposed += 1;
	d.magic = 1 + proposed;
	d.shift = fl;
	return d;
}

// -------------------------------------------------------------- parameters
// Spaced seed (MaskedKmer::mask(), BloomDBG/MaskedKmer.h:25-48; SpacedSeed.h): '1' positions
// take part in hashing and comparison, '0' positions do not.  maskHash (nthash.hpp:537-547)
// XORs the contribution of every '0' position out of the two strand hashes; the table holds
// those contributions per masked position and base.
constexpr int MAX_K = MAX_NW * 32;
struct MaskTab {
	uint32_t k, nmasked;
	uint8_t ones[MAX_K];              // ones[i] = 1 when mask[i] == '1'
	uint16_t ones_prefix[MAX_K + 1];  // number of '1's in mask[0, i)
	uint8_t pos[MAX_K];               // the masked ('0') positions
	uint64_t F[MAX_K][4];             // srol^(k-1-pos)(seed[b])    : term of base b in the forward hash
	uint64_t R[MAX_K][4];             // srol^(pos)(seed[3-b])      : term of base b in the reverse hash
	uint32_t nruns;                   // maximal runs [run_a, run_b) of masked positions
	uint8_t run_a[MAX_K / 2], run_b[MAX_K / 2];
};
struct Params {
	uint32_t k;        // k-mer size
	uint32_t nh;       // number of hash functions (H)
	uint32_t kc;       // minimum count threshold (--kc)
	uint32_t trim;     // max branch length to trim (-t)
	uint32_t nw;       // ceil(k / 32)
	uint32_t solid_bits; // PASS 2: the `cnt` handed to the solid-filter probes is the bit plane "counter >= kc" (probe_c)
	Mod64 mod;         // counters == visited bits (bloom-dbg.h:910)
	uint64_t kmul;     // k * multiSeed, for NTE64 (nthash.hpp:337-342)
	uint64_t seed_k[4];     // srol^k(seed[b])
	uint64_t seedrc_k[4];   // srol^k(seed[3-b])
	uint64_t seedrc_km1[4]; // srol^(k-1)(seed[3-b])
	uint64_t seed_km1[4];   // srol^(k-1)(seed[b])
	uint64_t care[MAX_NW];  // 2 bits per base, 11 where the position is compared/hashed ('1' or no mask)
	const MaskTab* mask;    // device copy of the spaced-seed table; NULL without a spaced seed
	uint32_t ident_fast;    // even k, no mask: see vtx_ident
};
inline Params make_params(uint32_t k, uint32_t nh, uint32_t kc, uint32_t trim, uint64_t m)
{
	Params p;
	p.k = k; p.nh = nh; p.kc = kc; p.trim = trim; p.nw = (k + 31) / 32; p.solid_bits = 0;
	p.mod = make_mod64(m);
	p.kmul = (uint64_t)k * MULTISEED;
	for (unsigned b = 0; b < 4; b++) {
		p.seed_k[b] = srol_n(seed_of(b), k);
		p.seedrc_k[b] = srol_n(seed_of(3 - b), k);
		p.seedrc_km1[b] = srol_n(seed_of(3 - b), k - 1);
		p.seed_km1[b] = srol_n(seed_of(b), k - 1);
	}
	for (int j = 0; j < MAX_NW; j++) p.care[j] = 0;
	for (unsigned i = 0; i < k; i++) p.care[i >> 5] |= 3ULL << (2 * (i & 31));
	p.mask = nullptr;
	p.ident_fast = (k & 1) ? 0u : 1u;
	return p;
}
// host-side construction of the spaced-seed table; `mask` is k characters of '0'/'1'
inline void make_mask(const char* mask, uint32_t k, MaskTab& t, Params& p)
{
	t.k = k; t.nmasked = 0; t.nruns = 0;
	t.ones_prefix[0] = 0;
	p.ident_fast = 0;
	for (unsigned i = 0; i < k;) {
		if (mask[i] == '1') { i++; continue; }
		unsigned e = i;
		while (e < k && mask[e] != '1') e++;
		t.run_a[t.nruns] = (uint8_t)i; t.run_b[t.nruns] = (uint8_t)e; t.nruns++;
		i = e;
	}
	for (int j = 0; j < MAX_NW; j++) p.care[j] = 0;
	for (unsigned i = 0; i < k; i++) {
		bool one = mask[i] == '1';
		t.ones[i] = one ? 1 : 0;
		t.ones_prefix[i + 1] = (uint16_t)(t.ones_prefix[i] + (one ? 1 : 0));
		if (one) { p.care[i >> 5] |= 3ULL << (2 * (i & 31)); continue; }
		unsigned j = t.nmasked++;
		t.pos[j] = (uint8_t)i;
		for (unsigned b = 0; b < 4; b++) {
			t.F[j][b] = srol_n(seed_of(b), k - 1 - i);
			t.R[j][b] = srol_n(seed_of(3 - b), i);
		}
	}
}
ABG_HD bool pos_cared(const Params& p, unsigned i) { return ((p.care[i >> 5] >> (2 * (i & 31))) & 1ULL) != 0; }

// NTE64 (nthash.hpp:337-342; note precedence i ^ (k * multiSeed)); hash 0 is the
// canonical hash itself (RollingHash::getHashes, RollingHash.h:141-146).
ABG_HD uint64_t hash_i(const Params& p, uint64_t h, unsigned i)
{
	if (i == 0) return h;
	uint64_t t = h * ((uint64_t)i ^ p.kmul);
	t ^= t >> MULTISHIFT;
	return t;
}
ABG_HD uint64_t pos_i(const Params& p, uint64_t h, unsigned i)
{
	return mod64(p.mod, hash_i(p, h, i));
}
// One probe of the solid filter as PASS 2 makes them: every caller only ever compares the counter
// with kc (CountingBloomFilter::contains with a threshold, CountingBloomFilter.hpp:169-184).  Once
// PASS 1 is over that is ONE BIT per counter, so the engine hands PASS 2 the bit plane
// "counter >= kc" (Engine::solid_plane) in place of the counters: an eighth of the bytes, i.e. a working
// set of m/8 bytes for the walkers' and the classification's random probes (238 MB for B=2G: inside the
// 256 MB Infinity Cache) instead of m.  Returns a value to compare with kc: the counter itself, or
// 255 / 0 from the plane.  (Coverage sums -- solid_min_count -- keep reading the counters.)
// (ONE load instruction whatever the mode: with a load in each arm of `if (p.solid_bits)` the compiler waits for every
// probe before it issues the next -- a group of probes meant to be in flight together became a chain of round trips)
ABG_HD unsigned probe_c(const Params& p, const uint8_t* __restrict__ cnt, uint64_t pos)
{
	const bool bits = p.solid_bits != 0;
	const unsigned byte = cnt[bits ? pos >> 3 : pos];
	return bits ? (((byte >> (pos & 7)) & 1u) ? 255u : 0u) : byte;
}

// ------------------------------------------------------------ 2-bit k-mers
// Base i lives in bits [2*(i%32), 2*(i%32)+2) of word i/32; bits past 2k are zero.
// The template parameter NW of everything below encodes two things: the number of 64-bit
// words of a k-mer (NW & 7, 1..6) and whether the code is built for a spaced seed (NW >= 8).
// The spaced-seed build carries the masked-out hash terms in every vertex and calls the mask
// helpers; the plain build has neither, so the headline path pays nothing for them (registers,
// LDS frame size) -- the engine instantiates its kernels for NW and for NW_MASKED + NW and picks
// at run time.
constexpr int NW_MASKED = 8;
template <int NW> constexpr int KW = NW & 7;
template <int NW> constexpr bool MASKED_BUILD = NW >= NW_MASKED;
template <int NW>
struct Kmer {
	uint64_t w[KW<NW>];
};
template <int NW>
ABG_HD unsigned kmer_get(const Kmer<NW>& s, unsigned i)
{
	unsigned r = 0;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++)
		if ((int)(i >> 5) == j) r = (unsigned)(s.w[j] >> (2 * (i & 31))) & 3u;
	return r;
}
template <int NW>
ABG_HD void kmer_set(Kmer<NW>& s, unsigned i, unsigned b)
{
#pragma unroll
	for (int j = 0; j < KW<NW>; j++)
		if ((int)(i >> 5) == j) {
			unsigned sh = 2 * (i & 31);
			s.w[j] = (s.w[j] & ~(3ULL << sh)) | ((uint64_t)b << sh);
		}
}
// LightweightKmer::shift (LightweightKmer.h:52-62)
template <int NW>
ABG_HD void kmer_shift(Kmer<NW>& s, unsigned k, int sense, unsigned b)
{
	if (sense == SENSE) {
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) {
			uint64_t hi = (j + 1 < KW<NW>) ? s.w[j + 1] : 0;
			s.w[j] = (s.w[j] >> 2) | (hi << 62);
		}
		kmer_set(s, k - 1, b);
	} else {
#pragma unroll
		for (int j = KW<NW> - 1; j >= 0; j--) {
			uint64_t lo = (j > 0) ? s.w[j - 1] : 0;
			s.w[j] = (s.w[j] << 2) | (lo >> 62);
		}
		s.w[0] = (s.w[0] & ~3ULL) | b;
		// clear the base shifted past position k-1
		unsigned top = k & 31;
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) {
			if (j == (int)(k >> 5)) s.w[j] = top ? (s.w[j] & ((1ULL << (2 * top)) - 1)) : 0;
			else if (j > (int)(k >> 5)) s.w[j] = 0;
		}
	}
}
// LightweightKmer::reverseComplement (LightweightKmer.h:114-129)
template <int NW>
ABG_HD Kmer<NW> kmer_revcomp(const Kmer<NW>& s, unsigned k)
{
	Kmer<NW> r;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) r.w[j] = 0;
	for (unsigned i = 0; i < k; i++)
		kmer_set(r, k - 1 - i, 3u - kmer_get(s, i));
	return r;
}
// The same without a loop over the bases: reverse the 2-bit groups of every word (and the word
// order), complement, then move the k bases down over the unused high positions.
ABG_HD uint64_t rc_word(uint64_t x)
{
	x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
	return ~__builtin_bswap64(x);
}
template <int NW>
ABG_HD Kmer<NW> kmer_revcomp_fast(const Kmer<NW>& s, unsigned k)
{
	constexpr int W = KW<NW>;
	uint64_t t[W];
#pragma unroll
	for (int j = 0; j < W; j++) t[j] = rc_word(s.w[W - 1 - j]);
	// base i now sits at position 32 W - 1 - i; it belongs at k - 1 - i
	const unsigned sh = (32u * W - k) * 2u, ws = sh >> 6, bs = sh & 63u;
	Kmer<NW> r;
#pragma unroll
	for (int j = 0; j < W; j++) {
		uint64_t lo = 0, hi = 0;
#pragma unroll
		for (int q = 0; q < W; q++) {
			if (q == j + (int)ws) lo = t[q];
			if (q == j + (int)ws + 1) hi = t[q];
		}
		r.w[j] = bs ? ((lo >> bs) | (hi << (64u - bs))) : lo;
	}
	return r;
}
// The k-mer starting at base `pos` of a packed sequence (16 bases per 32-bit word, the sequence
// starting at word `woff`): only the words holding bases [pos, pos + k) are read.
template <int NW>
ABG_HD Kmer<NW> window_kmer(const uint32_t* __restrict__ words, uint64_t woff, uint32_t pos, unsigned k)
{
	Kmer<NW> s;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) {
		s.w[j] = 0;
		if (32u * j >= k) continue;
		const unsigned nb = k - 32u * j < 32u ? k - 32u * j : 32u; // bases of this word
		const uint32_t b0 = pos + 32u * j;
		const uint64_t q = woff + (b0 >> 4);
		const unsigned sh = 2u * (b0 & 15u), need = sh + 2u * nb;  // bits [sh, need) of words q, q + 1, q + 2
		// (three loads, none under a condition -- a load in a conditional arm is waited for on the spot: a word that is not
		// needed is read again from an address that is, and left out)
		const uint32_t w0 = words[q], w1 = words[need > 32 ? q + 1 : q], w2 = words[need > 64 ? q + 2 : q];
		const uint64_t lo = (uint64_t)w0 | (need > 32 ? (uint64_t)w1 << 32 : 0ull);
		uint64_t v = lo >> sh;
		if (need > 64) v |= (uint64_t)w2 << (64u - sh);
		s.w[j] = nb < 32 ? (v & ((1ULL << (2u * nb)) - 1)) : v;
	}
	return s;
}
// LightweightKmer::isCanonical (LightweightKmer.h:88-101): compares only the first
// k/2 bases with the complement of the mirrored ones; ties count as canonical.
template <int NW>
ABG_HD bool kmer_is_canonical(const Kmer<NW>& s, unsigned k)
{
	for (unsigned i = 0; i < k / 2; i++) {
		unsigned c1 = kmer_get(s, i), c2 = 3u - kmer_get(s, k - 1 - i);
		if (c1 > c2) return false;
		if (c1 < c2) return true;
	}
	return true;
}
// LightweightKmer::operator== (LightweightKmer.h:131-146): positional equality over the
// compared ('1') positions
template <int NW>
ABG_HD bool kmer_equal(const Params& p, const Kmer<NW>& a, const Kmer<NW>& b)
{
	bool e = true;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) e = e && (((a.w[j] ^ b.w[j]) & p.care[j]) == 0);
	return e;
}

// ------------------------------------------------------------------ vertex
// RollingBloomDBGVertex (RollingBloomDBG.h:33-38) with value semantics: the k-mer
// and the forward / reverse-complement ntHash state of RollingHash (RollingHash.h:211-219).
// Spaced-seed build only: XOR of the masked positions' terms in fh / rh, so that the strand
// hashes maskHash would compute (nthash.hpp:537-547) are fh ^ df and rh ^ dr.
template <bool M> struct VtxMask { uint64_t df, dr; };
template <> struct VtxMask<false> {};
template <int NW>
struct Vtx : VtxMask<MASKED_BUILD<NW>> {
	Kmer<NW> s;
	uint64_t fh, rh;
};
template <int NW> ABG_HD uint64_t vtx_df(const Vtx<NW>& v) { if constexpr (MASKED_BUILD<NW>) return v.df; else return 0; }
template <int NW> ABG_HD uint64_t vtx_dr(const Vtx<NW>& v) { if constexpr (MASKED_BUILD<NW>) return v.dr; else return 0; }
template <int NW> ABG_HD void vtx_set_d(Vtx<NW>& v, uint64_t df, uint64_t dr)
{
	if constexpr (MASKED_BUILD<NW>) { v.df = df; v.dr = dr; } else { (void)v; (void)df; (void)dr; }
}
// (fh, rh) of the identity below
struct VKey { uint64_t fh, rh; };
// XOR of the masked positions' terms of a k-mer, from scratch (O(masked positions))
template <int NW>
ABG_HD void masked_terms(const Params& p, const Kmer<NW>& s, uint64_t& df, uint64_t& dr)
{
	df = 0; dr = 0;
	if (p.mask) {
		const MaskTab& m = *p.mask;
		for (unsigned j = 0; j < m.nmasked; j++) {
			unsigned b = kmer_get(s, m.pos[j]);
			df ^= m.F[j][b];
			dr ^= m.R[j][b];
		}
	}
}
// The same for the k-mer shifted one base in direction `sense`, from the unshifted k-mer's
// terms in O(runs of masked positions): within a run every base moves one position, which
// rotates its term by one bit, and the run gains a base at one end and loses one at the other
// (the incoming base lands on an end of the k-mer, which a mask never covers, MaskedKmer.h:43).
// srol1 / sror1 are bit permutations, hence linear over XOR.
template <int NW>
ABG_HD void masked_terms_shifted(const Params& p, const Kmer<NW>& old, uint64_t df, uint64_t dr, int sense,
    uint64_t& ndf, uint64_t& ndr)
{
	const MaskTab& m = *p.mask;
	const unsigned k = p.k;
	for (unsigned r = 0; r < m.nruns; r++) {
		// SENSE: position a leaves the run, position b enters it; ANTISENSE: b - 1 leaves, a - 1 enters
		const unsigned ia = (sense == SENSE) ? m.run_a[r] : m.run_a[r] - 1u;
		const unsigned ib = (sense == SENSE) ? m.run_b[r] : m.run_b[r] - 1u;
		const unsigned xa = kmer_get(old, ia), xb = kmer_get(old, ib);
		df ^= srol_n(seed_of(xa), k - 1 - ia) ^ srol_n(seed_of(xb), k - 1 - ib);
		dr ^= srol_n(seed_of(3u - xa), ia) ^ srol_n(seed_of(3u - xb), ib);
	}
	ndf = (sense == SENSE) ? srol1(df) : sror1(df);
	ndr = (sense == SENSE) ? sror1(dr) : srol1(dr);
}
// The two strand hashes that make the canonical hash: the rolling (fh, rh) themselves, or,
// under a spaced seed, maskHash's fsVal / rsVal: every masked position's term XORed out.
template <int NW>
ABG_HD void strand_hashes(const Vtx<NW>& v, uint64_t& fs, uint64_t& rs)
{
	fs = v.fh ^ vtx_df(v); rs = v.rh ^ vtx_dr(v);
}
// canonical hash (RollingHash.h:28-31,74-79)
template <int NW>
ABG_HD uint64_t vtx_hash(const Params& p, const Vtx<NW>& v)
{
	(void)p;
	uint64_t fs, rs;
	strand_hashes(v, fs, rs);
	return rs < fs ? rs : fs;
}
// Identity of a vertex under RollingBloomDBGVertex::operator== (RollingBloomDBG.h:92-159):
// equal canonical hash AND equal k-mers when each is read in the orientation its own
// isCanonical() selects, over the compared positions.  The compared string of a vertex is
// its k-mer if isCanonical() holds, else the reverse complement; the strand hash of that
// orientation stands for the string, so identity = (hash of the canonical orientation, hash
// of the other one) -- their minimum is the canonical hash.  This reproduces the reference's
// quirk for odd k (both orientations of a k-mer with a reverse-palindromic flank claim to be
// canonical: (fs, rs) != (rs, fs), two different vertices) and, under a spaced seed, its
// use of the UNMASKED k-mer to pick the orientation.  For even k without a mask isCanonical()
// can only tie on a true palindrome, so the ordered pair (min, max) is the same relation
// and needs no look at the k-mer (p.ident_fast).
template <int NW>
ABG_HD VKey vtx_ident(const Params& p, const Vtx<NW>& v)
{
	VKey key;
	if (p.ident_fast) {
		key.fh = v.rh < v.fh ? v.rh : v.fh;
		key.rh = v.rh < v.fh ? v.fh : v.rh;
		return key;
	}
	uint64_t fs, rs;
	strand_hashes(v, fs, rs);
	bool canon = kmer_is_canonical(v.s, p.k);
	key.fh = canon ? fs : rs;
	key.rh = canon ? rs : fs;
	return key;
}
ABG_HD bool key_equal(const VKey& a, const VKey& b) { return a.fh == b.fh && a.rh == b.rh; }
// vtx_ident of a k-mer given with its rolling hashes and (spaced seed) masked-out terms
template <int NW>
ABG_HD VKey kmer_ident(const Params& p, const Kmer<NW>& s, uint64_t fh, uint64_t rh, uint64_t df, uint64_t dr)
{
	Vtx<NW> t;
	t.s = s; t.fh = fh; t.rh = rh;
	vtx_set_d(t, df, dr);
	return vtx_ident(p, t);
}

// Canonical hash of one k-mer computed from scratch: RollingHash::reset (RollingHash.h:69-80),
// i.e. NTF64 / NTR64 base forms (nthash.hpp:220-239), and under a spaced seed what NTMC64 +
// maskHash (nthash.hpp:537-547,560-575) leave: only the '1' positions contribute (so a
// non-ACGT character under a '0' is never looked at).  get(i) returns the code 0..3 of base i.
template <class Get>
ABG_HD void scratch_hashes(const Params& p, Get get, uint64_t& fh, uint64_t& rh)
{
	fh = 0; rh = 0;
	const unsigned k = p.k;
	if (!p.mask) {
		for (unsigned i = 0; i < k; i++) {
			fh = srol1(fh) ^ seed_of(get(i));
			rh = srol1(rh) ^ seed_of(3u - get(k - 1 - i));
		}
	} else {
		for (unsigned i = 0; i < k; i++) {
			fh = srol1(fh); rh = srol1(rh);
			if (pos_cared(p, i)) fh ^= seed_of(get(i));
			if (pos_cared(p, k - 1 - i)) rh ^= seed_of(3u - get(k - 1 - i));
		}
	}
}
template <class Get>
ABG_HD uint64_t scratch_hash(const Params& p, Get get)
{
	uint64_t fh, rh;
	scratch_hashes(p, get, fh, rh);
	return rh < fh ? rh : fh;
}
// NTF64 / NTR64 base forms (nthash.hpp:220-239) of a k-mer held in words, without a spaced seed:
// the loops of vtx_rehash with the bases taken off the words as they come
template <int NW>
ABG_HD void kmer_hashes(const Kmer<NW>& s, unsigned k, uint64_t& fh_out, uint64_t& rh_out)
{
	uint64_t fh = 0, rh = 0;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) {
		if (32u * j >= k) continue;
		const unsigned nb = k - 32u * j < 32u ? k - 32u * j : 32u;
		uint64_t x = s.w[j];
		for (unsigned i = 0; i < nb; i++) { fh = srol1(fh) ^ seed_of((unsigned)x & 3u); x >>= 2; }
	}
#pragma unroll
	for (int j = KW<NW> - 1; j >= 0; j--) {
		if (32u * j >= k) continue;
		const unsigned nb = k - 32u * j < 32u ? k - 32u * j : 32u;
		uint64_t x = s.w[j] << (64u - 2u * nb);
		for (unsigned i = 0; i < nb; i++) { rh = srol1(rh) ^ seed_of(3u - (unsigned)(x >> 62)); x <<= 2; }
	}
	fh_out = fh; rh_out = rh;
}
// RollingHash::reset: the rolling (unmasked) state of a vertex
template <int NW>
ABG_HD void vtx_rehash(const Params& p, Vtx<NW>& v)
{
	kmer_hashes(v.s, p.k, v.fh, v.rh);
	if constexpr (MASKED_BUILD<NW>) masked_terms(p, v.s, v.df, v.dr);
}
// Vertex::shift (RollingBloomDBG.h:55-63): RollingHash::rollRight / rollLeft
// (RollingHash.h:88-124; NTC64 nthash.hpp:242-257,275-279; NTC64L :282-304).
template <int NW>
ABG_HD void vtx_shift(const Params& p, Vtx<NW>& v, int sense, unsigned in)
{
	unsigned k = p.k;
	if constexpr (MASKED_BUILD<NW>) masked_terms_shifted(p, v.s, v.df, v.dr, sense, v.df, v.dr);
	if (sense == SENSE) {
		unsigned out = kmer_get(v.s, 0);
		v.fh = srol1(v.fh) ^ seed_of(in) ^ p.seed_k[out];
		v.rh = sror1(v.rh ^ p.seedrc_k[in] ^ seed_of(3u - out));
	} else {
		unsigned out = kmer_get(v.s, k - 1);
		v.fh = sror1(v.fh ^ p.seed_k[in] ^ seed_of(out));
		v.rh = srol1(v.rh) ^ seed_of(3u - in) ^ p.seedrc_k[out];
	}
	kmer_shift(v.s, k, sense, in);
}
// Vertex::reverseComplement (RollingBloomDBG.h:71-75)
template <int NW>
ABG_HD void vtx_revcomp(const Params& p, Vtx<NW>& v)
{
	v.s = kmer_revcomp(v.s, p.k);
	uint64_t t = v.fh; v.fh = v.rh; v.rh = t;
	if constexpr (MASKED_BUILD<NW>) { t = v.df; v.df = v.dr; v.dr = t; } // the mask is symmetric
}
template <int NW>
ABG_HD bool vtx_equal(const Params& p, const Vtx<NW>& a, const Vtx<NW>& b)
{
	return key_equal(vtx_ident(p, a), vtx_ident(p, b));
}

// --------------------------------------------------------- Bloom filter probes
// CountingBloomFilter::contains (CountingBloomFilter.hpp:190-196): min over the H
// counters >= threshold.  All H loads are issued before any is tested.
ABG_HD bool solid_contains(const Params& p, const uint8_t* __restrict__ cnt, uint64_t h)
{
	bool ok = true;
	for (unsigned i = 0; i < p.nh; i++)
		ok = ok & (probe_c(p, cnt, pos_i(p, h, i)) >= p.kc);
	return ok;
}
// CountingBloomFilter::minCount (CountingBloomFilter.hpp:53-64)
ABG_HD unsigned solid_min_count(const Params& p, const uint8_t* __restrict__ cnt, uint64_t h)
{
	unsigned mn = 255;
	for (unsigned i = 0; i < p.nh; i++) {
		unsigned c = cnt[pos_i(p, h, i)];
		mn = c < mn ? c : mn;
	}
	return mn;
}
// BloomFilter::contains (BloomFilter.hpp:249-259)
ABG_HD bool visited_contains(const Params& p, const uint8_t* __restrict__ vis, uint64_t h)
{
	bool ok = true;
	if (p.nh <= 4) { // (up to four hash functions: the loads together, none under a condition -- in the loop below each is a round trip of its own)
		uint8_t b4[4]; uint64_t q4[4];
#pragma unroll
		for (unsigned i = 0; i < 4; i++) { q4[i] = pos_i(p, h, i < p.nh ? i : 0u); b4[i] = vis[q4[i] >> 3]; }
#pragma unroll
		for (unsigned i = 0; i < 4; i++) ok = ok & (((b4[i] >> (q4[i] & 7)) & 1u) != 0); // (i >= nh: position 0 again)
		return ok;
	}
	for (unsigned i = 0; i < p.nh; i++) {
		uint64_t q = pos_i(p, h, i);
		ok = ok & (((vis[q >> 3] >> (q & 7)) & 1u) != 0);
	}
	return ok;
}

// ---- wave-cooperative helpers ------------------------------------------------------
// A "cooperative" caller is a whole wavefront executing the same code with identical
// state in every lane (one unitig walker per wave).  Probes then spread over the lanes:
// lane l handles hash function (l & 7) of k-mer (l >> 3), and a ballot gathers the verdict.
// Non-cooperative callers (one item per lane, or the serial host check) pass coop = false.
// lanes of one wavefront exchanging data through memory they share (LDS or global): everything
// written before is visible to the wave's other lanes after
#if defined(__HIP_DEVICE_COMPILE__)
ABG_HD void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#else
ABG_HD void wave_sync() {}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// AND over the lanes of a wave-per-item kernel (nlanes == 64) / identity for serial callers
ABG_HD bool wave_all_lanes(bool v, uint32_t nlanes) { return nlanes > 1 ? __ballot(v ? 0 : 1) == 0 : v; }
ABG_HD unsigned lane_id() { return __lane_id(); }
ABG_HD uint64_t wave_ballot(bool v) { return __ballot(v ? 1 : 0); }
ABG_HD bool wave_any(bool v) { return __ballot(v ? 1 : 0) != 0; }
#else
ABG_HD bool wave_all_lanes(bool v, uint32_t) { return v; }
ABG_HD unsigned lane_id() { return 0; }
ABG_HD uint64_t wave_ballot(bool v) { return v ? 1 : 0; }
ABG_HD bool wave_any(bool v) { return v; }
#endif

// bits 2i, 2i + 1 of the result = bit i of lo, bit i of hi (i < 32)
ABG_HD uint64_t interleave32(uint64_t lo, uint64_t hi)
{
	auto spread = [](uint64_t x) -> uint64_t {
		x &= 0xFFFFFFFFULL;
		x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
		x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
		x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
		x = (x | (x << 2)) & 0x3333333333333333ULL;
		x = (x | (x << 1)) & 0x5555555555555555ULL;
		return x;
	};
	return spread(lo) | (spread(hi) << 1);
}
// A k-mer whose bases come one at a time from get(i) (a load each).  A cooperative caller has its
// lanes fetch 64 bases at once and folds them into the words with two ballots; everybody else
// loops.  get(i) is only called for i < k.
template <int NW, class Get>
ABG_HD Kmer<NW> gather_kmer(unsigned k, bool coop, Get get)
{
	Kmer<NW> s;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) s.w[j] = 0;
	if (coop) {
		const unsigned lane = lane_id();
#pragma unroll
		for (int c = 0; c < (KW<NW> + 1) / 2; c++) {
			if (64u * c >= k) continue;
			const unsigned i = 64u * c + lane;
			const unsigned b = i < k ? get(i) : 0u;
			const uint64_t lo = wave_ballot((b & 1u) != 0), hi = wave_ballot((b & 2u) != 0);
			s.w[2 * c] = interleave32(lo, hi);
			if (2 * c + 1 < KW<NW>) s.w[2 * c + 1] = interleave32(lo >> 32, hi >> 32);
		}
		return s;
	}
	for (unsigned i = 0; i < k; i++) kmer_set(s, i, get(i));
	return s;
}

// A value that is the same in every lane of a cooperative wave, read back through lane 0 so that
// the compiler knows it (scalar registers, scalar ALU); the identity for other callers.
// (uni32 / uni64 are not only a hint: some callers hold a value in lane 0 only -- a ticket drawn by one lane -- and
// broadcast it this way.  A build that made them the identity walked garbage.)
template <bool COOP> ABG_HD uint32_t uni32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP) return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#endif
	return v;
}
template <bool COOP> ABG_HD uint64_t uni64(uint64_t v)
{
	return ((uint64_t)uni32<COOP>((uint32_t)(v >> 32)) << 32) | uni32<COOP>((uint32_t)v);
}
template <bool COOP, class T> ABG_HD T* uniptr(T* p) { return (T*)uni64<COOP>((uint64_t)p); }
// ---- ntHash of up to 64 CONSECUTIVE k-mers of a packed sequence, one per lane, by prefix XOR --------
// NTF64 / NTR64 are XORs of per-base seeds rotated by the base's offset in the k-mer
// (nthash.hpp:220-239), and srol -- a rotation of the low 33 and of the high 31 bits -- distributes
// over XOR.  With a_t / c_t the seeds of base t and of its complement, t counted from the first base
// of the stretch:
//     fh(j) = XOR_i srol^(k-1-i)(a_(j+i)) = srol^(k-1+j)( P(j+k-1) ^ P(j-1) ),   P(t) = XOR_(u<=t) srol^(-u)(a_u)
//     rh(j) = XOR_i srol^(i)(c_(j+i))     = srol^(-j)   ( Q(j+k-1) ^ Q(j-1) ),   Q(t) = XOR_(u<=t) srol^(u)(c_u)
// so a wave hashes n <= 64 neighbouring k-mers with two prefix scans over its n + k - 1 bases (a few bases
// per lane) and two look-ups per lane -- some 300 operations a lane where hashing one k-mer from scratch
// (kmer_hashes) costs 27 x 2k.  The bulk steps of the walkers and the chain searches examine exactly
// such stretches of a read (walk_bulk, chain_bulk).  srol_by: rotation by a / b places of the two parts.
ABG_HD uint64_t srol_by(uint64_t v, unsigned a, unsigned b)
{
	uint64_t lo = v & 0x1FFFFFFFFULL, hi = v >> 33;
	if (a) lo = ((lo << a) | (lo >> (33 - a))) & 0x1FFFFFFFFULL;
	if (b) hi = ((hi << b) | (hi >> (31 - b))) & 0x7FFFFFFFULL;
	return (hi << 33) | lo;
}
ABG_HD uint64_t srol_fwd(uint64_t v, unsigned n) { return srol_by(v, n % 33, n % 31); }
ABG_HD uint64_t srol_back(uint64_t v, unsigned n) { return srol_by(v, (33 - n % 33) % 33, (31 - n % 31) % 31); }
constexpr unsigned STRETCH_EPL = 4; // bases per lane at most: 64 + MAX_K - 1 <= 4 x 64
// The serial form of the same computation over arrays (hostcheck's self-test checks it against
// kmer_hashes; the wave form below is this with the prefixes spread over the lanes).
template <int NW>
inline void stretch_hashes_serial(const uint32_t* words, uint64_t woff, uint32_t qlo, uint32_t n, unsigned k, uint64_t* fh, uint64_t* rh)
{
	const uint32_t nb = n + k - 1;
	uint64_t P[64 + MAX_K], Q[64 + MAX_K];
	uint64_t pf = 0, pr = 0;
	for (uint32_t t = 0; t < nb; t++) {
		const uint32_t bp = qlo + t;
		const unsigned base = (words[woff + (bp >> 4)] >> (2u * (bp & 15u))) & 3u;
		pf ^= srol_back(seed_of(base), t); pr ^= srol_fwd(seed_of(3u - base), t);
		P[t] = pf; Q[t] = pr;
	}
	for (uint32_t j = 0; j < n; j++) {
		fh[j] = srol_fwd(P[j + k - 1] ^ (j ? P[j - 1] : 0), k - 1 + j);
		rh[j] = srol_back(Q[j + k - 1] ^ (j ? Q[j - 1] : 0), j);
	}
}
#if defined(__HIP_DEVICE_COMPILE__)
ABG_HD uint64_t shfl64(uint64_t v, unsigned src)
{
	return ((uint64_t)(uint32_t)__shfl((int)(v >> 32), (int)src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, (int)src);
}
// lane l < n: the strand hashes of the k-mer at base qlo + l (all 64 lanes must call)
template <int NW>
ABG_HD void stretch_hashes_wave(const uint32_t* __restrict__ words, uint64_t woff, uint32_t qlo, uint32_t n, unsigned k,
    uint64_t& fh, uint64_t& rh)
{
	const unsigned lane = __lane_id();
	const uint32_t nb = n + k - 1;
	const uint32_t epl = (nb + 63) >> 6; // bases per lane: element e = lane * epl + i
	uint64_t pf[STRETCH_EPL], pr[STRETCH_EPL];
	uint64_t af = 0, ar = 0;
	// the lane's (at most four, consecutive) bases sit in one packed word or two: both loads go out before anything waits
	const uint32_t t0 = lane * epl, bp0 = qlo + t0;
	uint64_t two = 0;
	if (t0 < nb) {
		const uint32_t tl = t0 + epl - 1u < nb ? t0 + epl - 1u : nb - 1u; // the last base this lane takes
		const uint64_t wi = woff + (bp0 >> 4);
		const uint32_t lo = words[wi];
		const uint32_t hi = ((qlo + tl) >> 4) != (bp0 >> 4) ? words[wi + 1] : 0u;
		two = (uint64_t)lo | ((uint64_t)hi << 32);
	}
#pragma unroll
	for (unsigned i = 0; i < STRETCH_EPL; i++) {
		const uint32_t t = t0 + i;
		if (i < epl && t < nb) {
			const unsigned base = (unsigned)(two >> (2u * ((bp0 & 15u) + i))) & 3u;
			af ^= srol_back(seed_of(base), t); ar ^= srol_fwd(seed_of(3u - base), t);
		}
		pf[i] = af; pr[i] = ar;
	}
	// exclusive scan of the lanes' totals
	uint64_t sf = af, sr = ar;
#pragma unroll
	for (unsigned d = 1; d < 64; d <<= 1) {
		const uint64_t tf = shfl64(sf, lane >= d ? lane - d : lane), tr = shfl64(sr, lane >= d ? lane - d : lane);
		if (lane >= d) { sf ^= tf; sr ^= tr; }
	}
	sf ^= af; sr ^= ar;
#pragma unroll
	for (unsigned i = 0; i < STRETCH_EPL; i++) { pf[i] ^= sf; pr[i] ^= sr; }
	// P / Q at the two ends of this lane's k-mer: elements l + k - 1 and l - 1
	auto fetch = [&](uint32_t t, uint64_t& vf, uint64_t& vr) {
		const unsigned src = t / epl, idx = t - src * epl;
		vf = 0; vr = 0;
#pragma unroll
		for (unsigned i = 0; i < STRETCH_EPL; i++) {
			if (i >= epl) break; // (wave-uniform)
			const uint64_t xf = shfl64(pf[i], src & 63u), xr = shfl64(pr[i], src & 63u);
			if (idx == i) { vf = xf; vr = xr; }
		}
	};
	const uint32_t l = lane < n ? lane : 0;
	uint64_t hf, hr, lf, lr;
	fetch(l + k - 1, hf, hr);
	fetch(l ? l - 1 : 0, lf, lr);
	if (!l) { lf = 0; lr = 0; }
	fh = srol_fwd(hf ^ lf, k - 1 + l);
	rh = srol_back(hr ^ lr, l);
}
#endif

// The vertex of k bases fetched one by one (get(i), i < k): gather_kmer + vtx_rehash.  A cooperative caller
// has base i in lane i while gathering, so the strand hashes come out of the same pass -- every lane rotates
// its base's seed into place (NTF64 / NTR64 term by term, nthash.hpp:220-239) and six XOR butterflies sum
// them -- instead of a k-round hash repeated by all 64 lanes afterwards.
template <int NW, class Get>
ABG_HD Vtx<NW> gather_vertex(const Params& p, bool coop, Get get)
{
	Vtx<NW> v;
#if defined(__HIP_DEVICE_COMPILE__)
	if (coop) {
		const unsigned k = p.k, lane = __lane_id();
		uint64_t tf = 0, tr = 0;
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) v.s.w[j] = 0;
#pragma unroll
		for (int c = 0; c < (KW<NW> + 1) / 2; c++) {
			if (64u * c >= k) continue;
			const unsigned i = 64u * c + lane;
			const unsigned b = i < k ? get(i) : 0u;
			const uint64_t lo = __ballot((b & 1u) != 0), hi = __ballot((b & 2u) != 0);
			v.s.w[2 * c] = interleave32(lo, hi);
			if (2 * c + 1 < KW<NW>) v.s.w[2 * c + 1] = interleave32(lo >> 32, hi >> 32);
			if (i < k) { tf ^= srol_fwd(seed_of(b & 3u), k - 1u - i); tr ^= srol_fwd(seed_of(3u - (b & 3u)), i); }
		}
#pragma unroll
		for (unsigned d = 1; d < 64; d <<= 1) { tf ^= shfl64(tf, lane ^ d); tr ^= shfl64(tr, lane ^ d); }
		v.fh = tf; v.rh = tr;
		if constexpr (MASKED_BUILD<NW>) masked_terms(p, v.s, v.df, v.dr);
		return v;
	}
#endif
	v.s = gather_kmer<NW>(p.k, coop, get);
	vtx_rehash(p, v);
	return v;
}
// a copy of the parameters whose fields the compiler knows to be wave-uniform
template <bool COOP> ABG_HD Params uniform_params(const Params& p)
{
	Params u;
	u.k = uni32<COOP>(p.k); u.nh = uni32<COOP>(p.nh); u.kc = uni32<COOP>(p.kc);
	u.trim = uni32<COOP>(p.trim); u.nw = uni32<COOP>(p.nw); u.solid_bits = uni32<COOP>(p.solid_bits);
	u.mod.m = uni64<COOP>(p.mod.m); u.mod.magic = uni64<COOP>(p.mod.magic);
	u.mod.shift = uni32<COOP>(p.mod.shift); u.mod.pow2 = uni32<COOP>(p.mod.pow2);
	u.kmul = uni64<COOP>(p.kmul);
#pragma unroll
	for (int b = 0; b < 4; b++) {
		u.seed_k[b] = uni64<COOP>(p.seed_k[b]); u.seedrc_k[b] = uni64<COOP>(p.seedrc_k[b]);
		u.seedrc_km1[b] = uni64<COOP>(p.seedrc_km1[b]); u.seed_km1[b] = uni64<COOP>(p.seed_km1[b]);
	}
#pragma unroll
	for (int j = 0; j < MAX_NW; j++) u.care[j] = uni64<COOP>(p.care[j]);
	u.mask = uniptr<COOP>(p.mask);
	u.ident_fast = uni32<COOP>(p.ident_fast);
	return u;
}

// cooperative callers work on the uniform copy, the others on the original
template <bool COOP> struct ParamsView {
	Params v;
	ABG_HD explicit ParamsView(const Params& p) : v(uniform_params<true>(p)) {}
	ABG_HD const Params& get() const { return v; }
};
template <> struct ParamsView<false> {
	const Params& r;
	ABG_HD explicit ParamsView(const Params& p) : r(p) {}
	ABG_HD const Params& get() const { return r; }
};

// 8-bit mask: which of eight canonical hashes does the solid filter contain
// (CountingBloomFilter::contains, CountingBloomFilter.hpp:190-196: min over the H
// counters >= threshold)?  Serial form: the probe positions of all k-mers (up to four
// hash functions at a time) are computed first and their loads issued back to back, so a
// probe round costs one memory latency.  Cooperative form: one (k-mer, hash) per lane.
ABG_HD unsigned solid_mask8(const Params& p, const uint8_t* __restrict__ cnt, const uint64_t h[8], bool coop)
{
	unsigned ok = 0xFFu;
	if (coop) {
		const unsigned lane = lane_id();
		const unsigned b = lane >> 3, i = lane & 7;
		uint64_t hb = h[0];
#pragma unroll
		for (unsigned q = 1; q < 8; q++) hb = (b == q) ? h[q] : hb;
		for (unsigned base = 0; base < p.nh; base += 8) {
			bool bad = false;
			if (base + i < p.nh) bad = probe_c(p, cnt, pos_i(p, hb, base + i)) < p.kc;
			uint64_t m = wave_ballot(bad);
#pragma unroll
			for (unsigned q = 0; q < 8; q++)
				if ((m >> (8 * q)) & 0xFFu) ok &= ~(1u << q);
		}
		return ok;
	}
	for (unsigned base = 0; base < p.nh; base += 4) {
		uint8_t c[8][4];
#pragma unroll
		for (unsigned b = 0; b < 8; b++) {
#pragma unroll
			for (unsigned i = 0; i < 4; i++) {
				unsigned ii = base + i < p.nh ? base + i : 0; // surplus slots re-probe hash 0
				c[b][i] = (uint8_t)probe_c(p, cnt, pos_i(p, h[b], ii));
			}
		}
#pragma unroll
		for (unsigned b = 0; b < 8; b++) {
#pragma unroll
			for (unsigned i = 0; i < 4; i++)
				if (c[b][i] < p.kc) ok &= ~(1u << b);
		}
	}
	return ok;
}
// Split form of the cooperative probe for callers that want to overlap the probe latency
// with other memory operations: probe8_issue starts this lane's load (valid for
// num_hashes <= 8), probe8_collect turns the loaded counters into the 8-bit mask.
struct Probe8 { uint8_t c; bool active; };
ABG_HD Probe8 probe8_issue(const Params& p, const uint8_t* __restrict__ cnt, const uint64_t h[8])
{
	const unsigned lane = lane_id();
	const unsigned b = lane >> 3, i = lane & 7;
	uint64_t hb = h[0];
#pragma unroll
	for (unsigned q = 1; q < 8; q++) hb = (b == q) ? h[q] : hb;
	Probe8 r;
	r.active = i < p.nh;
	r.c = 255;
	if (r.active) r.c = (uint8_t)probe_c(p, cnt, pos_i(p, hb, i));
	return r;
}
ABG_HD unsigned probe8_collect(const Params& p, const Probe8& r)
{
	uint64_t m = wave_ballot(r.active && r.c < p.kc);
	unsigned ok = 0xFFu;
#pragma unroll
	for (unsigned q = 0; q < 8; q++)
		if ((m >> (8 * q)) & 0xFFu) ok &= ~(1u << q);
	return ok;
}

// the same for four hashes
ABG_HD unsigned solid_mask4(const Params& p, const uint8_t* __restrict__ cnt, const uint64_t h[4], bool coop)
{
	if (coop) {
		uint64_t h8[8];
#pragma unroll
		for (unsigned q = 0; q < 4; q++) { h8[q] = h[q]; h8[4 + q] = h[q]; }
		return solid_mask8(p, cnt, h8, true) & 0xFu;
	}
	unsigned ok = 0xFu;
	for (unsigned base = 0; base < p.nh; base += 4) {
		uint8_t c[4][4];
#pragma unroll
		for (unsigned b = 0; b < 4; b++) {
#pragma unroll
			for (unsigned i = 0; i < 4; i++) {
				unsigned ii = base + i < p.nh ? base + i : 0;
				c[b][i] = (uint8_t)probe_c(p, cnt, pos_i(p, h[b], ii));
			}
		}
#pragma unroll
		for (unsigned b = 0; b < 4; b++) {
#pragma unroll
			for (unsigned i = 0; i < 4; i++)
				if (c[b][i] < p.kc) ok &= ~(1u << b);
		}
	}
	return ok;
}

// Neighbour enumeration of out_edge_iterator / in_edge_iterator
// (RollingBloomDBG.h:302-427): the k-mer shifted by one base with last (first) base
// A,C,G,T in that order, present iff the solid filter contains it (vertex_exists,
// :436-446).  The four neighbours' hashes are XOR deltas off one shifted state.
// a[i] for a run-time i without indexing memory: keeps a table held in registers in registers
ABG_HD uint64_t sel4(const uint64_t a[4], unsigned i)
{
	uint64_t r = a[0];
	r = (i == 1) ? a[1] : r;
	r = (i == 2) ? a[2] : r;
	r = (i == 3) ? a[3] : r;
	return r;
}
template <int NW>
ABG_HD void neighbour_hashes(const Params& p, const Vtx<NW>& u, int sense, uint64_t fh4[4], uint64_t rh4[4])
{
	unsigned k = p.k;
	if (sense == SENSE) {
		unsigned out = kmer_get(u.s, 0);
		uint64_t fb = srol1(u.fh) ^ sel4(p.seed_k, out);
		uint64_t rb = sror1(u.rh ^ seed_of(3u - out));
#pragma unroll
		for (unsigned b = 0; b < 4; b++) {
			fh4[b] = fb ^ seed_of(b);
			rh4[b] = rb ^ p.seedrc_km1[b];
		}
	} else {
		unsigned out = kmer_get(u.s, k - 1);
		uint64_t fb = sror1(u.fh ^ seed_of(out));
		uint64_t rb = srol1(u.rh) ^ sel4(p.seedrc_k, out);
#pragma unroll
		for (unsigned b = 0; b < 4; b++) {
			fh4[b] = fb ^ p.seed_km1[b];
			rh4[b] = rb ^ seed_of(3u - b);
		}
	}
}
// Under a spaced seed the four neighbours in one direction share their masked-out terms: the
// mask starts and ends with '1' (MaskedKmer.h:43), so no masked position holds the base that
// differs between them.  (df, dr) turn their rolling hashes into maskHash's strand hashes.
template <int NW>
ABG_HD void neighbour_mask_delta(const Params& p, const Vtx<NW>& u, int sense, uint64_t& df, uint64_t& dr)
{
	df = 0; dr = 0;
	if constexpr (MASKED_BUILD<NW>) masked_terms_shifted(p, u.s, u.df, u.dr, sense, df, dr);
	else (void)p;
}
// Returns a 4-bit mask (bit b = neighbour with base b exists) and the (rolling) hash pairs.
template <int NW>
ABG_HD unsigned neighbour_mask(const Params& p, const uint8_t* __restrict__ cnt,
    const Vtx<NW>& u, int sense, uint64_t fh4[4], uint64_t rh4[4], bool coop)
{
	neighbour_hashes(p, u, sense, fh4, rh4);
	uint64_t df, dr;
	neighbour_mask_delta(p, u, sense, df, dr);
	uint64_t h[4];
#pragma unroll
	for (unsigned b = 0; b < 4; b++) {
		uint64_t fs = fh4[b] ^ df, rs = rh4[b] ^ dr;
		h[b] = rs < fs ? rs : fs;
	}
	return solid_mask4(p, cnt, h, coop);
}
// the neighbour of u with last (first) base b, hashes included
template <int NW>
ABG_HD Vtx<NW> neighbour_vertex(const Params& p, const Vtx<NW>& u, int sense, unsigned b)
{
	Vtx<NW> v = u;
	vtx_shift(p, v, sense, b);
	return v;
}
template <int NW>
ABG_HD Vtx<NW> make_neighbour(const Params& p, const Vtx<NW>& u, int sense, unsigned b,
    uint64_t fh, uint64_t rh)
{
	Vtx<NW> v = u;
	if constexpr (MASKED_BUILD<NW>) masked_terms_shifted(p, u.s, u.df, u.dr, sense, v.df, v.dr);
	kmer_shift(v.s, p.k, sense, b);
	v.fh = fh;
	v.rh = rh;
	return v;
}

// The rolling-hash tables as named scalars, for loops that must stay in registers: an array
// held in registers that is indexed at run time -- even through a chain of selects, which the
// optimiser folds back into an indexed load -- is demoted to per-lane scratch memory.
// (Round 4, late: not the sixteen rotated seeds themselves -- thirty-two scalar registers alive across a search loop whose scalars
// already spill into vector lanes by the hundred -- but the four rotation amounts: srol^k(seed[b]) is worked out where it is
// used, from an immediate and two shifts per half.)
struct SeedTabs {
	uint32_t ka, kb, ma, mb; // k % 33, k % 31, (k - 1) % 33, (k - 1) % 31: the split rotations of srol_n
	ABG_HD static uint64_t rot(uint64_t v, unsigned a, unsigned b)
	{
		uint64_t lo = v & 0x1FFFFFFFFULL, hi = v >> 33; // (a == 0 / b == 0: the right shifts leave nothing, no special case)
		lo = ((lo << a) | (lo >> (33u - a))) & 0x1FFFFFFFFULL;
		hi = ((hi << b) | (hi >> (31u - b))) & 0x7FFFFFFFULL;
		return (hi << 33) | lo;
	}
	ABG_HD uint64_t sk(unsigned b) const { return rot(seed_of(b), ka, kb); }        // p.seed_k[b]
	ABG_HD uint64_t rk(unsigned b) const { return rot(seed_of(3u - b), ka, kb); }   // p.seedrc_k[b]
	ABG_HD uint64_t sm(unsigned b) const { return rot(seed_of(b), ma, mb); }        // p.seed_km1[b]
	ABG_HD uint64_t rm(unsigned b) const { return rot(seed_of(3u - b), ma, mb); }   // p.seedrc_km1[b]
};
ABG_HD SeedTabs seed_tabs(const Params& p)
{
	SeedTabs t;
	t.ka = p.k % 33u; t.kb = p.k % 31u; t.ma = (p.k - 1u) % 33u; t.mb = (p.k - 1u) % 31u;
	return t;
}
ABG_HD uint64_t pick4(unsigned i, uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3)
{
	return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3;
}
// rolling state of `v` shifted one base in direction `sense`, before the incoming base is added
// (the part of neighbour_hashes the four neighbours share)
template <int NW>
ABG_HD void nbr_base(const SeedTabs& t, const Vtx<NW>& v, unsigned k, int sense, uint64_t& fb, uint64_t& rb)
{
	if (sense == SENSE) {
		const unsigned out = kmer_get(v.s, 0);
		fb = srol1(v.fh) ^ t.sk(out);
		rb = sror1(v.rh ^ seed_of(3u - out));
	} else {
		const unsigned out = kmer_get(v.s, k - 1);
		fb = sror1(v.fh ^ seed_of(out));
		rb = srol1(v.rh) ^ t.rk(out);
	}
}
// hashes of the neighbour with base b
ABG_HD void nbr_hash(const SeedTabs& t, int sense, uint64_t fb, uint64_t rb, unsigned b, uint64_t& fh, uint64_t& rh)
{
	if (sense == SENSE) { fh = fb ^ seed_of(b); rh = rb ^ t.rm(b); }
	else { fh = fb ^ t.sm(b); rh = rb ^ seed_of(3u - b); }
}

// 4-bit mask of the neighbours of `v` in direction `sense` that the solid filter contains
// (neighbour_mask without arrays; cooperative callers probe one (neighbour, hash) per lane)
template <int NW, bool COOP>
ABG_HD unsigned nbr_mask_lean(const Params& p, const SeedTabs& t, const uint8_t* __restrict__ cnt, const Vtx<NW>& v, int sense)
{
	uint64_t fb, rb, df, dr;
	nbr_base(t, v, p.k, sense, fb, rb);
	neighbour_mask_delta(p, v, sense, df, dr);
	unsigned ok = 0xFu;
	if (COOP) {
		const unsigned lane = lane_id(), b = (lane >> 3) & 3u, i = lane & 7u;
		uint64_t fh, rh;
		nbr_hash(t, sense, fb, rb, b, fh, rh);
		fh ^= df; rh ^= dr;
		const uint64_t h = rh < fh ? rh : fh;
		for (unsigned base = 0; base < p.nh; base += 8) {
			bool bad = false;
			if (lane < 32 && base + i < p.nh) bad = probe_c(p, cnt, pos_i(p, h, base + i)) < p.kc;
			const uint64_t m = wave_ballot(bad);
#pragma unroll
			for (unsigned q = 0; q < 4; q++)
				if ((m >> (8 * q)) & 0xFFu) ok &= ~(1u << q);
		}
	} else {
		for (unsigned q = 0; q < 4; q++) {
			uint64_t fh, rh;
			nbr_hash(t, sense, fb, rb, q, fh, rh);
			fh ^= df; rh ^= dr;
			if (!solid_contains(p, cnt, rh < fh ? rh : fh)) ok &= ~(1u << q);
		}
	}
	return ok;
}
// Which neighbours of a vertex exist is a pure function of the vertex and the solid filter, and
// the searches ask it of the same vertices again and again: trueBranch wants a vertex's neighbours
// ahead when it enters it and the ones behind when it turns around, lookAhead re-explores what the
// search before it explored, and in a tangle successive successor() calls walk the same few
// hundred vertices.  A walker therefore keeps the answers for both directions in a small
// direct-mapped table in fast memory (LDS), filled by ONE probe round per vertex: lanes 0-31 probe
// the neighbours in SENSE direction, lanes 32-63 the ones in ANTISENSE direction.
constexpr uint32_t MC_N = 256;
struct MaskCache {
	uint64_t fh[MC_N], rh[MC_N]; // the vertex (its oriented strand hashes)
	uint8_t m[MC_N];             // bits 0-3: neighbours in SENSE direction, bits 4-7: in ANTISENSE direction
	uint8_t valid[MC_N];
};
template <int NW, bool COOP>
ABG_HD unsigned nbr_mask_cached(const Params& p, const SeedTabs& t, const uint8_t* __restrict__ cnt, const Vtx<NW>& v, int sense,
    MaskCache* mc)
{
	if constexpr (MASKED_BUILD<NW>) { (void)mc; return nbr_mask_lean<NW, COOP>(p, t, cnt, v, sense); }
	else {
	if (!mc || p.nh > 8) return nbr_mask_lean<NW, COOP>(p, t, cnt, v, sense);
	const uint32_t slot = (uint32_t)((v.fh ^ (v.rh >> 17)) * 0x9E3779B97F4A7C15ULL >> 40) & (MC_N - 1);
	if (mc->valid[slot] && mc->fh[slot] == v.fh && mc->rh[slot] == v.rh) {
		const unsigned m = mc->m[slot];
		return sense == SENSE ? (m & 0xFu) : (m >> 4);
	}
	unsigned ok = 0xFFu;
	uint64_t fb_s, rb_s, fb_a, rb_a;
	nbr_base(t, v, p.k, SENSE, fb_s, rb_s);
	nbr_base(t, v, p.k, ANTISENSE, fb_a, rb_a);
	if (COOP) {
		const unsigned lane = lane_id(), s = lane >> 5, b = (lane >> 3) & 3u, i = lane & 7u;
		uint64_t fh, rh;
		nbr_hash(t, s ? ANTISENSE : SENSE, s ? fb_a : fb_s, s ? rb_a : rb_s, b, fh, rh);
		const uint64_t h = rh < fh ? rh : fh;
		bool bad = false;
		if (i < p.nh) bad = probe_c(p, cnt, pos_i(p, h, i)) < p.kc;
		const uint64_t bm = wave_ballot(bad);
#pragma unroll
		for (unsigned q = 0; q < 8; q++)
			if ((bm >> (8 * q)) & 0xFFu) ok &= ~(1u << q);
	} else {
		for (unsigned q = 0; q < 8; q++) {
			uint64_t fh, rh;
			nbr_hash(t, q < 4 ? SENSE : ANTISENSE, q < 4 ? fb_s : fb_a, q < 4 ? rb_s : rb_a, q & 3u, fh, rh);
			if (!solid_contains(p, cnt, rh < fh ? rh : fh)) ok &= ~(1u << q);
		}
	}
	mc->fh[slot] = v.fh; mc->rh[slot] = v.rh; mc->m[slot] = (uint8_t)ok; mc->valid[slot] = 1;
	wave_sync();
	return sense == SENSE ? (ok & 0xFu) : (ok >> 4);
	}
}
// Both masks of a vertex from ONE probe round, without the cache: bits 0-3 the neighbours in SENSE direction,
// bits 4-7 the ones in ANTISENSE direction (lanes 0-31 / 32-63 of a cooperative caller).  trueBranch asks for the
// neighbours ahead of a vertex when it enters it and for the ones behind when it comes back from a dead end below
// it -- a second dependent round trip per vertex of a failed branch unless both come with the first.
// (Not under a spaced seed or with more than 8 hash functions: returns 0x100 and the caller asks per direction.)
template <int NW, bool COOP>
ABG_HD unsigned nbr_mask_both(const Params& p, const SeedTabs& t, const uint8_t* __restrict__ cnt, const Vtx<NW>& v)
{
	if constexpr (MASKED_BUILD<NW>) { (void)t; (void)cnt; (void)v; return 0x100u; }
	else {
	if (p.nh > 8) return 0x100u;
	unsigned ok = 0xFFu;
	uint64_t fb_s, rb_s, fb_a, rb_a;
	nbr_base(t, v, p.k, SENSE, fb_s, rb_s);
	nbr_base(t, v, p.k, ANTISENSE, fb_a, rb_a);
	if (COOP) {
		const unsigned lane = lane_id(), s = lane >> 5, b = (lane >> 3) & 3u, i = lane & 7u;
		uint64_t fh, rh;
		nbr_hash(t, s ? ANTISENSE : SENSE, s ? fb_a : fb_s, s ? rb_a : rb_s, b, fh, rh);
		const uint64_t h = rh < fh ? rh : fh;
		bool bad = false;
		if (i < p.nh) bad = probe_c(p, cnt, pos_i(p, h, i)) < p.kc;
		const uint64_t bm = wave_ballot(bad);
#pragma unroll
		for (unsigned q = 0; q < 8; q++)
			if ((bm >> (8 * q)) & 0xFFu) ok &= ~(1u << q);
	} else {
		for (unsigned q = 0; q < 8; q++) {
			uint64_t fh, rh;
			nbr_hash(t, q < 4 ? SENSE : ANTISENSE, q < 4 ? fb_s : fb_a, q < 4 ? rb_s : rb_a, q & 3u, fh, rh);
			if (!solid_contains(p, cnt, rh < fh ? rh : fh)) ok &= ~(1u << q);
		}
	}
	return ok;
	}
}
template <int NW>
ABG_HD Vtx<NW> nbr_vertex_lean(const Params& p, const SeedTabs& t, const Vtx<NW>& v, int sense, unsigned b)
{
	uint64_t fb, rb, fh, rh;
	nbr_base(t, v, p.k, sense, fb, rb);
	nbr_hash(t, sense, fb, rb, b, fh, rh);
	return make_neighbour(p, v, sense, b, fh, rh);
}

// --------------------------------------------------------------- atomics
#if defined(__HIP_DEVICE_COMPILE__)
ABG_HD uint64_t ld_coherent(const uint64_t* p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ABG_HD uint32_t ld_coherent(const uint32_t* p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ABG_HD void st_coherent(uint64_t* p, uint64_t v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ABG_HD uint64_t cas_u64(uint64_t* p, uint64_t expect, uint64_t val)
{
	return (uint64_t)atomicCAS((unsigned long long*)p, (unsigned long long)expect,
	    (unsigned long long)val);
}
ABG_HD uint32_t cas_u32(uint32_t* p, uint32_t expect, uint32_t val) { return atomicCAS(p, expect, val); }
ABG_HD uint32_t atomic_min_u32(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
ABG_HD uint64_t atomic_min_u64(uint64_t* p, uint64_t v)
{
	return (uint64_t)atomicMin((unsigned long long*)p, (unsigned long long)v);
}
ABG_HD uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
ABG_HD uint64_t atomic_add_u64(uint64_t* p, uint64_t v)
{
	return (uint64_t)atomicAdd((unsigned long long*)p, (unsigned long long)v);
}
ABG_HD uint32_t atomic_or_u32(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
ABG_HD uint64_t atomic_exch_u64(uint64_t* p, uint64_t v)
{
	return (uint64_t)atomicExch((unsigned long long*)p, (unsigned long long)v);
}
#else
// serial execution (tests/hostcheck): one item at a time, plain memory
ABG_HD uint64_t ld_coherent(const uint64_t* p) { return *p; }
ABG_HD uint32_t ld_coherent(const uint32_t* p) { return *p; }
ABG_HD void st_coherent(uint64_t* p, uint64_t v) { *p = v; }
ABG_HD uint64_t cas_u64(uint64_t* p, uint64_t expect, uint64_t val)
{
	uint64_t old = *p;
	if (old == expect) *p = val;
	return old;
}
ABG_HD uint32_t cas_u32(uint32_t* p, uint32_t expect, uint32_t val) { uint32_t o = *p; if (o == expect) *p = val; return o; }
ABG_HD uint32_t atomic_min_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
ABG_HD uint64_t atomic_min_u64(uint64_t* p, uint64_t v) { uint64_t o = *p; if (v < o) *p = v; return o; }
ABG_HD uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
ABG_HD uint64_t atomic_add_u64(uint64_t* p, uint64_t v) { uint64_t o = *p; *p = o + v; return o; }
ABG_HD uint32_t atomic_or_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
ABG_HD uint64_t atomic_exch_u64(uint64_t* p, uint64_t v) { uint64_t o = *p; *p = v; return o; }
#endif

// Append slots for a one-item-per-lane kernel: every lane of the wave that `want`s a slot
// gets a distinct index from *counter with ONE atomic per wavefront (ballot + prefix popcount)
// instead of one same-address atomic per lane.  Must be reached by all active lanes together.
#if defined(__HIP_DEVICE_COMPILE__)
ABG_HD uint32_t wave_append_slot(uint32_t* counter, bool want)
{
	const uint64_t m = __ballot(want ? 1 : 0);
	if (m == 0) return 0;
	const unsigned lane = __lane_id();
	const int leader = __ffsll((unsigned long long)m) - 1;
	uint32_t base = 0;
	if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
	base = (uint32_t)__shfl((int)base, leader);
	return base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
}
#else
ABG_HD uint32_t wave_append_slot(uint32_t* counter, bool want)
{
	if (!want) return 0;
	uint32_t o = *counter; *counter = o + 1; return o;
}
#endif

// *counter += the lanes of the wavefront that `want`, with one atomic per wavefront.
#if defined(__HIP_DEVICE_COMPILE__)
ABG_HD void wave_count_add(uint64_t* counter, bool want)
{
	const uint64_t m = __ballot(want ? 1 : 0);
	if (m && (int)__lane_id() == __ffsll((unsigned long long)m) - 1) atomicAdd((unsigned long long*)counter, (unsigned long long)__popcll(m));
}
#else
ABG_HD void wave_count_add(uint64_t* counter, bool want) { if (want) *counter += 1; }
#endif

// A lane's rank among the lanes of its wavefront that `want`, and how many do.  Must be reached by all active lanes together.
// (A serial caller is a wave of one lane.)
#if defined(__HIP_DEVICE_COMPILE__)
ABG_HD uint32_t wave_rank(bool want, uint32_t& count)
{
	const uint64_t m = __ballot(want ? 1 : 0);
	count = (uint32_t)__popcll(m);
	return (uint32_t)__popcll(m & ((1ull << __lane_id()) - 1));
}
#else
ABG_HD uint32_t wave_rank(bool want, uint32_t& count) { count = want ? 1u : 0u; return 0; }
#endif

// Atomics issued by a cooperative caller (a whole wavefront in lock step, see
// abg_core.h): lane 0 performs the operation, every lane receives its result.
#if defined(__HIP_DEVICE_COMPILE__)
ABG_HD uint64_t wu_cas_u64(uint64_t* p, uint64_t expect, uint64_t val, bool coop)
{
	if (!coop) return cas_u64(p, expect, val);
	uint64_t r = 0;
	if (__lane_id() == 0) r = cas_u64(p, expect, val);
	return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r >> 32)) << 32) |
	       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r);
}
ABG_HD uint32_t wu_atomic_min_u32(uint32_t* p, uint32_t v, bool coop)
{
	if (!coop) return atomic_min_u32(p, v);
	uint32_t r = 0;
	if (__lane_id() == 0) r = atomic_min_u32(p, v);
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
ABG_HD uint32_t wu_atomic_add_u32(uint32_t* p, uint32_t v, bool coop)
{
	if (!coop) return atomic_add_u32(p, v);
	uint32_t r = 0;
	if (__lane_id() == 0) r = atomic_add_u32(p, v);
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
ABG_HD uint64_t wu_atomic_add_u64(uint64_t* p, uint64_t v, bool coop)
{
	if (!coop) return atomic_add_u64(p, v);
	uint64_t r = 0;
	if (__lane_id() == 0) r = atomic_add_u64(p, v);
	return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r >> 32)) << 32) |
	       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r);
}
// write-through stores do not coalesce across lanes: one lane stores for the wave
ABG_HD void wu_st_coherent(uint64_t* p, uint64_t v, bool coop)
{
	if (!coop || __lane_id() == 0) st_coherent(p, v);
}
ABG_HD void wu_st_u32(uint32_t* p, uint32_t v, bool coop)
{
	if (!coop || __lane_id() == 0) *p = v;
}
ABG_HD void wu_st_u8(uint8_t* p, uint8_t v, bool coop)
{
	if (!coop || __lane_id() == 0) *p = v;
}
#else
ABG_HD void wu_st_coherent(uint64_t* p, uint64_t v, bool) { st_coherent(p, v); }
ABG_HD void wu_st_u32(uint32_t* p, uint32_t v, bool) { *p = v; }
ABG_HD void wu_st_u8(uint8_t* p, uint8_t v, bool) { *p = v; }
ABG_HD uint64_t wu_cas_u64(uint64_t* p, uint64_t e, uint64_t v, bool) { return cas_u64(p, e, v); }
ABG_HD uint32_t wu_atomic_min_u32(uint32_t* p, uint32_t v, bool) { return atomic_min_u32(p, v); }
ABG_HD uint32_t wu_atomic_add_u32(uint32_t* p, uint32_t v, bool) { return atomic_add_u32(p, v); }
ABG_HD uint64_t wu_atomic_add_u64(uint64_t* p, uint64_t v, bool) { return atomic_add_u64(p, v); }
#endif

// ---------------------------------------------------------------- read-guided bulk steps
// A unitig walk is a dependent chain -- every step needs the previous head -- but where the path
// runs along a READ the chain is known ahead: the read's next k-mers are a prediction of the next
// heads, and whether each predicted head really is the unique continuation can be checked for all
// of them at once (the same 8 x H probes per vertex the step-by-step walk would issue, one vertex
// per lane).  The guide table maps the canonical hash of a k-mer of a sample of the reads to where
// that k-mer sits in the packed reads.  It is a direct-mapped, lossy table of HINTS: a wrong, stale
// or missing entry only costs time, because every hint is verified against the head's k-mer and
// every predicted step against the solid filter before it is taken (walk_bulk).
struct Guide {
	const uint64_t* tab;    // [mask + 1] hints (0 = none), see guide_pack
	uint64_t mask;
	const uint32_t* words;  // the packed reads the hints point into
	uint64_t nwords;
	// What a bulk step found out about the k-mer starting at base b of `words`, kept for the next walker that comes along it
	// (a unitig is walked by every read of the batch that lies on it, 5-6 times over): one byte per base position, 0 = nothing
	// known yet, else SV_VALID | SV_SIMPLE (exactly one neighbour in the solid filter on either side) | the base of the one
	// after it (bits 1-2) and of the one before it (bits 3-4), both as the READ runs.  A fact about the k-mer and the solid
	// filter only, so whoever wrote it wrote the same; NULL: nothing is kept (no room for a byte per base).
	uint8_t* seen;
};
constexpr unsigned SV_VALID = 0x80u, SV_SIMPLE = 0x01u;
constexpr uint32_t GUIDE_MAX_NK = 256; // k-mers of a sequence that can serve as a guide
ABG_HD uint64_t guide_slot(uint64_t hm, uint64_t mask)
{
	uint64_t x = hm * 0x9E3779B97F4A7C15ULL;
	x ^= x >> 29;
	return x & mask;
}
ABG_HD uint32_t guide_tag(uint64_t hm) { return (uint32_t)(hm >> 56); }
// valid bit | tag (8) | k-mers of the read - 1 (8) | k-mer index (8) | word offset of the read (39)
ABG_HD uint64_t guide_pack(uint64_t woff, uint32_t pos, uint32_t nk, uint32_t tag)
{
	return (1ULL << 63) | ((uint64_t)(tag & 0xFFu) << 55) | ((uint64_t)(nk - 1) << 47) | ((uint64_t)pos << 39) | woff;
}
constexpr uint64_t GUIDE_MAX_WOFF = (1ULL << 39) - 1;
constexpr uint32_t BULK_LANES = 64, BULK_MIN = 4;
struct BulkScratch {
	VKey key[BULK_LANES];         // identity of every predicted vertex
	uint8_t good[BULK_LANES];     // the vertex is new to the walker, simple, and continues as predicted
	uint8_t fbase[BULK_LANES];    // its one neighbour ahead
	uint32_t dup[2 * BULK_LANES]; // open-addressing set of the chunk's identities (lane + 1)
	uint32_t dupstop;             // first position that repeats an earlier vertex of the chunk
	uint32_t full;                // the vertex table has no room
	uint64_t hw[MAX_NW], hfh, hrh, hdf, hdr; // hand-over of the new head (hdf, hdr: its masked-out terms, spaced seed)
	// chain_bulk: where each of the (up to four) branches of a successor() call stands
	struct Chain { uint64_t w[MAX_NW], fh, rh, df, dr; uint32_t depth, state; } chain[4];
};
enum { CB_ACTIVE = 0,    // still a plain chain at (vertex, depth): the guide has nothing more to say
       CB_TRUE = 1,      // trueBranch answers true
       CB_NOT_CHAIN = 2, // a vertex with no or several neighbours ahead: the general search decides
       CB_NONE = 3,      // not examined
       CB_FALSE = 4 };   // a plain dead-end tip, `depth` edges long: trueBranch answers false (see chain_true_branches)
enum { WSTAT_BULK_CALLS = 0, WSTAT_BULK_STEPS, WSTAT_LIN_STEPS, WSTAT_CHAIN_STEPS, WSTAT_MEMO_HITS, WSTAT_MEMO_ADDS,
       WSTAT_OVF_POOL, WSTAT_OVF_RECS, // walkers that ran out of contig pool / contig records (the host grows what ran out)
       WSTAT_CLS_COVERED,              // reads whose classification took k-mers from the archive of committed contigs (ContigArchive)
       WSTAT_CLS_DECIDED,              // ... of which this many got their whole verdict there, look-aheads included (arc_ends_decided)
       WSTAT_N = 16 };

// ------------------------------------------------------- memo of successor()
// successor(u, dir) with its iterative deepening over trueBranch searches (ExtendPath.h:314-362) is
// a pure function of the oriented vertex, the direction and the solid filter, and by far the most
// expensive thing a walker does where the graph is tangled -- a collapsed repeat at several times
// the coverage carries a thicket of recurring sequencing errors, and every read of it searches the
// same thicket.  Answers (code and, for a unique successor, its base) are therefore kept in a
// device-wide table for as long as the solid filter stays as it is: open addressing, entries only
// ever added, bounded probing; a full neighbourhood just means the answer is not kept.
struct SuccMemo {
	uint64_t* k0;   // [cap] u.fh        (MEMO_EMPTY when free)
	uint64_t* k1;   // [cap] u.rh
	uint64_t* val;  // [cap] 0 until complete, then 1 << 63 | dir << 8 | code << 4 | base
	uint64_t mask;  // cap - 1; k0 == NULL: no memo
};
constexpr uint64_t MEMO_EMPTY = ~0ULL;
constexpr unsigned MEMO_PROBES = 8;
ABG_HD uint64_t memo_slot(const SuccMemo& m, uint64_t fh, uint64_t rh, int dir)
{
	uint64_t x = (fh ^ (rh * 0x9E3779B97F4A7C15ULL)) + (uint64_t)dir;
	x ^= x >> 31; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 29;
	return x & m.mask;
}
// -1: not there; else code << 4 | base
ABG_HD int memo_find(const SuccMemo& m, uint64_t fh, uint64_t rh, int dir)
{
	if (fh == MEMO_EMPTY || rh == MEMO_EMPTY) return -1;
	uint64_t s = memo_slot(m, fh, rh, dir);
	for (unsigned i = 0; i < MEMO_PROBES; i++, s = (s + 1) & m.mask) {
		const uint64_t a = ld_coherent(&m.k0[s]);
		if (a == MEMO_EMPTY) return -1;
		if (a != fh) continue;
		const uint64_t v = ld_coherent(&m.val[s]);
		if (!(v >> 63) || (int)((v >> 8) & 1u) != dir) continue; // (an entry still being written counts as absent)
		if (ld_coherent(&m.k1[s]) != rh) continue;
		return (int)(v & 0xFFu);
	}
	return -1;
}
ABG_HD void memo_add(const SuccMemo& m, uint64_t fh, uint64_t rh, int dir, unsigned code, unsigned base, bool coop)
{
	if (fh == MEMO_EMPTY || rh == MEMO_EMPTY) return;
	uint64_t s = memo_slot(m, fh, rh, dir);
	for (unsigned i = 0; i < MEMO_PROBES; i++, s = (s + 1) & m.mask) {
		const uint64_t cur = wu_cas_u64(&m.k0[s], MEMO_EMPTY, fh, coop);
		if (cur == MEMO_EMPTY) {
			wu_st_coherent(&m.k1[s], rh, coop);
			wu_st_coherent(&m.val[s], (1ULL << 63) | ((uint64_t)(dir & 1) << 8) | ((uint64_t)code << 4) | base, coop);
			return;
		}
		if (cur != fh) continue;
		const uint64_t v = ld_coherent(&m.val[s]);
		if ((v >> 63) && (int)((v >> 8) & 1u) == dir && ld_coherent(&m.k1[s]) == rh) return; // already there
	}
}

// ------------------------------------------------------------ search scratch
// Explicit stacks for the reference's recursive searches.  One SearchScratch per
// concurrently running searcher (GPU thread); capacities are fixed at launch and
// overflow is reported (never silently truncated).
template <int NW>
struct TBFrame {        // one active call of trueBranch (ExtendPath.h:174-244)
	Vtx<NW> v;          // the vertex this call inserted into `visited`
	// (the vertex this call came from -- skipped when changing direction -- is the vertex of the call
	// below it on the stack, i.e. the key one slot down; the root call's comes with the search)
	uint16_t depth;
	uint8_t dir;        // direction of this call
	uint8_t stage;      // 0: same-direction children, 1: other-direction children
	uint8_t next;       // next base to try in the current stage
	uint8_t mask_same, mask_other;
	uint8_t have_other; // mask_other computed
};
template <int NW>
struct LAFrame {        // one active call of lookAhead (ExtendPath.h:100-139)
	Vtx<NW> v;
	uint8_t mask, next;
};
constexpr uint32_t WALK_DBG_N = 28; // per-walker profiling counters (ABG_WALK_DEBUG)
constexpr int LA_MAX_VISITED = 1366; // 4^0 + ... + 4^5 + 1
constexpr uint32_t LA_FAST = 96;      // ... of which this many live in a walker's fast memory
// The trueBranch stack is two-tier: the first tbf_cap frames live in fast memory (LDS on
// the device), deeper ones in the global pool.
template <int NW>
struct SearchScratch {
	TBFrame<NW>* tb;       // [tb_cap] frames beyond the fast tier
	VKey* tb_keys;         // [tb_cap] vtx_ident of tb[i].v: what the on-stack test scans
	uint32_t tb_cap;
	bool coop;             // the caller is a whole wavefront in lock step (see solid_mask8)
	TBFrame<NW>* tbf;      // [tbf_cap] fast tier (may be NULL with tbf_cap == 0)
	VKey* tbf_keys;        // [tbk_cap] the on-stack test scans keys only, so more of them than frames are kept in fast memory
	uint32_t tbf_cap, tbk_cap;
	uint32_t overflow;     // set when a stack capacity was exceeded
	uint32_t dbg_calls;    // profiling aid: out-of-line successor() calls and the clock ticks spent in them
	uint64_t dbg_search;
	uint64_t dbg_nodes;    // trueBranch calls entered (frames pushed)
	LAFrame<NW> la_local[FP_TRIM + 1]; // used when no fast memory is available
	MaskCache* mcache;     // neighbour masks of the vertices the searches have looked at (NULL: none)
	SuccMemo memo;         // answers of successor() shared by all walkers; k0 == NULL: off
	uint32_t n_memo_hits, n_memo_adds;
	uint64_t* wstats;      // the engine's work counters; may be NULL
	Guide guide;           // read-guided chains (chain_bulk); tab == NULL: off
	BulkScratch* bulk;
	uint32_t n_chain_steps; // chain vertices settled by chain_bulk (work counter)
	uint16_t chain_d[4];    // chain_true_branches: per branch proven FALSE (a plain dead-end tip), the deepest call of its search
	uint64_t dbg_chain;    // profiling aid: clock ticks in chain_true_branches (when dbg_on)
	uint32_t dbg_on;
	uint64_t dbg_la; uint32_t dbg_la_calls; // ... in the lookAhead calls of trueBranch
	uint64_t dbg_mask, dbg_memo; uint32_t dbg_mask_n; // ... waiting for the neighbour masks of trueBranch's vertices (and how many probe rounds), in the memo
	LAFrame<NW>* la;       // [FP_TRIM + 1] lookAhead frames (LDS on the device: private arrays indexed at
	                       // run time would live in per-lane scratch, 64 copies per cooperative wave)
	VKey* la_visited;      // [LA_MAX_VISITED] (global memory)
	VKey* la_fast;         // [la_fast_cap] the first entries of lookAhead's visited set, in fast memory (may be NULL / 0)
	uint32_t la_fast_cap;
};

// lookAhead (ExtendPath.h:100-161): is there a path of >= `limit` further vertices
// from `start` in direction `dir`?  Depth-first, `visited` shared by the whole search
// and never erased, neighbours tried in A,C,G,T order.
template <int NW, bool COOP>
ABG_HDX bool look_ahead_t(const Params& p_in, const uint8_t* __restrict__ cnt_in, const Vtx<NW>& start,
    int dir, unsigned limit, SearchScratch<NW>& sc)
{
	const ParamsView<COOP> pview(p_in);
	const Params& p = pview.get();
	const SeedTabs tabs = seed_tabs(p);
	const uint8_t* __restrict__ cnt = uniptr<COOP>(cnt_in);
	const int sense = ((int)uni32<COOP>((uint32_t)dir) == FORWARD) ? SENSE : ANTISENSE;
	limit = uni32<COOP>(limit);
	unsigned nv = 0;
	// the visited set: its first entries in fast memory (a search rarely sees more), the rest in global memory
	VKey* const vis_fast = uniptr<COOP>(sc.la_fast);
	const unsigned nfast = vis_fast ? uni32<COOP>(sc.la_fast_cap) : 0u;
	VKey* const vis_slow = uniptr<COOP>(sc.la_visited);
	auto vis = [&](unsigned i) -> VKey& { return i < nfast ? vis_fast[i] : vis_slow[i - nfast]; };
	LAFrame<NW>* const la = uniptr<COOP>(sc.la);
	vis(nv++) = vtx_ident(p, start);
	if (limit == 0) return true;
	if (limit > FP_TRIM) { sc.overflow = 1; return true; }
	int depth = 0;
	la[0].v = start;
	MaskCache* const mcache = uniptr<COOP>(sc.mcache);
	la[0].mask = (uint8_t)nbr_mask_cached<NW, COOP>(p, tabs, cnt, start, sense, mcache);
	la[0].next = 0;
	while (depth >= 0) {
		LAFrame<NW>& f = la[depth];
		unsigned nx = uni32<COOP>((uint32_t)f.next);
		const unsigned fm = uni32<COOP>((uint32_t)f.mask);
		while (nx < 4 && !((fm >> nx) & 1u)) nx++;
		if (nx >= 4) { depth--; continue; }
		const unsigned b = nx;
		f.next = (uint8_t)(nx + 1);
		Vtx<NW> fv;
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) fv.s.w[j] = uni64<COOP>(f.v.s.w[j]);
		fv.fh = uni64<COOP>(f.v.fh); fv.rh = uni64<COOP>(f.v.rh);
		vtx_set_d(fv, uni64<COOP>(vtx_df(f.v)), uni64<COOP>(vtx_dr(f.v)));
		const Vtx<NW> w = nbr_vertex_lean(p, tabs, fv, sense, b);
		const VKey wk = vtx_ident(p, w);
		// visited.find(w): cooperative callers spread the scan over the lanes
		bool seen = false;
		for (unsigned i = COOP ? lane_id() : 0u; i < nv; i += (COOP ? 64u : 1u)) seen = seen | key_equal(vis(i), wk);
		if (COOP) seen = wave_any(seen);
		if (seen) continue;
		// recursive call lookAhead(w, depth + 1)
		if (nv < (unsigned)LA_MAX_VISITED) vis(nv++) = wk;
		else sc.overflow = 1;
		if ((unsigned)(depth + 1) >= limit) return true;
		depth++;
		la[depth].v = w;
		la[depth].mask = (uint8_t)nbr_mask_cached<NW, COOP>(p, tabs, cnt, w, sense, mcache);
		la[depth].next = 0;
	}
	return false;
}
template <int NW>
ABG_HD bool look_ahead(const Params& p, const uint8_t* __restrict__ cnt, const Vtx<NW>& start,
    int dir, unsigned limit, SearchScratch<NW>& sc)
{
	return sc.coop ? look_ahead_t<NW, true>(p, cnt, start, dir, limit, sc)
	               : look_ahead_t<NW, false>(p, cnt, start, dir, limit, sc);
}

// lookAhead for a caller that is one lane among 64 doing the same for other vertices (the classification's two blunt-end tests
// per read, bloom-dbg.h:489-532): the same search as look_ahead_t -- depth first, bases in A, C, G, T order, one visited set for
// the whole search -- with nothing in memory.  look_ahead_t keeps its frames, its visited set and the caller's SearchScratch in
// scratch / global memory and asks the filter one probe at a time (solid_contains stops at the first miss): some seventy
// dependent round trips per search, which is what a wave of classifications spent most of its time waiting for.  Here the
// vertex is rolled forwards and BACK (a level keeps the base that left, not the vertex), the five levels' {mask, next
// neighbour, base that left} sit in one word, the visited set in LA_REG_KEYS register pairs, and the 4 x H probes of a vertex's
// neighbours go out together: one round trip per vertex entered.  Returns 0 / 1, or 2 when the visited set outgrew the
// registers (the caller then runs look_ahead_t, which gives the same answer from the start).  Plain builds only (no spaced seed).
constexpr unsigned LA_REG_KEYS = 8;
template <int NW>
ABG_HD unsigned nbr_mask_wide(const Params& p, const SeedTabs& t, const uint8_t* __restrict__ cnt, const Vtx<NW>& v, int sense)
{
	uint64_t fb, rb, h[4];
	nbr_base(t, v, p.k, sense, fb, rb);
#pragma unroll
	for (unsigned q = 0; q < 4; q++) {
		uint64_t fh, rh;
		nbr_hash(t, sense, fb, rb, q, fh, rh);
		h[q] = rh < fh ? rh : fh;
	}
	unsigned ok = 0xFu;
	for (unsigned base = 0; base < p.nh; base += 4) {
		unsigned c[4][4];
#pragma unroll
		for (unsigned q = 0; q < 4; q++) {
#pragma unroll
			for (unsigned j = 0; j < 4; j++) c[q][j] = probe_c(p, cnt, pos_i(p, h[q], base + j < p.nh ? base + j : 0u));
		}
#pragma unroll
		for (unsigned q = 0; q < 4; q++) {
#pragma unroll
			for (unsigned j = 0; j < 4; j++) if (c[q][j] < p.kc) ok &= ~(1u << q);
		}
	}
	return ok;
}
template <int NW>
ABG_HD unsigned look_ahead_reg(const Params& p, const uint8_t* __restrict__ cnt, const Vtx<NW>& start, int dir)
{
	static_assert(!MASKED_BUILD<NW>, "plain builds only");
	const SeedTabs tabs = seed_tabs(p);
	const int sense = dir == FORWARD ? SENSE : ANTISENSE, back = dir == FORWARD ? ANTISENSE : SENSE;
	const unsigned limit = FP_TRIM;
	VKey keys[LA_REG_KEYS];
	unsigned nv = 1;
	keys[0] = vtx_ident(p, start);
#pragma unroll
	for (unsigned i = 1; i < LA_REG_KEYS; i++) keys[i] = keys[0];
	Vtx<NW> cur = start;
	// level d of the search in bits [9 d, 9 d + 9) of `st`: neighbour mask (4), next neighbour to try (3), the base that left when the level was entered (2)
	uint64_t st = nbr_mask_wide(p, tabs, cnt, cur, sense);
	int depth = 0;
	while (depth >= 0) {
		const unsigned sh = 9u * (unsigned)depth;
		const unsigned fm = (unsigned)(st >> sh) & 0xFu;
		unsigned nx = (unsigned)(st >> (sh + 4)) & 7u;
		while (nx < 4 && !((fm >> nx) & 1u)) nx++;
		if (nx >= 4) {
			// back to the level above: the vertex rolled back by the base that left
			if (depth > 0) vtx_shift(p, cur, back, (unsigned)(st >> (sh + 7)) & 3u);
			depth--;
			continue;
		}
		st = (st & ~(7ULL << (sh + 4))) | ((uint64_t)(nx + 1) << (sh + 4));
		const unsigned left = sense == SENSE ? kmer_get(cur.s, 0) : kmer_get(cur.s, p.k - 1);
		const Vtx<NW> w = nbr_vertex_lean(p, tabs, cur, sense, nx);
		const VKey wk = vtx_ident(p, w);
		bool seen = false;
#pragma unroll
		for (unsigned i = 0; i < LA_REG_KEYS; i++) seen = seen | (i < nv && key_equal(keys[i], wk));
		if (seen) continue;
		if (nv >= LA_REG_KEYS) return 2;
#pragma unroll
		for (unsigned i = 0; i < LA_REG_KEYS; i++) if (i == nv) keys[i] = wk;
		nv++;
		if ((unsigned)(depth + 1) >= limit) return 1;
		depth++;
		cur = w;
		const uint64_t lvl = (uint64_t)nbr_mask_wide(p, tabs, cnt, cur, sense) | ((uint64_t)left << 7);
		st = (st & ~(0x1FFULL << (9u * (unsigned)depth))) | (lvl << (9u * (unsigned)depth));
	}
	return 0;
}

// trueBranch (ExtendPath.h:174-261).  Edge (u -> v) walked in direction `dir`.  Every
// `return true` of the recursion propagates to the root, and `visited` holds exactly the
// vertices of the active calls (inserted on entry, erased on a false return), so the
// recursion is a depth-first search over an explicit frame stack that stops at the first
// call that would return true.
template <int NW, bool COOP>
ABG_HDX bool true_branch_t(const Params& p_in, const uint8_t* __restrict__ cnt_in, const Vtx<NW>& u0,
    const Vtx<NW>& v0, int dir0, unsigned trim_in, SearchScratch<NW>& sc, unsigned* max_depth)
{
	unsigned maxd = 0; // deepest call entered (reported when the answer is false, see successor_m)
	// wave-uniform copies of everything the loop reads (scalar registers for cooperative callers)
	const ParamsView<COOP> pview(p_in);
	const Params& p = pview.get();
	const SeedTabs tabs = seed_tabs(p);
	const uint8_t* __restrict__ cnt = uniptr<COOP>(cnt_in);
	const unsigned trim = uni32<COOP>(trim_in);
	TBFrame<NW>* const tbf = uniptr<COOP>(sc.tbf);
	TBFrame<NW>* const tbs = uniptr<COOP>(sc.tb);
	VKey* const tbf_keys = uniptr<COOP>(sc.tbf_keys);
	VKey* const tbs_keys = uniptr<COOP>(sc.tb_keys);
	const uint32_t tbf_cap = uni32<COOP>(sc.tbf_cap), tbk_cap = uni32<COOP>(sc.tbk_cap);
	const int cap = (int)(tbf_cap + uni32<COOP>(sc.tb_cap));
	MaskCache* const mcache = uniptr<COOP>(sc.mcache);
	auto frame = [&](int i) -> TBFrame<NW>& { return (uint32_t)i < tbf_cap ? tbf[i] : tbs[(uint32_t)i - tbf_cap]; };
	auto keyat = [&](int i) -> VKey& { return (uint32_t)i < tbk_cap ? tbf_keys[i] : tbs_keys[(uint32_t)i - tbk_cap]; };
	auto uniform_vtx = [&](const Vtx<NW>& x) {
		Vtx<NW> r;
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) r.s.w[j] = uni64<COOP>(x.s.w[j]);
		r.fh = uni64<COOP>(x.fh); r.rh = uni64<COOP>(x.rh);
		vtx_set_d(r, uni64<COOP>(vtx_df(x)), uni64<COOP>(vtx_dr(x)));
		return r;
	};
	int top = -1;
	// "call" trueBranch(u0 -> v0, depth 0, dir0)
	Vtx<NW> cv = uniform_vtx(v0);
	VKey cuk = vtx_ident(p, uniform_vtx(u0)); // identity of the vertex the call comes from
	const VKey root_uk = cuk;
	unsigned cdepth = 0;
	int cdir = (int)uni32<COOP>((uint32_t)dir0);
	for (;;) {
#if defined(__HIP_DEVICE_COMPILE__)
		const uint64_t te0 = sc.dbg_on ? wall_clock64() : 0;
#endif
		// ---- entry of a call (u, v=cv, depth=cdepth, dir=cdir)
		// visited.find(v): scan the keys of the active calls (no early exit: the loads pipeline)
		bool on_stack = false;
		const VKey ck = vtx_ident(p, cv);
		{
			// cooperative callers spread the scan over the lanes
			const int first = COOP ? (int)lane_id() : 0, step = COOP ? 64 : 1;
			for (int i = first; i <= top; i += step) {
				const VKey kk = keyat(i);
				on_stack = on_stack | ((kk.fh == ck.fh) & (kk.rh == ck.rh));
			}
			if (COOP) on_stack = wave_any(on_stack);
		}
		if (on_stack) return true;
		if (cdepth >= trim) return true;
		if (top + 1 >= cap) { sc.overflow = 1; return true; }
		maxd = cdepth > maxd ? cdepth : maxd;
		top++;
		sc.dbg_nodes++;
		{
			keyat(top) = ck;
			TBFrame<NW>& f = frame(top);
			f.v = cv;
			f.depth = (uint16_t)cdepth; f.dir = (uint8_t)cdir; f.stage = 0; f.next = 0;
			f.have_other = 0; f.mask_other = 0;
#if defined(__HIP_DEVICE_COMPILE__)
			const uint64_t tm0 = sc.dbg_on ? wall_clock64() : 0;
			if (sc.dbg_on) sc.dbg_memo += tm0 - te0; // (ABG_WALK_DEBUG: the entry of a call up to its probe round, booked under "memo")
#endif
			const unsigned both = mcache ? 0x100u : nbr_mask_both<NW, COOP>(p, tabs, cnt, cv);
			if (both < 0x100u) {
				// (both directions from the one round trip: the way back from a dead end does not wait again)
				f.mask_same = (uint8_t)(cdir == FORWARD ? (both & 0xFu) : (both >> 4));
				f.mask_other = (uint8_t)(cdir == FORWARD ? (both >> 4) : (both & 0xFu));
				f.have_other = 1;
			} else
			f.mask_same = (uint8_t)nbr_mask_cached<NW, COOP>(p, tabs, cnt, cv, cdir == FORWARD ? SENSE : ANTISENSE, mcache);
#if defined(__HIP_DEVICE_COMPILE__)
			if (sc.dbg_on) { sc.dbg_mask += wall_clock64() - tm0; sc.dbg_mask_n++; }
#endif
		}
		// ---- resume frames until one of them makes a new call
		bool called = false;
		while (top >= 0 && !called) {
			TBFrame<NW>& f = frame(top);
			// (the frame in ONE copy: its fields read one by one are a dozen loads each waited for on the spot)
			const TBFrame<NW> fc = f;
			const Vtx<NW> fv = uniform_vtx(fc.v);
			const unsigned fdepth = uni32<COOP>((uint32_t)fc.depth);
			const int fdir = (int)uni32<COOP>((uint32_t)fc.dir);
			const int sense = (fdir == FORWARD) ? SENSE : ANTISENSE;
			unsigned f_next = uni32<COOP>((uint32_t)fc.next), f_mask_other = uni32<COOP>((uint32_t)fc.mask_other);
			if (uni32<COOP>((uint32_t)fc.stage) == 0) {
				unsigned nx = f_next;
				const unsigned ms = uni32<COOP>((uint32_t)fc.mask_same);
				while (nx < 4 && !((ms >> nx) & 1u)) nx++;
				if (nx < 4) {
					unsigned b = nx++;
					f.next = (uint8_t)nx;
					cuk = keyat(top); cuk.fh = uni64<COOP>(cuk.fh); cuk.rh = uni64<COOP>(cuk.rh);
					cv = nbr_vertex_lean(p, tabs, fv, sense, b);
					cdepth = fdepth + 1u;
					cdir = fdir;
					called = true;
					break;
				}
				f.next = (uint8_t)nx;
				// same-direction children exhausted: may we change direction?
				// (depth >= fpTrim || lookAhead(v, dir, fpTrim), ExtendPath.h:208,230)
				bool flip = fdepth >= FP_TRIM;
				if (!flip) {
#if defined(__HIP_DEVICE_COMPILE__)
					const uint64_t tl0 = sc.dbg_on ? wall_clock64() : 0;
#endif
					flip = look_ahead_t<NW, COOP>(p, cnt, fv, fdir, FP_TRIM, sc);
#if defined(__HIP_DEVICE_COMPILE__)
					if (sc.dbg_on) { sc.dbg_la += wall_clock64() - tl0; sc.dbg_la_calls++; }
#endif
				}
				if (!flip) { top--; continue; } // visited.erase(v); return false
				f.stage = 1;
				f.next = 0; f_next = 0;
				if (!uni32<COOP>((uint32_t)fc.have_other)) {
#if defined(__HIP_DEVICE_COMPILE__)
					const uint64_t tm0 = sc.dbg_on ? wall_clock64() : 0;
#endif
					f_mask_other = nbr_mask_cached<NW, COOP>(p, tabs, cnt, fv, fdir == FORWARD ? ANTISENSE : SENSE, mcache);
					f.mask_other = (uint8_t)f_mask_other;
#if defined(__HIP_DEVICE_COMPILE__)
					if (sc.dbg_on) { sc.dbg_mask += wall_clock64() - tm0; sc.dbg_mask_n++; }
#endif
					f.have_other = 1;
				}
			}
			// stage 1: other-direction children, skipping the vertex we came from
			{
				const int osense = (fdir == FORWARD) ? ANTISENSE : SENSE;
				const int odir = (fdir == FORWARD) ? REVERSE : FORWARD;
				bool made = false;
				unsigned nx = f_next;
				const unsigned mo = f_mask_other;
				const VKey uk = top > 0 ? keyat(top - 1) : root_uk;
				const uint64_t ufh = uni64<COOP>(uk.fh), urh = uni64<COOP>(uk.rh);
				while (nx < 4) {
					unsigned b = nx++;
					if (!((mo >> b) & 1u)) continue;
					const Vtx<NW> w = nbr_vertex_lean(p, tabs, fv, osense, b);
					const VKey wk = vtx_ident(p, w);
					if ((wk.fh == ufh) & (wk.rh == urh)) continue; // source(*iei) == u
					cuk = keyat(top); cuk.fh = uni64<COOP>(cuk.fh); cuk.rh = uni64<COOP>(cuk.rh);
					cv = w; cdepth = 0; cdir = odir;
					made = true;
					break;
				}
				f.next = (uint8_t)nx;
				if (made) { called = true; break; }
				top--; // visited.erase(v); return false
			}
		}
		if (!called) { *max_depth = maxd; return false; } // root call returned false
	}
}
template <int NW>
ABG_HD bool true_branch(const Params& p, const uint8_t* __restrict__ cnt, const Vtx<NW>& u0,
    const Vtx<NW>& v0, int dir0, unsigned trim, SearchScratch<NW>& sc, unsigned* max_depth)
{
	return sc.coop ? true_branch_t<NW, true>(p, cnt, u0, v0, dir0, trim, sc, max_depth)
	               : true_branch_t<NW, false>(p, cnt, u0, v0, dir0, trim, sc, max_depth);
}

// successor (ExtendPath.h:314-362): iterative deepening over the branch-length
// threshold i = 0,1,2,4,...,trim.  Returns the code and (for LENGTH_LIMIT) the unique
// successor; for AMBI_OUT the last true branch found, for DEAD_END `u` itself.
// `mask`, `nfh`, `nrh` are the neighbour mask / hashes of `u` in direction `dir`.
// The level-0 decision of successor(): with at most one neighbour present the answer is
// DEAD_END or that neighbour; returns -1 when deeper levels must be consulted.
template <int NW>
ABG_HD int successor_fast(const Params& p, const Vtx<NW>& u, int dir, unsigned mask,
    const uint64_t nfh[4], const uint64_t nrh[4], Vtx<NW>& vout)
{
	if (mask == 0) { vout = u; return ER_DEAD_END; }
	if (mask & (mask - 1)) return -1;
	unsigned b = (mask & 1u) ? 0u : (mask & 2u) ? 1u : (mask & 4u) ? 2u : 3u;
	uint64_t fh = nfh[0], rh = nrh[0];
#pragma unroll
	for (unsigned q = 1; q < 4; q++) { fh = (b == q) ? nfh[q] : fh; rh = (b == q) ? nrh[q] : rh; }
	vout = make_neighbour(p, u, (dir == FORWARD) ? SENSE : ANTISENSE, b, fh, rh);
	return ER_LENGTH_LIMIT;
}
// chain_true_branches' descent along ONE branch, a read's worth of vertices at a time.  The
// sequential rule (see chain_true_branches): at vertex v, `depth` edges down the branch -- v among
// the chain's earlier vertices or depth >= trim: true; otherwise v joins the chain, and with
// exactly one neighbour ahead the descent goes on there, else the branch is no plain chain.  Where
// a guide read holds v, its following k-mers predict the next vertices: one lane each computes
// their identities and their neighbours ahead, and the rule is then applied to all of them in
// order.  Updates (v, depth, keys[0, depth)); returns CB_TRUE, CB_NOT_CHAIN, or CB_ACTIVE when the
// guide has no (more) advice for v.  The guide is keyed by the
// k-mers' UNMASKED canonical hashes (the walkers' rolling state), whatever the seed.
template <int NW> struct SearchScratch;
// (state goes in and out through bs.chain[gi]: vertex, depth, and on return the verdict)
template <int NW, bool COOP>
ABG_HDX uint32_t chain_bulk(const Params& p_in, const uint8_t* __restrict__ cnt_in, SearchScratch<NW>& sc,
    const unsigned gi_in, const int sense_in, const unsigned trim_in, VKey* keys_in)
{
	{
	const Params p = uniform_params<COOP>(p_in);
	const uint8_t* __restrict__ cnt = uniptr<COOP>(cnt_in);
	Guide g;
	g.tab = uniptr<COOP>(sc.guide.tab); g.mask = uni64<COOP>(sc.guide.mask);
	g.words = uniptr<COOP>(sc.guide.words); g.nwords = uni64<COOP>(sc.guide.nwords);
	BulkScratch& bs = *uniptr<COOP>(sc.bulk);
	const unsigned gi = uni32<COOP>(gi_in), trim = uni32<COOP>(trim_in);
	const int sense = (int)uni32<COOP>((uint32_t)sense_in);
	VKey* const keys = uniptr<COOP>(keys_in);
	Vtx<NW> v;
#pragma unroll
	for (int j = 0; j < KW<NW>; j++) v.s.w[j] = uni64<COOP>(bs.chain[gi].w[j]);
	v.fh = uni64<COOP>(bs.chain[gi].fh); v.rh = uni64<COOP>(bs.chain[gi].rh);
	vtx_set_d(v, uni64<COOP>(bs.chain[gi].df), uni64<COOP>(bs.chain[gi].dr));
	uint32_t depth_io = uni32<COOP>(bs.chain[gi].depth);
	auto done = [&](uint32_t d, uint32_t st) -> uint32_t {
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) bs.chain[gi].w[j] = v.s.w[j];
		bs.chain[gi].fh = v.fh; bs.chain[gi].rh = v.rh; bs.chain[gi].df = vtx_df(v); bs.chain[gi].dr = vtx_dr(v);
		bs.chain[gi].depth = d; bs.chain[gi].state = st;
		return st;
	};
	const unsigned k = p.k;
	const uint64_t sk0 = p.seed_k[0], sk1 = p.seed_k[1], sk2 = p.seed_k[2], sk3 = p.seed_k[3];
	const uint64_t rk0 = p.seedrc_k[0], rk1 = p.seedrc_k[1], rk2 = p.seedrc_k[2], rk3 = p.seedrc_k[3];
	const uint64_t sm0 = p.seed_km1[0], sm1 = p.seed_km1[1], sm2 = p.seed_km1[2], sm3 = p.seed_km1[3];
	const uint64_t rm0 = p.seedrc_km1[0], rm1 = p.seedrc_km1[1], rm2 = p.seedrc_km1[2], rm3 = p.seedrc_km1[3];
	const uint32_t lane0 = COOP ? lane_id() : 0u, lstep = COOP ? BULK_LANES : 1u;
	uint32_t depth = depth_io;
	for (;;) {
		if (depth >= trim) return done(depth, CB_TRUE);
		const uint64_t hm = v.rh < v.fh ? v.rh : v.fh;
		const uint64_t hint = uni64<COOP>(g.tab[guide_slot(hm, g.mask)]);
		if (!(hint >> 63) || ((uint32_t)(hint >> 55) & 0xFFu) != guide_tag(hm)) break;
		const uint64_t woff = hint & GUIDE_MAX_WOFF;
		const uint32_t pos = (uint32_t)(hint >> 39) & 0xFFu, nk = ((uint32_t)(hint >> 47) & 0xFFu) + 1u;
		if (pos >= nk || woff + ((nk + k - 1 + 15) >> 4) > g.nwords) break;
		bool same = true, anti = true;
		{
			const Kmer<NW> rk = window_kmer<NW>(g.words, woff, pos, k);
			const Kmer<NW> hr = kmer_revcomp_fast(v.s, k);
#pragma unroll
			for (int j = 0; j < KW<NW>; j++) {
				const uint64_t x = uni64<COOP>(rk.w[j]);
				same = same & (x == v.s.w[j]);
				anti = anti & (x == hr.w[j]);
			}
		}
		if (!same && !anti) break;
		const bool up = (sense == SENSE) == same;
		uint32_t n = up ? nk - pos : pos + 1u;
		if (n > BULK_LANES) n = BULK_LANES;
		if (n > trim - depth + 1u) n = trim - depth + 1u;
		if (n < 2) break;
		auto vertex_at = [&](uint32_t l, Kmer<NW>& s, uint64_t& fh, uint64_t& rh) {
			s = window_kmer<NW>(g.words, woff, up ? pos + l : pos - l, k);
			if (!same) s = kmer_revcomp_fast(s, k);
			kmer_hashes(s, k, fh, rh);
		};
		auto nbr = [&](const Kmer<NW>& s, uint64_t fh, uint64_t rh, unsigned b, uint64_t& nfh, uint64_t& nrh) {
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
		if (lane0 == 0) bs.dupstop = n;
		Kmer<NW> my_s; uint64_t my_fh = 0, my_rh = 0;
		// spaced seed: the masked-out terms of the lane's vertex and of its four neighbours ahead
		uint64_t my_df = 0, my_dr = 0, my_ndf = 0, my_ndr = 0;
		auto terms = [&]() {
			if constexpr (MASKED_BUILD<NW>) {
				masked_terms(p, my_s, my_df, my_dr);
				masked_terms_shifted(p, my_s, my_df, my_dr, sense, my_ndf, my_ndr);
			}
		};
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) my_s.w[j] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
		uint64_t st_fh = 0, st_rh = 0; // (see walk_bulk: the chunk's hashes by prefix scans over the read's bases)
		if (COOP) {
			stretch_hashes_wave<NW>(g.words, woff, up ? pos : pos - (n - 1u), n, k, st_fh, st_rh);
			if (!up) { const unsigned src = lane0 < n ? n - 1u - lane0 : lane0; st_fh = shfl64(st_fh, src); st_rh = shfl64(st_rh, src); }
			if (!same) { const uint64_t t = st_fh; st_fh = st_rh; st_rh = t; }
		}
#endif
		for (uint32_t l = lane0; l < n; l += lstep) {
#if defined(__HIP_DEVICE_COMPILE__)
			if (COOP) {
				my_s = window_kmer<NW>(g.words, woff, up ? pos + l : pos - l, k);
				if (!same) my_s = kmer_revcomp_fast(my_s, k);
				my_fh = st_fh; my_rh = st_rh;
			} else
#endif
			vertex_at(l, my_s, my_fh, my_rh);
			terms();
			const VKey key = kmer_ident(p, my_s, my_fh, my_rh, my_df, my_dr);
			bs.key[l] = key;
			unsigned bad = 0; // bit q: the neighbour ahead with base q is not in the solid filter
			for (unsigned base = 0; base < p.nh; base += 4) {
				uint8_t c[4][4];
#pragma unroll
				for (unsigned q = 0; q < 4; q++) {
					uint64_t nfh, nrh;
					nbr(my_s, my_fh, my_rh, q, nfh, nrh);
					nfh ^= my_ndf; nrh ^= my_ndr;
					const uint64_t h = nrh < nfh ? nrh : nfh;
#pragma unroll
					for (unsigned i = 0; i < 4; i++) c[q][i] = (uint8_t)probe_c(p, cnt, pos_i(p, h, base + i < p.nh ? base + i : 0u));
				}
#pragma unroll
				for (unsigned q = 0; q < 4; q++) {
#pragma unroll
					for (unsigned i = 0; i < 4; i++) bad |= (c[q][i] < p.kc ? 1u : 0u) << q;
				}
			}
			bool hit = false; // visited.find(v) among the chain's vertices before this chunk
			for (uint32_t i = 0; i < depth; i++) hit = hit | key_equal(keys[i], key);
			const unsigned cm = ~bad & 0xFu;
			const unsigned fb = (cm & 1u) ? 0u : (cm & 2u) ? 1u : (cm & 4u) ? 2u : 3u; // (the first neighbour ahead, as the search would take them)
			bool pred = true; // the read's next k-mer is a neighbour ahead
			if (l + 1 < n) {
				const uint32_t rb_pos = up ? pos + l + k : pos - l - 1u;
				const unsigned rb = (g.words[woff + (rb_pos >> 4)] >> (2u * (rb_pos & 15u))) & 3u;
				pred = ((cm >> (same ? rb : 3u - rb)) & 1u) != 0;
			}
			bs.good[l] = (uint8_t)((cm != 0 ? 1u : 0u) | (pred ? 2u : 0u) | (hit ? 4u : 0u) | ((cm & (cm - 1)) ? 8u : 0u));
			bs.fbase[l] = (uint8_t)fb;
		}
		wave_sync();
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
		// the rule, in order: T = first position that answers true, B = first one that is no chain
		// vertex, C = first one after which the read stops predicting
		uint32_t T = n, B = n, C = n - 1;
		{
			const uint32_t ds = uni32<COOP>(ld_coherent(&bs.dupstop));
			if (COOP) {
				const unsigned f = lane0 < n ? (unsigned)bs.good[lane0] : 3u;
				const uint64_t bt = wave_ballot(lane0 < n && ((f & 4u) || lane0 == ds || depth + lane0 >= trim));
				const uint64_t bb = wave_ballot(lane0 < n && !(f & 1u));
				const uint64_t bc = wave_ballot(lane0 < n && !(f & 2u));
				if (bt) T = (uint32_t)__builtin_ctzll(bt);
				if (bb) B = (uint32_t)__builtin_ctzll(bb);
				if (bc) C = (uint32_t)__builtin_ctzll(bc);
			} else {
				for (uint32_t l = 0; l < n; l++) {
					const unsigned f = bs.good[l];
					if (T == n && ((f & 4u) || l == ds || depth + l >= trim)) T = l;
					if (B == n && !(f & 1u)) B = l;
					if (C == n - 1 && !(f & 2u)) C = l;
				}
			}
		}
		if (T <= C) { sc.n_chain_steps += T; return done(depth + T, CB_TRUE); }
		if (B <= C) { // (B < C cannot be: a position before C has the read's next k-mer ahead)
			sc.n_chain_steps += B;
			// The read runs into a dead end B edges from the branch's first vertex.  If that is a PLAIN tip -- one neighbour ahead at
			// every vertex before the end, and one behind (the vertex the search came from) wherever trueBranch would turn round:
			// at t_i with i >= fpTrim or, lookAhead on a plain chain, i + fpTrim <= B -- every call of the search answers false and
			// the deepest one was B deep (chain_true_branches has the argument).  Only when the whole tip lies in this chunk.
			if (depth != 0) return done(depth + B, CB_NOT_CHAIN);
			bool ok = true;
			const int osense = (sense == SENSE) ? ANTISENSE : SENSE;
			for (uint32_t l = lane0; l <= B; l += lstep) {
				if (l < B && (bs.good[l] & 8u)) ok = false;
				if (!(l >= FP_TRIM || l + FP_TRIM <= B)) continue;
#if defined(__HIP_DEVICE_COMPILE__)
				if (!COOP)
#endif
				{ vertex_at(l, my_s, my_fh, my_rh); terms(); }
				uint64_t odf = 0, odr = 0;
				if constexpr (MASKED_BUILD<NW>) masked_terms_shifted(p, my_s, my_df, my_dr, osense, odf, odr);
				unsigned obad = 0;
				for (unsigned base = 0; base < p.nh; base += 4) {
					uint8_t c[4][4];
#pragma unroll
					for (unsigned q = 0; q < 4; q++) {
						// (the neighbour behind with base q: nbr() the other way round)
						uint64_t nfh, nrh;
						if (osense == SENSE) {
							const unsigned out = (unsigned)my_s.w[0] & 3u;
							nfh = srol1(my_fh) ^ pick4(out, sk0, sk1, sk2, sk3) ^ seed_of(q);
							nrh = sror1(my_rh ^ seed_of(3u - out)) ^ pick4(q, rm0, rm1, rm2, rm3);
						} else {
							const unsigned out = kmer_get(my_s, k - 1);
							nfh = sror1(my_fh ^ seed_of(out)) ^ pick4(q, sm0, sm1, sm2, sm3);
							nrh = (srol1(my_rh) ^ pick4(out, rk0, rk1, rk2, rk3)) ^ seed_of(3u - q);
						}
						nfh ^= odf; nrh ^= odr;
						const uint64_t h = nrh < nfh ? nrh : nfh;
#pragma unroll
						for (unsigned i = 0; i < 4; i++) c[q][i] = (uint8_t)probe_c(p, cnt, pos_i(p, h, base + i < p.nh ? base + i : 0u));
					}
#pragma unroll
					for (unsigned q = 0; q < 4; q++) {
#pragma unroll
						for (unsigned i = 0; i < 4; i++) obad |= (c[q][i] < p.kc ? 1u : 0u) << q;
					}
				}
				const unsigned om = ~obad & 0xFu;
				if (om == 0 || (om & (om - 1))) ok = false;
			}
			if (COOP) ok = !wave_any(!ok);
			return done(depth + B, ok ? CB_FALSE : CB_NOT_CHAIN);
		}
		sc.n_chain_steps += C + 1;
		// positions 0..C join the chain; the descent goes on at the neighbour ahead of position C
		for (uint32_t l = lane0; l <= C; l += lstep) {
			keys[depth + l] = bs.key[l];
			if (l == C) {
				if (!COOP) { vertex_at(l, my_s, my_fh, my_rh); terms(); }
				uint64_t nfh, nrh;
				const unsigned fb = bs.fbase[l];
				nbr(my_s, my_fh, my_rh, fb, nfh, nrh);
				kmer_shift(my_s, k, sense, fb);
#pragma unroll
				for (int j = 0; j < KW<NW>; j++) bs.hw[j] = my_s.w[j];
				bs.hfh = nfh; bs.hrh = nrh; bs.hdf = my_ndf; bs.hdr = my_ndr;
			}
		}
		wave_sync();
#pragma unroll
		for (int j = 0; j < KW<NW>; j++) v.s.w[j] = uni64<COOP>(bs.hw[j]);
		v.fh = uni64<COOP>(bs.hfh); v.rh = uni64<COOP>(bs.hrh);
		vtx_set_d(v, uni64<COOP>(bs.hdf), uni64<COOP>(bs.hdr));
		depth += C + 1;
	}
	return done(depth, CB_ACTIVE);
	}
}

// trueBranch for the common shape of a real branch: a chain.  While every vertex reached has
// exactly one neighbour ahead, trueBranch's recursion (ExtendPath.h:174-244) is a straight
// descent that answers true as soon as it meets a vertex of the chain again (visited.find) or
// gets `trim` edges deep -- no backtracking, no direction change, no lookAhead.  This follows
// up to four such chains (the branches of `u` in direction `dir` named by `mask`) at once, 16
// lanes each, so that one probe round trip advances all of them; a cooperative caller's
// successor() thereby pays the depth of one branch instead of the sum over the branches, and
// pays it without frame traffic.  Branches proven true are returned in true_mask; any other
// outcome (a vertex with no or several neighbours ahead, more than four hash functions, chains
// longer than the key space) is left to the general search.
template <int NW, bool COOP>
ABG_HDX unsigned chain_true_branches(const Params& p_in, const uint8_t* __restrict__ cnt_in, const Vtx<NW>& u,
    const int dir_in, const unsigned trim_in, const unsigned mask_in, SearchScratch<NW>& sc)
{
	if (p_in.nh > 4) return 0;
	const Params p = uniform_params<COOP>(p_in);
	const uint8_t* __restrict__ cnt = uniptr<COOP>(cnt_in);
	const int dir = (int)uni32<COOP>((uint32_t)dir_in);
	const unsigned trim = uni32<COOP>(trim_in), mask = uni32<COOP>(mask_in);
	const int sense = (dir == FORWARD) ? SENSE : ANTISENSE;
	const SeedTabs tabs = seed_tabs(p);
	// the fast tier of the trueBranch stack is idle here: it holds the chains' vertex identities
	VKey* const keys = uniptr<COOP>(sc.tbf_keys);
	const uint32_t key_space = (uint32_t)(((char*)uniptr<COOP>(sc.tbf + sc.tbf_cap) - (char*)keys) / sizeof(VKey));
	const uint32_t per_chain = key_space / 4;
	if (!keys || trim > per_chain) return 0;
	const unsigned lane = COOP ? lane_id() : 0u;
	const unsigned ngroups = COOP ? 4u : 1u;
	BulkScratch* const bulk = uniptr<COOP>(sc.bulk);
	const bool use_guide = uniptr<COOP>(sc.guide.tab) != nullptr && bulk != nullptr;
	unsigned true_mask = 0;
	// serial callers take the branches one after the other (group 0); cooperative ones all at once
	for (unsigned first = 0; first < 4; first += ngroups) {
		const unsigned grp = COOP ? lane >> 4 : 0u, sub = COOP ? lane & 15u : 0u;
		// the branch of this group: the (first + grp)-th set bit of mask
		unsigned my_b = 4, seen = 0;
#pragma unroll
		for (unsigned b = 0; b < 4; b++)
			if ((mask >> b) & 1u) { if (seen == first + grp) my_b = b; seen++; }
		if (COOP ? first >= seen : my_b >= 4) break;
		bool active = my_b < 4;
		bool is_true = false;
		Vtx<NW> v = u;
		if (active) {
			uint64_t fb, rb, fh, rh;
			nbr_base(tabs, u, p.k, sense, fb, rb);
			nbr_hash(tabs, sense, fb, rb, my_b, fh, rh);
			v = make_neighbour(p, u, sense, my_b, fh, rh);
		}
		VKey* const mykeys = keys + (uint64_t)grp * per_chain;
		unsigned depth = 0;
		// a plain tip (see below): every vertex so far has one neighbour ahead; ... and, where the search would turn round, one behind
		bool plain = true, is_false = false, behind_ok = true;
		unsigned behind_lo = 0; // bit i: the vertex at depth i < FP_TRIM has exactly one neighbour behind
		// The walk below is a depth-first search over the neighbours AHEAD only (no turning round, no lookAhead): at a fork it takes
		// the first neighbour, as trueBranch would, and remembers the others; at a dead end it goes back to the last fork that has
		// one left.  Any walk of `trim` edges it finds answers true (trueBranch is an OR over such walks); when it runs out of
		// forks or of its budget of steps nothing is known and the general search decides.  The forks live behind the chain's keys.
		struct Fork { Vtx<NW> v; uint32_t depth, mask; };
		Fork* const forks = (Fork*)(mykeys + trim);
		const unsigned fork_cap = (unsigned)(((uint64_t)(per_chain - trim) * sizeof(VKey)) / sizeof(Fork)) < 8u ? (unsigned)(((uint64_t)(per_chain - trim) * sizeof(VKey)) / sizeof(Fork)) : 8u;
		unsigned nforks = 0, steps = 0;
		const unsigned budget = 4u * trim + 64u;
		if (use_guide) {
			// read-guided descent first: one branch at a time, the whole wave on it (chain_bulk);
			// the lock-step loop below carries on from wherever the guide leaves a branch
			BulkScratch& bs = *bulk;
			for (unsigned g = 0; g < ngroups; g++) {
				unsigned gb = 4, seen3 = 0;
#pragma unroll
				for (unsigned b = 0; b < 4; b++)
					if ((mask >> b) & 1u) { if (seen3 == first + g) gb = b; seen3++; }
				bs.chain[g].state = CB_NONE;
				if (gb >= 4) continue;
				{
					uint64_t fb, rb, fh, rh;
					nbr_base(tabs, u, p.k, sense, fb, rb);
					nbr_hash(tabs, sense, fb, rb, gb, fh, rh);
					const Vtx<NW> gv = make_neighbour(p, u, sense, gb, fh, rh);
#pragma unroll
					for (int j = 0; j < KW<NW>; j++) bs.chain[g].w[j] = gv.s.w[j];
					bs.chain[g].fh = gv.fh; bs.chain[g].rh = gv.rh; bs.chain[g].df = vtx_df(gv); bs.chain[g].dr = vtx_dr(gv);
					bs.chain[g].depth = 0;
				}
				wave_sync();
				chain_bulk<NW, COOP>(p_in, cnt_in, sc, g, sense, trim, keys + (uint64_t)g * per_chain);
			}
			wave_sync();
			if (active) {
				const BulkScratch::Chain& cs = bs.chain[grp];
				const uint32_t st = cs.state;
				if (st != CB_NONE) {
#pragma unroll
					for (int j = 0; j < KW<NW>; j++) v.s.w[j] = cs.w[j];
					v.fh = cs.fh; v.rh = cs.rh; vtx_set_d(v, cs.df, cs.dr); depth = cs.depth;
					if (depth) plain = false; // (the guide followed a read: nobody looked behind those vertices)
					if (st == CB_TRUE) { is_true = true; active = false; }
					else if (st == CB_FALSE) { is_false = true; active = false; }
					else if (st == CB_NOT_CHAIN) {
						// (the read led into a dead end that is no plain tip: the walks below start over from the branch's first vertex)
						uint64_t fb0, rb0, fh0, rh0;
						nbr_base(tabs, u, p.k, sense, fb0, rb0);
						nbr_hash(tabs, sense, fb0, rb0, my_b, fh0, rh0);
						v = make_neighbour(p, u, sense, my_b, fh0, rh0);
						depth = 0;
					}
				}
			}
			wave_sync();
		}
		unsigned from_depth = depth; // where the guide left the walk
		while (COOP ? wave_any(active) : active) {
			if (active) {
				const VKey key = vtx_ident(p, v);
				// visited.find(v): the vertices of this chain so far
				bool hit = false;
				for (unsigned i = sub; i < depth; i += (COOP ? 16u : 1u)) hit = hit | key_equal(mykeys[i], key);
				if (COOP) hit = ((wave_ballot(hit) >> (16 * grp)) & 0xFFFFull) != 0;
				if (hit || depth >= trim) { is_true = true; active = false; }
				else {
					if (sub == 0) mykeys[depth] = key;
					// the neighbours ahead and behind: lane (b, i) of the group probes hash i of neighbour b on either side
					const int osense = (sense == SENSE) ? ANTISENSE : SENSE;
					uint64_t fb, rb, ndf, ndr, ofb, orb, odf, odr;
					nbr_base(tabs, v, p.k, sense, fb, rb);
					neighbour_mask_delta(p, v, sense, ndf, ndr); // spaced seed: the neighbours' masked-out terms
					nbr_base(tabs, v, p.k, osense, ofb, orb);
					neighbour_mask_delta(p, v, osense, odf, odr);
					unsigned cm = 0, om = 0;
					if (COOP) {
						const unsigned b = sub >> 2, i = sub & 3u, ii = i < p.nh ? i : 0u;
						uint64_t fh, rh, gh, hh;
						nbr_hash(tabs, sense, fb, rb, b, fh, rh);
						fh ^= ndf; rh ^= ndr;
						nbr_hash(tabs, osense, ofb, orb, b, gh, hh);
						gh ^= odf; hh ^= odr;
						// (both loads unconditional and side by side: one round trip)
						const unsigned c1 = probe_c(p, cnt, pos_i(p, rh < fh ? rh : fh, ii));
						const unsigned c2 = probe_c(p, cnt, pos_i(p, hh < gh ? hh : gh, ii));
						const unsigned gb = (unsigned)((wave_ballot(c1 < p.kc) >> (16 * grp)) & 0xFFFFull);
						const unsigned ob = (unsigned)((wave_ballot(c2 < p.kc) >> (16 * grp)) & 0xFFFFull);
#pragma unroll
						for (unsigned q = 0; q < 4; q++) {
							if (((gb >> (4 * q)) & 0xFu) == 0) cm |= 1u << q;
							if (((ob >> (4 * q)) & 0xFu) == 0) om |= 1u << q;
						}
					} else {
						for (unsigned q = 0; q < 4; q++) {
							uint64_t fh, rh;
							nbr_hash(tabs, sense, fb, rb, q, fh, rh);
							fh ^= ndf; rh ^= ndr;
							if (solid_contains(p, cnt, rh < fh ? rh : fh)) cm |= 1u << q;
							if (plain) {
								nbr_hash(tabs, osense, ofb, orb, q, fh, rh);
								fh ^= odf; rh ^= odr;
								if (solid_contains(p, cnt, rh < fh ? rh : fh)) om |= 1u << q;
							}
						}
					}
					{
						const bool one_behind = om != 0 && !(om & (om - 1)); // (the vertex the walk came from is there: one means it and no other)
						if (depth < FP_TRIM) behind_lo |= (one_behind ? 1u : 0u) << depth; else behind_ok = behind_ok && one_behind;
					}
					if (cm & (cm - 1)) plain = false;
					if (cm == 0 && plain && behind_ok) {
						// A plain dead-end tip t_0 .. t_depth: trueBranch (ExtendPath.h:174-244) enters every vertex once (one neighbour
						// ahead each, none after the last) and on the way back may turn round at t_i only if i >= fpTrim or
						// lookAhead(t_i, dir, fpTrim) -- on a plain chain: i + fpTrim <= depth -- where it finds no neighbour behind
						// but the vertex it came from: every call answers false, the deepest one was `depth` deep.
						bool ok = true;
						for (unsigned i = 0; i < FP_TRIM && i + FP_TRIM <= depth; i++) ok = ok && ((behind_lo >> i) & 1u);
						if (ok) is_false = true;
					}
					steps++;
					if (cm == 0) {
						// this walk ends here: back to the last fork with a neighbour left, if there is one
						if (!is_false && nforks > 0 && steps < budget) {
							wave_sync(); // (the fork was written by the group's first lane)
							const Fork f = forks[nforks - 1];
							unsigned m = f.mask;
							const unsigned c = (m & 1u) ? 0u : (m & 2u) ? 1u : (m & 4u) ? 2u : 3u;
							m &= m - 1;
							if (m) { if (sub == 0) forks[nforks - 1].mask = m; } else nforks--;
							uint64_t fb2, rb2, fh2, rh2;
							nbr_base(tabs, f.v, p.k, sense, fb2, rb2);
							nbr_hash(tabs, sense, fb2, rb2, c, fh2, rh2);
							v = make_neighbour(p, f.v, sense, c, fh2, rh2);
							depth = f.depth + 1u;
						} else if (!is_false && from_depth > 0 && steps < budget) {
							// (the guide handed the walk over some way down a read that led nowhere: once more from the branch's first vertex)
							from_depth = 0;
							uint64_t fb0, rb0, fh0, rh0;
							nbr_base(tabs, u, p.k, sense, fb0, rb0);
							nbr_hash(tabs, sense, fb0, rb0, my_b, fh0, rh0);
							v = make_neighbour(p, u, sense, my_b, fh0, rh0);
							depth = 0; nforks = 0;
						} else active = false;
					} else {
						const unsigned c = (cm & 1u) ? 0u : (cm & 2u) ? 1u : (cm & 4u) ? 2u : 3u;
						const unsigned rest = cm & (cm - 1);
						if (rest && nforks < fork_cap) {
							if (sub == 0) { forks[nforks].v = v; forks[nforks].depth = depth; forks[nforks].mask = rest; }
							nforks++;
						}
						uint64_t fh, rh;
						nbr_hash(tabs, sense, fb, rb, c, fh, rh);
						v = make_neighbour(p, v, sense, c, fh, rh);
						depth++;
					}
				}
			}
		}
		if (is_false && sub == 0) sc.chain_d[my_b] = (uint16_t)depth;
		if (COOP) {
			const uint64_t tb = wave_ballot(is_true && sub == 0), fbm = wave_ballot(is_false && sub == 0);
#pragma unroll
			for (unsigned g = 0; g < 4; g++) {
				const unsigned t = (unsigned)((tb >> (16 * g)) & 1ull), f = (unsigned)((fbm >> (16 * g)) & 1ull);
				if (!t && !f) continue;
				// which branch was group g's?
				unsigned seen2 = 0;
#pragma unroll
				for (unsigned b = 0; b < 4; b++)
					if ((mask >> b) & 1u) { if (seen2 == first + g) true_mask |= (t << b) | (f << (4 + b)); seen2++; }
			}
		} else if (is_true) true_mask |= 1u << my_b;
		else if (is_false) true_mask |= 1u << (4 + my_b);
	}
	wave_sync(); // (chain_d is read by the caller)
	return true_mask;
}

// (The neighbour hashes are recomputed here rather than passed in: an array handed to this
// out-of-line function would have to live in memory at every call site, i.e. in the per-lane
// scratch of the unbranched walking loop.)
// PRECONDITION: u is in the solid filter.  The plain-tip shortcut of the branch tests (CB_FALSE in chain_bulk, is_false in
// chain_true_branches) reads "exactly one solid neighbour behind the branch's first vertex" as "that neighbour is u, where the
// search came from" -- true only when u itself is solid.  Every caller passes a vertex of a path being extended: a k-mer of an
// entirely solid read or a vertex successor() returned (walk_extend, walk_linear, walk_bulk); extendPath in the reference has
// the same property (ExtendPath.h:405-459 extends paths of graph vertices, and a vertex of RollingBloomDBG is a solid k-mer).
template <int NW>
ABG_HDX int successor_m(const Params& p, const uint8_t* __restrict__ cnt, const Vtx<NW>& u, int dir,
    unsigned trim, unsigned mask, Vtx<NW>& vout, SearchScratch<NW>& sc)
{
	int sense = (dir == FORWARD) ? SENSE : ANTISENSE;
	uint64_t nfh[4], nrh[4];
	neighbour_hashes(p, u, sense, nfh, nrh);
	vout = u;
	// (the memo holds answers for the walkers' trim only.  Under a spaced seed the key is still the
	// vertex's two UNMASKED rolling hashes: they stand for the whole oriented k-mer, positions under
	// a '0' included, which is what the search's later steps depend on.)
	const bool use_memo = sc.memo.k0 != nullptr && trim == p.trim && (mask & (mask - 1));
	if (use_memo) {
		const int hit = memo_find(sc.memo, u.fh, u.rh, dir);
		if (hit >= 0) {
			sc.n_memo_hits++;
			const int code = hit >> 4;
			const unsigned b = (unsigned)hit & 3u;
			if (code == ER_LENGTH_LIMIT || code == ER_AMBI_OUT) vout = make_neighbour(p, u, sense, b, nfh[b], nrh[b]);
			return code;
		}
	}
	auto answer = [&](int code) -> int {
		if (use_memo && !sc.overflow) { // (a search that ran out of stack answers anything: its walker is restarted)
			// the successor's base: the last (SENSE) or first (ANTISENSE) base of vout
			const unsigned b = (code == ER_LENGTH_LIMIT || code == ER_AMBI_OUT) ? kmer_get(vout.s, sense == SENSE ? p.k - 1 : 0u) : 0u;
			memo_add(sc.memo, u.fh, u.rh, dir, (unsigned)code, b, sc.coop);
			sc.n_memo_adds++;
		}
		return code;
	};
	// Exact by monotonicity of trueBranch in its threshold: every condition that makes
	// trueBranch(e, i) return true (vertex on the stack, depth >= i, a true child) also holds
	// for any smaller threshold, while exploration order, direction changes and the visited set
	// do not depend on the threshold.  So each edge is searched ONCE, at `trim`:
	//  - an edge that is true at `trim` is true at every level of the loop of successor();
	//  - an edge that is false at `trim` was explored exhaustively without meeting the stack, and
	//    its search at a threshold i stops with true exactly when some call is i deep: it is
	//    true at level i iff i <= D, the deepest call of that exhaustive search.
	// Two edges true at `trim` => the loop runs to i == trim and answers AMBI_OUT; otherwise the
	// levels i = 0, 1, 2, 4, ..., trim are replayed on the recorded depths.
	unsigned depth_of[4] = { 0, 0, 0, 0 }; // per edge: trim if true at trim, else D
	if (trim > 0 && (mask & (mask - 1))) {
		// branches that are plain chains are settled together (see chain_true_branches)
#if defined(__HIP_DEVICE_COMPILE__)
		const uint64_t tc0 = sc.dbg_on ? wall_clock64() : 0;
#endif
		const unsigned chain_res = (trim > 1) ? (sc.coop ? chain_true_branches<NW, true>(p, cnt, u, dir, trim, mask, sc)
		                                                 : chain_true_branches<NW, false>(p, cnt, u, dir, trim, mask, sc))
		                                      : 0u;
		const unsigned chain_true = chain_res & 0xFu, chain_false = (chain_res >> 4) & 0xFu;
#if defined(__HIP_DEVICE_COMPILE__)
		if (sc.dbg_on) sc.dbg_chain += wall_clock64() - tc0;
#endif
		
		unsigned tb = 0;
		for (unsigned b = 0; b < 4; b++) {
			if (!((mask >> b) & 1u)) continue;
			if ((chain_false >> b) & 1u) { depth_of[b] = sc.chain_d[b]; continue; } // (a plain dead-end tip)
			Vtx<NW> w = make_neighbour(p, u, sense, b, nfh[b], nrh[b]);
			unsigned d = 0;
			if (((chain_true >> b) & 1u) || true_branch(p, cnt, u, w, dir, trim, sc, &d)) {
				vout = w;
				depth_of[b] = trim;
				if (++tb >= 2) return answer(ER_AMBI_OUT);
			} else {
				depth_of[b] = d;
			}
		}
	}
	for (unsigned i = 0;; i = (i == 0) ? 1u : (trim < 2 * i ? trim : 2 * i)) {
		unsigned tb = 0;
		for (unsigned b = 0; b < 4; b++) {
			if (!((mask >> b) & 1u)) continue;
			// trueBranch(e, dir, g, i, fpTrim) with a fresh visited set; at i == 0 every existing
			// edge is a true branch (depth 0 >= trim 0)
			if (i == 0 || depth_of[b] >= i) {
				vout = make_neighbour(p, u, sense, b, nfh[b], nrh[b]);
				if (++tb >= 2) break;
			}
		}
		if (tb == 0) return answer(ER_DEAD_END);
		if (tb == 1) return answer(ER_LENGTH_LIMIT);
		if (i == trim) return answer(ER_AMBI_OUT);
	}
}
template <int NW>
ABG_HDN int successor(const Params& p, const uint8_t* __restrict__ cnt, const Vtx<NW>& u, int dir,
    unsigned trim, Vtx<NW>& vout, SearchScratch<NW>& sc)
{
	uint64_t nfh[4], nrh[4];
	unsigned mask = neighbour_mask(p, cnt, u, (dir == FORWARD) ? SENSE : ANTISENSE, nfh, nrh, sc.coop);
	return successor_m(p, cnt, u, dir, trim, mask, vout, sc);
}

} // namespace abg
