// oracle/shim/btllib/nthash.hpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A restatement of the part of btllib's NtHash that ABySS's RResolver uses (RResolver/BloomFilters.cpp,
// RResolver/RAlgorithmsShort.cpp:316-363), so that the UNMODIFIED RResolver sources under /root/reference compile into
// oracle/_ref/abyss-rresolver-short (oracle/Makefile).  btllib itself is NOT under /root/reference (configure.ac:268-276
// requires an installed copy, "conda install btllib", no version pinned) and cannot be fetched here, so this header
// restates its published algorithm (ntHash, Mohamadi et al. 2016; ntHash2, Kazemi et al. 2022 -- the canonical hash of
// btllib >= 1.4 is forward + reverse, the extra hashes are the multiply-shift family that vendor/nthash/nthash.hpp:
// 56-57,306-322 of the reference also uses):
//   * per-base seeds, the split rotation of the 33 low / 31 high bits (srol), NTF64 / NTR64 rolling;
//   * k-mers holding a character other than ACGT (either case) are skipped;
//   * hashes[0] = fwd + rev, hashes[i] = h0 * (i ^ k * MULTISEED), then x ^= x >> MULTISHIFT.
// PARITY STATUS: unpinned against a real btllib build (none available) -- the RResolver goldens made with this shim pin
// the product to "the reference's RResolver over this restatement", which is what a maintainer can re-check by building
// the reference against btllib proper and diffing tests/golden/rresolver/.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace btllib {

namespace nthash_detail {
static const uint64_t SEED_A = 0x3c8bfbb395c60474ULL, SEED_C = 0x3193c18562a02b4cULL, SEED_G = 0x20323ed082572324ULL,
                      SEED_T = 0x295549f54be24456ULL;
static const uint64_t MULTISEED = 0x90b45d39fb6da1faULL;
static const unsigned MULTISHIFT = 27;
inline int code(unsigned char c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return -1;
	}
}
inline uint64_t seed(int c) { return c == 0 ? SEED_A : c == 1 ? SEED_C : c == 2 ? SEED_G : SEED_T; }
inline uint64_t srol(uint64_t x)
{
	const uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
	return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
}
inline uint64_t sror(uint64_t x)
{
	const uint64_t m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32);
	return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m;
}
inline uint64_t srol(uint64_t x, unsigned n) { for (unsigned i = 0; i < n; i++) x = srol(x); return x; }
} // namespace nthash_detail

class NtHash {
public:
	NtHash(const char* seq, size_t seq_len, unsigned hash_num, unsigned k, size_t pos = 0)
	    : seq_(seq, seq_len), hash_num_(hash_num), k_(k), pos_(pos), hashes_(new uint64_t[hash_num ? hash_num : 1]) {}
	NtHash(const std::string& seq, unsigned hash_num, unsigned k, size_t pos = 0)
	    : NtHash(seq.data(), seq.size(), hash_num, k, pos) {}
	// the next k-mer that holds only ACGT; false when the sequence is exhausted
	bool roll()
	{
		using namespace nthash_detail;
		if (!initialized_) return init();
		if (pos_ + k_ >= seq_.size()) return false;
		const int in = code((unsigned char)seq_[pos_ + k_]);
		if (in < 0) { pos_ += k_ + 1; initialized_ = false; return init(); } // (the next k-mer free of the bad character)
		const int out = code((unsigned char)seq_[pos_]);
		fwd_ = srol(fwd_) ^ seed(in) ^ srol(seed(out), k_);
		rev_ = sror(rev_ ^ srol(seed(3 - in), k_) ^ seed(3 - out));
		pos_++;
		extend();
		return true;
	}
	// the hashes of the current k-mer with the bases at `positions` replaced (RAlgorithmsShort.cpp:331-344, -e only)
	void sub(const std::vector<unsigned>& positions, const std::vector<unsigned char>& new_bases)
	{
		using namespace nthash_detail;
		uint64_t f = fwd_, r = rev_;
		for (size_t i = 0; i < positions.size(); i++) {
			const unsigned p = positions[i];
			const int o = code((unsigned char)seq_[pos_ + p]), n = code(new_bases[i]);
			if (o < 0 || n < 0) continue;
			f ^= srol(seed(o), k_ - 1 - p) ^ srol(seed(n), k_ - 1 - p);
			r ^= srol(seed(3 - o), p) ^ srol(seed(3 - n), p);
		}
		const uint64_t kf = fwd_, kr = rev_;
		fwd_ = f; rev_ = r; extend(); fwd_ = kf; rev_ = kr;
	}
	const uint64_t* hashes() const { return hashes_.get(); }
	size_t get_pos() const { return pos_; }
	unsigned get_hash_num() const { return hash_num_; }
	unsigned get_k() const { return k_; }
	uint64_t get_forward_hash() const { return fwd_; }
	uint64_t get_reverse_hash() const { return rev_; }

private:
	bool init()
	{
		using namespace nthash_detail;
		if (k_ == 0) return false;
		while (pos_ + k_ <= seq_.size()) {
			size_t bad = k_;
			for (size_t i = k_; i-- > 0;) if (code((unsigned char)seq_[pos_ + i]) < 0) { bad = i; break; }
			if (bad < k_) { pos_ += bad + 1; continue; }
			fwd_ = 0; rev_ = 0;
			for (unsigned i = 0; i < k_; i++) {
				fwd_ = srol(fwd_) ^ seed(code((unsigned char)seq_[pos_ + i]));
				rev_ = srol(rev_) ^ seed(3 - code((unsigned char)seq_[pos_ + k_ - 1 - i]));
			}
			initialized_ = true;
			extend();
			return true;
		}
		return false;
	}
	void extend()
	{
		using namespace nthash_detail;
		const uint64_t h0 = fwd_ + rev_;
		hashes_[0] = h0;
		for (unsigned i = 1; i < hash_num_; i++) {
			uint64_t t = h0 * (i ^ k_ * MULTISEED);
			t ^= t >> MULTISHIFT;
			hashes_[i] = t;
		}
	}
	std::string seq_;
	unsigned hash_num_, k_;
	size_t pos_;
	bool initialized_ = false;
	uint64_t fwd_ = 0, rev_ = 0;
	std::unique_ptr<uint64_t[]> hashes_;
};

} // namespace btllib
