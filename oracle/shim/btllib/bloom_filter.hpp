// oracle/shim/btllib/bloom_filter.hpp -- TEST INFRASTRUCTURE ONLY.  See nthash.hpp in this directory: a restatement of
// the btllib classes RResolver/BloomFilters.h:4-26 names, for the oracle build of the unmodified RResolver sources.
//   BloomFilter       an array of `bytes` (rounded up to 8) bytes; bit h % bits of byte (h % bits) / 8 is set for
//                     every hash of an element (btllib's published layout: one bit per hash, BIT_MASKS[i] = 1 << i)
//   KmerBloomFilter   ... over the ntHash values of every k-mer of a sequence; contains(seq) counts the k-mers found
//   SeedBloomFilter   spaced-seed variant, only reached with abyss-rresolver-short -e (off by default, never passed by
//                     bin/abyss-pe:581-585): declared so that the sources compile, aborts when used.
#pragma once
#include "nthash.hpp"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

namespace btllib {

class BloomFilter {
public:
	BloomFilter(size_t bytes, unsigned hash_num)
	    : bytes_((bytes + 7) / 8 * 8), bits_(bytes_ * 8), hash_num_(hash_num), array_(new std::atomic<uint8_t>[bytes_])
	{
		for (size_t i = 0; i < bytes_; i++) array_[i].store(0, std::memory_order_relaxed);
	}
	void insert(const uint64_t* hashes)
	{
		for (unsigned i = 0; i < hash_num_; i++) {
			const uint64_t n = hashes[i] % bits_;
			array_[n / 8].fetch_or((uint8_t)(1u << (n % 8)), std::memory_order_relaxed);
		}
	}
	bool contains(const uint64_t* hashes) const
	{
		for (unsigned i = 0; i < hash_num_; i++) {
			const uint64_t n = hashes[i] % bits_;
			if (!(array_[n / 8].load(std::memory_order_relaxed) & (1u << (n % 8)))) return false;
		}
		return true;
	}
	size_t get_bytes() const { return bytes_; }
	unsigned get_hash_num() const { return hash_num_; }
	uint64_t get_pop_cnt() const
	{
		uint64_t n = 0;
		for (size_t i = 0; i < bytes_; i++) n += (uint64_t)__builtin_popcount(array_[i].load(std::memory_order_relaxed));
		return n;
	}
	double get_occupancy() const { return double(get_pop_cnt()) / double(bits_); }
	double get_fpr() const { return std::pow(get_occupancy(), double(hash_num_)); }
	const std::atomic<uint8_t>* data() const { return array_.get(); }

private:
	size_t bytes_, bits_;
	unsigned hash_num_;
	std::unique_ptr<std::atomic<uint8_t>[]> array_;
};

class KmerBloomFilter {
public:
	KmerBloomFilter(size_t bytes, unsigned hash_num, unsigned k) : k_(k), bf_(bytes, hash_num) {}
	void insert(const char* seq, size_t len)
	{
		NtHash h(seq, len, bf_.get_hash_num(), k_);
		while (h.roll()) bf_.insert(h.hashes());
	}
	void insert(const std::string& seq) { insert(seq.data(), seq.size()); }
	void insert(const uint64_t* hashes) { bf_.insert(hashes); }
	unsigned contains(const char* seq, size_t len) const
	{
		unsigned n = 0;
		NtHash h(seq, len, bf_.get_hash_num(), k_);
		while (h.roll()) n += bf_.contains(h.hashes()) ? 1u : 0u;
		return n;
	}
	unsigned contains(const std::string& seq) const { return contains(seq.data(), seq.size()); }
	bool contains(const uint64_t* hashes) const { return bf_.contains(hashes); }
	size_t get_bytes() const { return bf_.get_bytes(); }
	uint64_t get_pop_cnt() const { return bf_.get_pop_cnt(); }
	double get_occupancy() const { return bf_.get_occupancy(); }
	unsigned get_hash_num() const { return bf_.get_hash_num(); }
	double get_fpr() const { return bf_.get_fpr(); }
	unsigned get_k() const { return k_; }
	const BloomFilter& get_bloom_filter() const { return bf_; }

private:
	unsigned k_;
	BloomFilter bf_;
};

typedef std::vector<unsigned> SpacedSeed;
class SeedBloomFilter {
public:
	SeedBloomFilter(size_t, unsigned k, const std::vector<std::string>& seeds, unsigned) : k_(k), seeds_(seeds)
	{
		for (const auto& s : seeds) {
			SpacedSeed p;
			for (unsigned i = 0; i < s.size(); i++) if (s[i] == '0') p.push_back(i);
			parsed_.push_back(p);
		}
	}
	[[noreturn]] static void unsupported()
	{
		fprintf(stderr, "oracle/shim/btllib: SeedBloomFilter (abyss-rresolver-short -e) is not restated in the oracle build\n");
		abort();
	}
	void insert(const std::string&) { unsupported(); }
	std::vector<std::vector<unsigned>> contains(const std::string&) const { unsupported(); }
	const std::vector<SpacedSeed>& get_parsed_seeds() const { return parsed_; }
	unsigned get_k() const { return k_; }
	double get_occupancy() const { return 0; }
	double get_fpr() const { return 0; }

private:
	unsigned k_;
	std::vector<std::string> seeds_;
	std::vector<SpacedSeed> parsed_;
};

} // namespace btllib
