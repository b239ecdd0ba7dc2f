// oracle/shim/btllib/seq_reader.hpp -- TEST INFRASTRUCTURE ONLY.  See nthash.hpp in this directory.  The part of
// btllib::SeqReader that RResolver uses (BloomFilters.cpp:167-169, RAlgorithmsShort.cpp:120-123): records of a FASTA or
// FASTQ file (plain, or .gz / .bz2 / .xz through the usual decompressor), read() safe to call from several threads, the
// sequence folded to upper case (btllib's default flag FOLD_CASE).  Record order is file order.
#pragma once
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

namespace btllib {

class SeqReader {
public:
	struct Flag {
		static const unsigned FOLD_CASE = 0, NO_FOLD_CASE = 1, NO_TRIM_MASKED = 0, TRIM_MASKED = 2, SHORT_MODE = 4, LONG_MODE = 8;
	};
	struct Record {
		size_t num = 0;
		std::string id, comment, seq, qual;
		bool valid = false;
		operator bool() const { return valid; }
	};
	SeqReader(const std::string& path, unsigned flags, unsigned /*threads*/ = 3) : flags_(flags)
	{
		auto ends = [&](const char* suf) { const size_t n = std::string(suf).size(); return path.size() >= n && path.compare(path.size() - n, n, suf) == 0; };
		std::string cmd;
		if (ends(".gz")) cmd = "gzip -dc '" + path + "'";
		else if (ends(".bz2")) cmd = "bzip2 -dc '" + path + "'";
		else if (ends(".xz")) cmd = "xz -dc '" + path + "'";
		if (!cmd.empty()) { f_ = popen(cmd.c_str(), "r"); piped_ = true; }
		else f_ = fopen(path.c_str(), "r");
		if (!f_) { fprintf(stderr, "oracle/shim/btllib: cannot open %s\n", path.c_str()); exit(EXIT_FAILURE); }
	}
	~SeqReader() { close(); }
	void close() { if (f_) { if (piped_) pclose(f_); else fclose(f_); f_ = nullptr; } }
	Record read()
	{
		std::lock_guard<std::mutex> g(m_);
		Record r;
		if (!f_) return r;
		std::string h;
		while (getline(h) && h.empty()) {}
		if (h.empty()) return r;
		const bool fastq = h[0] == '@';
		if (h[0] != '>' && !fastq) { fprintf(stderr, "oracle/shim/btllib: unrecognised record header\n"); exit(EXIT_FAILURE); }
		const size_t sp = h.find_first_of(" \t");
		r.id = h.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
		if (sp != std::string::npos) r.comment = h.substr(sp + 1);
		if (fastq) {
			std::string plus;
			getline(r.seq); getline(plus); getline(r.qual);
		} else {
			// (FASTA: the sequence may run over several lines)
			int c;
			std::string line;
			while ((c = fgetc(f_)) != EOF) {
				ungetc(c, f_);
				if (c == '>') break;
				if (!getline(line)) break;
				r.seq += line;
			}
		}
		if (!(flags_ & Flag::NO_FOLD_CASE)) for (auto& c : r.seq) c = (char)toupper((unsigned char)c);
		r.num = n_++;
		r.valid = true;
		return r;
	}

private:
	bool getline(std::string& s)
	{
		s.clear();
		int c;
		bool any = false;
		while ((c = fgetc(f_)) != EOF) {
			any = true;
			if (c == '\n') break;
			if (c != '\r') s.push_back((char)c);
		}
		return any;
	}
	FILE* f_ = nullptr;
	bool piped_ = false;
	unsigned flags_;
	size_t n_ = 0;
	std::mutex m_;
};

} // namespace btllib
