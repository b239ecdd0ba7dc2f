// fasta_reader.h -- host-side sequence input of the drop-in abyss-bloom-dbg binary.
// Restates the FASTA / FASTQ behaviour of the reference's FastaReader::read
// (DataLayer/FastaReader.cpp:130-421) as used by BloomDBG (flag FOLD_CASE): '#' comment
// lines, Casava chastity filter, multi-line FASTA, masked-end trimming, case folding,
// 3'/5' quality trimming (-q) and internal quality masking (-Q).  SAM / qseq / export
// inputs are not supported by this binary (it stops with an error, as the reference does
// for malformed input).  Compressed files are piped through the matching decompressor the
// way Common/Uncompress.cpp does.
#pragma once
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace abghost {

struct ReaderOptions {         // DataLayer/FastaReader.cpp:15-38 (namespace opt)
	int chastityFilter = 1;
	int trimMasked = 1;
	int qualityThreshold = 0;  // -q
	int qualityOffset = 0;     // --standard-quality 33 / --illumina-quality 64
	int internalQThreshold = 0;// -Q
};

class FastaReader {
  public:
	FastaReader(const std::string& path, const ReaderOptions& o) : m_path(path), m_opt(o)
	{
		static const struct { const char* ext; const char* cmd; } zs[] = {
			{ ".gz", "gunzip -c" }, { ".bz2", "bunzip2 -c" }, { ".xz", "xzdec -c" }, { ".zst", "zstd -dc" },
		};
		for (auto& z : zs) {
			size_t n = strlen(z.ext);
			if (path.size() > n && path.compare(path.size() - n, n, z.ext) == 0) {
				std::string cmd = std::string(z.cmd) + " '" + path + "'";
				m_f = popen(cmd.c_str(), "r");
				m_pipe = true;
			}
		}
		if (!m_f) m_f = (path == "-") ? stdin : fopen(path.c_str(), "r");
		if (!m_f) { // assert_good, Common/IOUtil.h:14-22
			fprintf(stderr, "error: `%s': %s\n", path.c_str(), strerror(errno));
			exit(EXIT_FAILURE);
		}
		setvbuf(m_f, nullptr, _IOFBF, 4u << 20); // one reader per stream: big buffer, unlocked reads
		flockfile(m_f);
		int c = peek();
		if (c == EOF) fprintf(stderr, "%s:0: warning: file is empty\n", m_path.c_str());
	}
	~FastaReader()
	{
		if (m_f) funlockfile(m_f);
		if (m_f && m_f != stdin) { if (m_pipe) pclose(m_f); else fclose(m_f); }
		free(m_line_buf);
	}
	// next record; false at end of file
	bool read(std::string& id, std::string& comment, std::string& s)
	{
		std::string q, header, line;
		for (;;) {
			id.clear(); comment.clear(); q.clear(); s.clear();
			while (peek() == '#') getline(line);
			int type = peek();
			if (type == EOF) return false;
			if (type != '>' && type != '@') die("Expected either `>' or `@' (SAM, qseq and export input are not supported by this binary)");
			getline(header);
			// ignore SAM headers
			if (header.size() > 3 && header[0] == '@' && isalpha((unsigned char)header[1]) &&
			    isalpha((unsigned char)header[2]) && header[3] == '\t')
				continue;
			size_t p = 1, e = p;
			while (e < header.size() && !isspace((unsigned char)header[e])) e++;
			id = header.substr(p, e - p);
			while (e < header.size() && isspace((unsigned char)header[e])) e++;
			comment = header.substr(e);
			bool skip = false;
			if (comment.size() > 3 && comment[1] == ':' && comment[3] == ':') { // Casava
				if (m_opt.chastityFilter && comment[2] == 'Y') {
					if (type == '@') { getline(line); getline(line); getline(line); }
					else { while (peek() != '>' && peek() != '#' && getline(line)) {} }
					skip = true;
				} else if (id.size() > 2 && id[id.size() - 2] != '/') {
					id += '/';
					id += comment[0];
				}
			}
			if (skip) continue;
			getline(s);
			if (type == '>') {
				while (peek() != '>' && peek() != '#' && getline(line)) s += line;
			} else {
				int c = getc_unlocked(m_f);
				if (c != '+') die("expected `+'");
				getline(line);
				getline(q);
			}
			if (s.empty()) die(("sequence with ID `" + id + "' is empty").c_str());
			if (!q.empty() && q.size() != s.size()) die("sequence and quality must be the same length");
			if (m_opt.trimMasked) {
				size_t front = 0;
				while (front < s.size() && islower((unsigned char)s[front])) front++;
				size_t back = s.size();
				while (back > 0 && islower((unsigned char)s[back - 1])) back--;
				if (back < front) back = front;
				s.erase(back); s.erase(0, front);
				if (!q.empty()) { q.erase(back); q.erase(0, front); }
			}
			for (auto& ch : s) ch = (char)toupper((unsigned char)ch); // FOLD_CASE
			unsigned qoff = m_opt.qualityOffset > 0 ? (unsigned)m_opt.qualityOffset : 33u;
			if (m_opt.qualityThreshold > 0 && !q.empty()) {
				// keep [first base with q >= threshold, last such base]
				int good = (int)qoff + m_opt.qualityThreshold;
				size_t front = std::string::npos, back = 0;
				for (size_t i = 0; i < q.size(); i++)
					if ((unsigned char)q[i] >= good && (unsigned char)q[i] <= '~') { if (front == std::string::npos) front = i; back = i + 1; }
				if (front == std::string::npos || front >= back) { s.erase(1); q.erase(1); }
				else if (front > 0 || back < q.size()) { s.erase(back); s.erase(0, front); q.erase(back); q.erase(0, front); }
			}
			if (m_opt.internalQThreshold > 0 && !q.empty()) {
				int good = (int)qoff + m_opt.internalQThreshold;
				for (size_t i = 0; i < q.size(); i++)
					if (!((unsigned char)q[i] >= good && (unsigned char)q[i] <= '~')) s[i] = 'N';
			}
			return true;
		}
	}

  private:
	std::string m_path;
	ReaderOptions m_opt;
	FILE* m_f = nullptr;
	bool m_pipe = false;
	unsigned m_line = 0;
	char* m_line_buf = nullptr;
	size_t m_line_cap = 0;
	int peek() { int c = getc_unlocked(m_f); if (c != EOF) ungetc(c, m_f); return c; }
	bool getline(std::string& out)
	{
		ssize_t n = ::getline(&m_line_buf, &m_line_cap, m_f);
		if (n < 0) { out.clear(); return false; }
		m_line++;
		while (n > 0 && (m_line_buf[n - 1] == '\n' || m_line_buf[n - 1] == '\r')) n--;
		out.assign(m_line_buf, (size_t)n);
		return true;
	}
	[[noreturn]] void die(const char* msg)
	{
		fprintf(stderr, "%s:%u: error: %s\n", m_path.c_str(), m_line, msg);
		exit(EXIT_FAILURE);
	}
};

} // namespace abghost
