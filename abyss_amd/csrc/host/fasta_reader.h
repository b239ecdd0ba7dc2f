// fasta_reader.h -- host-side sequence input of the drop-in abyss-bloom-dbg binary.
// Restates the FASTA / FASTQ behaviour of the reference's FastaReader::read
// (DataLayer/FastaReader.cpp:130-421) as used by BloomDBG (flag FOLD_CASE): '#' comment
// lines, Casava chastity filter, multi-line FASTA, masked-end trimming, case folding,
// 3'/5' quality trimming (-q) and internal quality masking (-Q); SAM records (secondary and
// QC-failed alignments skipped, reverse strand restored, /1 /2 suffixes) and qseq / export
// lines (:215-352).  Colour-space reads are not supported.  Compressed files are piped through
// the matching decompressor the way Common/Uncompress.cpp does.
#pragma once
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <functional>
#include <future>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <csignal>
#include <cstdint>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

namespace abghost {

struct ReaderOptions {         // DataLayer/FastaReader.cpp:15-38 (namespace opt)
	int chastityFilter = 1;
	int trimMasked = 1;
	int qualityThreshold = 0;  // -q
	int qualityOffset = 0;     // --standard-quality 33 / --illumina-quality 64
	int internalQThreshold = 0;// -Q
	int foldCase = 1;          // 0: FastaReader::NO_FOLD_CASE (RResolver/Contigs.cpp:128)
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
				// the decompressor as a child with an argv of its own (no shell: a file name may hold
				// quotes or spaces), its stdout our stream; its exit status is checked when the stream closes
				char prog[32], flag[8];
				if (sscanf(z.cmd, "%31s %7s", prog, flag) != 2) continue;
				int fd[2];
				if (pipe(fd)) { perror("pipe"); exit(EXIT_FAILURE); }
				m_child = fork();
				if (m_child < 0) { perror("fork"); exit(EXIT_FAILURE); }
				if (m_child == 0) {
					dup2(fd[1], 1); close(fd[0]); close(fd[1]);
					execlp(prog, prog, flag, path.c_str(), (char*)NULL);
					fprintf(stderr, "error: cannot run `%s': %s\n", prog, strerror(errno));
					_exit(127);
				}
				close(fd[1]);
				m_f = fdopen(fd[0], "r");
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
	// over a stream the caller opened (e.g. fmemopen on a block of a file: SequenceReader below);
	// `first_line` = number of lines before it in the file `path`, for error messages
	FastaReader(FILE* f, const std::string& path, unsigned first_line, const ReaderOptions& o)
	    : m_path(path), m_opt(o), m_f(f), m_line(first_line)
	{
		flockfile(m_f);
	}
	~FastaReader()
	{
		if (m_f) funlockfile(m_f);
		if (m_f && m_f != stdin) fclose(m_f);
		if (m_pipe && m_child > 0) {
			// (waitpid on this child only; if somebody else reaped it already its status is unknown here)
			int st = 0;
			if (waitpid(m_child, &st, 0) == m_child && !(WIFEXITED(st) && WEXITSTATUS(st) == 0) && !(WIFSIGNALED(st) && WTERMSIG(st) == SIGPIPE)) {
				fprintf(stderr, "error: decompressing `%s' failed\n", m_path.c_str());
				exit(EXIT_FAILURE);
			}
		}
		free(m_line_buf);
	}
	FastaReader(const FastaReader&) = delete;
	FastaReader& operator=(const FastaReader&) = delete;
	// next record; false at end of file
	bool read(std::string& id, std::string& comment, std::string& s)
	{
		std::string q, header, line;
		for (;;) {
			id.clear(); comment.clear(); q.clear(); s.clear();
			while (peek() == '#') getline(line);
			int type = peek();
			if (type == EOF) return false;
			unsigned qoff_default = 33;
			if (type != '>' && type != '@') {
				// a SAM, qseq or export line (FastaReader.cpp:256-352): no case folding, no masked-end trimming
				int r = read_tabular(id, comment, s, q, qoff_default);
				if (r == 0) continue; // record filtered out
				return finish(s, q, qoff_default);
			}
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
			if (m_opt.foldCase) for (auto& ch : s) ch = (char)toupper((unsigned char)ch); // FOLD_CASE
			return finish(s, q, qoff_default);
		}
	}

  private:
	// quality trimming / masking shared by every format (FastaReader.cpp:355-399)
	bool finish(std::string& s, std::string& q, unsigned qoff_default)
	{
		unsigned qoff = m_opt.qualityOffset > 0 ? (unsigned)m_opt.qualityOffset : qoff_default;
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
	static bool chaste(const std::string& f, bool& ok)
	{
		ok = true;
		if (f == "1" || f == "Y") return true;
		if (f == "0" || f == "N") return false;
		ok = false;
		return false;
	}
	static char complement(char c) // complementBaseChar, Common/Sequence.cpp:21-47
	{
		char rc;
		switch (toupper((unsigned char)c)) {
		case 'A': rc = 'T'; break; case 'C': rc = 'G'; break; case 'G': rc = 'C'; break; case 'T': rc = 'A'; break;
		case 'N': rc = 'N'; break; case '.': rc = '.'; break; case 'M': rc = 'K'; break; case 'R': rc = 'Y'; break;
		case 'W': rc = 'W'; break; case 'S': rc = 'S'; break; case 'Y': rc = 'R'; break; case 'K': rc = 'M'; break;
		case 'V': rc = 'B'; break; case 'H': rc = 'D'; break; case 'D': rc = 'H'; break; case 'B': rc = 'V'; break;
		default: fprintf(stderr, "error: unexpected character: `%c'\n", c); abort();
		}
		return islower((unsigned char)c) ? (char)tolower(rc) : rc;
	}
	// returns 1 with a record, 0 when the line was filtered out
	int read_tabular(std::string& id, std::string& comment, std::string& s, std::string& q, unsigned& qoff_default)
	{
		std::string line;
		getline(line);
		std::vector<std::string> f;
		{
			size_t a = 0;
			while (a <= line.size()) { // std::getline(in, field, '\t'): no empty last field after a trailing tab
				size_t b = line.find('\t', a);
				if (b == std::string::npos) { if (a < line.size()) f.push_back(line.substr(a)); break; }
				f.push_back(line.substr(a, b - a));
				a = b + 1;
			}
		}
		if (f.size() >= 11 && (f[9].size() == f[10].size() || f[10] == "*")) { // SAM
			unsigned flags = (unsigned)strtoul(f[1].c_str(), NULL, 0);
			if (flags & 0x100) return 0;                         // FSECONDARY
			if (m_opt.chastityFilter && (flags & 0x200)) return 0; // FQCFAIL
			id = f[0];
			char which = '0';
			switch (flags & 0xc1) { // FPAIRED|FREAD1|FREAD2
			case 0: case 1: break;
			case 0x41: id += "/1"; which = '1'; break;
			case 0x81: id += "/2"; which = '2'; break;
			default: die(("invalid flags: `" + id + "'").c_str());
			}
			comment = (flags & 0x200) ? "0:Y:0:" : "0:N:0:";
			comment[0] = which;
			s = f[9]; q = f[10];
			if (s == "*") s.clear();
			if (q == "*") q.clear();
			if (flags & 0x10) { // FREVERSE
				std::string rc(s.rbegin(), s.rend());
				for (auto& ch : rc) ch = complement(ch);
				s = rc;
				q.assign(q.rbegin(), q.rend());
			}
			qoff_default = 33;
			if (!q.empty() && q.size() != s.size()) die("sequence and quality must be the same length");
			return 1;
		}
		if (f.size() == 11 || f.size() == 22) { // qseq or export
			bool ok;
			bool ch = chaste(f.back(), ok);
			if (!ok) die("chastity filter should be one of 0, 1, N or Y");
			if (m_opt.chastityFilter && !ch) return 0;
			id = f[0];
			for (int i = 1; i < 6; i++) if (!f[i].empty()) { id += ':'; id += f[i]; }
			if (!f[6].empty() && f[6] != "0") { id += '#'; id += f[6]; }
			id += '/';
			id += (f[7] == "3" ? std::string("2") : f[7]);
			comment = f[7] + (ch ? ":N:0:" : ":Y:0:");
			s = f[8]; q = f[9];
			qoff_default = 64;
			if (q.size() != s.size()) die("sequence and quality must be the same length");
			return 1;
		}
		die("Expected either `>' or `@' or 11 fields");
	}
	std::string m_path;
	ReaderOptions m_opt;
	FILE* m_f = nullptr;
	bool m_pipe = false;
	pid_t m_child = -1; // the decompressor, when the input is compressed
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
		if (m_worker) {
			// One of several parser threads: exit() would run the static destructors and close every stream beside the threads that
			// are still parsing (seen once as a crash instead of status 1, on a busy host).  The first complaint is printed, what
			// the process has written to stdout goes out, and the process ends there and then.
			static std::atomic<bool> dying{ false };
			if (dying.exchange(true)) for (;;) pause();
			fprintf(stderr, "%s:%u: error: %s\n", m_path.c_str(), m_line, msg);
			fflush(stdout);
			_exit(EXIT_FAILURE);
		}
		fprintf(stderr, "%s:%u: error: %s\n", m_path.c_str(), m_line, msg);
		exit(EXIT_FAILURE);
	}
  public:
	// this reader runs on one of several parser threads (SequenceReader, read_fasta_blocks): see die()
	void on_worker_thread() { m_worker = true; }
  private:
	bool m_worker = false;
};

// Compressed inputs decompressed AHEAD: every one by a decompressor child of its own (gunzip -c
// and the like, as Common/Uncompress.cpp runs them), each drained into memory by a thread of its
// own, all at once -- the reference and the sequential reader below take the files one after the
// other at the pace of one gunzip.  Only when the inputs, inflated, fit this process's share of a
// quarter of the host's memory (estimated at 8 x the compressed size; `share` = the ranks of a
// --gpus run, every one of which reads every input); the estimate is backed by a hard cap on the
// bytes actually inflated -- a file that inflates past the budget is dropped from the prefetch
// and read as a stream like anything asked for a second time.
class Prefetch {
  public:
	static Prefetch& get() { static Prefetch p; return p; }
	static bool compressed(const std::string& path, const char** prog, const char** flag)
	{
		static const struct { const char* ext; const char* prog; const char* flag; } zs[] = {
			{ ".gz", "gunzip", "-c" }, { ".bz2", "bunzip2", "-c" }, { ".xz", "xzdec", "-c" }, { ".zst", "zstd", "-dc" },
		};
		for (auto& z : zs) {
			const size_t n = strlen(z.ext);
			if (path.size() > n && path.compare(path.size() - n, n, z.ext) == 0) { *prog = z.prog; *flag = z.flag; return true; }
		}
		return false;
	}
	void start(const std::vector<std::string>& paths, unsigned share = 1)
	{
		uint64_t total = 0;
		std::vector<std::string> todo;
		for (auto& p : paths) {
			const char* prog; const char* flag;
			struct stat st;
			if (!compressed(p, &prog, &flag) || stat(p.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
			if (std::find(todo.begin(), todo.end(), p) != todo.end()) continue;
			total += (uint64_t)st.st_size;
			todo.push_back(p);
		}
		const uint64_t ram = (uint64_t)sysconf(_SC_PHYS_PAGES) * (uint64_t)sysconf(_SC_PAGE_SIZE);
		m_budget = ram / 4 / (share ? share : 1);
		if (const char* e = getenv("ABG_PREFETCH_BUDGET_MB")) m_budget = strtoull(e, nullptr, 10) << 20; // (tests)
		if (todo.empty() || total * 8 > m_budget) return;
		for (auto& p : todo) {
			const char* prog; const char* flag;
			compressed(p, &prog, &flag);
			int fd[2];
			if (pipe(fd)) return;
			const pid_t pid = fork();
			if (pid < 0) { close(fd[0]); close(fd[1]); return; }
			if (pid == 0) {
				dup2(fd[1], 1); close(fd[0]); close(fd[1]);
				execlp(prog, prog, flag, p.c_str(), (char*)NULL);
				_exit(127);
			}
			close(fd[1]);
			Item* it = new Item;
			it->pid = pid;
			it->th = std::thread([this, it, rfd = fd[0]]() {
				std::string& d = it->data;
				size_t have = 0;
				bool over = false;
				for (;;) {
					if (d.size() - have < (8u << 20)) d.resize(d.size() + (64u << 20));
					const ssize_t got = ::read(rfd, &d[have], d.size() - have);
					if (got <= 0) break;
					have += (size_t)got;
					if (m_inflated.fetch_add((uint64_t)got) + (uint64_t)got > m_budget) { over = true; break; } // the estimate was wrong: give this one up
				}
				if (over) {
					kill(it->pid, SIGTERM);
					m_inflated.fetch_sub(have);
					std::string().swap(d);
				} else d.resize(have);
				close(rfd);
				int st = 0;
				it->ok = waitpid(it->pid, &st, 0) == it->pid && WIFEXITED(st) && WEXITSTATUS(st) == 0 && !over;
			});
			m_items[p] = it;
		}
	}
	// the inflated bytes of `path` if they were fetched ahead (once: the next request reads the stream)
	bool take(const std::string& path, std::string& data)
	{
		auto f = m_items.find(path);
		if (f == m_items.end()) return false;
		Item* it = f->second;
		m_items.erase(f);
		it->th.join();
		const bool ok = it->ok;
		if (ok) { m_inflated.fetch_sub(it->data.size()); data.swap(it->data); } // (the reader owns the bytes now)
		delete it;
		return ok; // (a failed decompressor: the ordinary reader runs it again and reports)
	}
	~Prefetch() { for (auto& kv : m_items) { kv.second->th.join(); delete kv.second; } }
  private:
	struct Item { std::thread th; std::string data; pid_t pid = -1; bool ok = false; };
	std::map<std::string, Item*> m_items;
	std::atomic<uint64_t> m_inflated{ 0 }; // bytes held by the drain threads
	uint64_t m_budget = 0;
};

// A plain FASTA file of a megabyte or more (unitigs: tens of megabytes that AdjList and abyss-rresolver-short read before anything
// else happens) parsed by several threads: cut at record starts -- a line beginning with '>' -- into a block per thread, every block
// through the same FastaReader (over fmemopen, with its first line's number for the messages), `hook` run on every record where it
// was parsed (work that needs nothing but the record).  The records come back per block, in file order.  false: not such a file
// (compressed, stdin, small, not beginning with '>', one thread) -- the caller reads it with FastaReader as before.
// (ABG_FASTA_BLOCKS_MIN: the size from which on, tests)
struct FastaRecord { std::string id, comment, seq, aux; };
inline bool read_fasta_blocks(const std::string& path, const ReaderOptions& ro, unsigned threads, std::vector<std::vector<FastaRecord>>& parts,
    const std::function<void(FastaRecord&)>& hook = nullptr)
{
	const unsigned T = std::min(16u, threads);
	const char* prog; const char* flag;
	struct stat st;
	const char* e = getenv("ABG_FASTA_BLOCKS_MIN");
	const long min_bytes = e ? atol(e) : (1L << 20);
	if (T < 2 || path == "-" || min_bytes < 0 || Prefetch::compressed(path, &prog, &flag) || stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < min_bytes || st.st_size == 0) return false;
	// (the file and its records are in memory together for a moment: not for a file that is a sizeable part of the host's memory)
	if ((uint64_t)st.st_size > (uint64_t)sysconf(_SC_PHYS_PAGES) * (uint64_t)sysconf(_SC_PAGE_SIZE) / 16) return false;
	std::string text((size_t)st.st_size, '\0');
	FILE* f = fopen(path.c_str(), "rb");
	const size_t got = f ? fread(&text[0], 1, text.size(), f) : 0;
	if (f) fclose(f);
	if (got != text.size() || text[0] != '>') return false;
	std::vector<size_t> start{ 0 };
	for (unsigned t = 1; t < T; t++) {
		const size_t at = text.find("\n>", text.size() / T * t);
		if (at != std::string::npos && at + 1 > start.back()) start.push_back(at + 1);
	}
	start.push_back(text.size());
	const size_t nb = start.size() - 1;
	parts.assign(nb, std::vector<FastaRecord>());
	std::vector<unsigned> line0(nb + 1, 0);
	std::vector<std::thread> pool;
	for (size_t b = 0; b < nb; b++)
		pool.emplace_back([&, b]() { line0[b + 1] = (unsigned)std::count(text.begin() + (ptrdiff_t)start[b], text.begin() + (ptrdiff_t)start[b + 1], '\n'); });
	for (auto& t : pool) t.join();
	pool.clear();
	for (size_t b = 0; b < nb; b++) line0[b + 1] += line0[b];
	for (size_t b = 0; b < nb; b++)
		pool.emplace_back([&, b]() {
			FILE* m = fmemopen((void*)(text.data() + start[b]), start[b + 1] - start[b], "r");
			if (!m) { fprintf(stderr, "error: fmemopen: %s\n", strerror(errno)); fflush(stdout); _exit(EXIT_FAILURE); } // (one of several threads: as FastaReader::die)
			FastaReader in(m, path, line0[b], ro);
			in.on_worker_thread();
			FastaRecord r;
			while (in.read(r.id, r.comment, r.seq)) {
				if (hook) hook(r);
				parts[b].push_back(std::move(r));
				r = FastaRecord();
			}
		});
	for (auto& t : pool) t.join();
	return true;
}

// FASTQ files parsed by several threads.  Parsing is what the host binary spends its time on once
// the kernels are fast (a single thread reads ~330 MB/s); records are independent, so a window of
// the file is cut into blocks at record boundaries and every block is parsed by the SAME
// FastaReader logic (over fmemopen), which keeps the reference's semantics -- chastity filter,
// trimming, masking, case folding -- by construction.  Records come out in file order.
// A FASTQ record starts at a line beginning with '@' whose second-next line begins with '+': a
// quality line may begin with '@', but then the line after it is a header and the one after that a
// sequence, which cannot begin with '+'.  Anything that is not a plain file beginning with '@'
// (FASTA, SAM/qseq/export, compressed input, stdin) goes through the sequential reader.
class SequenceReader {
  public:
	SequenceReader(const std::string& path, const ReaderOptions& o, unsigned threads) : m_path(path), m_opt(o), m_threads(threads)
	{
		bool plain = threads > 1 && path != "-";
		for (const char* ext : { ".gz", ".bz2", ".xz", ".zst" }) {
			size_t n = strlen(ext);
			if (path.size() > n && path.compare(path.size() - n, n, ext) == 0) plain = false;
		}
		if (!plain && Prefetch::get().take(path, m_buf)) {
			// inflated ahead, in memory: a FASTQ stream is parsed as one window, block-parallel; anything
			// else by the sequential reader over the buffer
			const size_t l1 = m_buf.find('\n');
			const size_t l2 = l1 == std::string::npos ? l1 : m_buf.find('\n', l1 + 1);
			const bool fastq = threads > 1 && !m_buf.empty() && m_buf[0] == '@' && l2 != std::string::npos && l2 + 1 < m_buf.size() && m_buf[l2 + 1] == '+';
			if (fastq) { m_eof = true; m_mem = true; m_window = 1; return; }
			if (m_buf.empty()) fprintf(stderr, "%s:0: warning: file is empty\n", m_path.c_str());
			FILE* f = m_buf.empty() ? fopen("/dev/null", "r") : fmemopen((void*)m_buf.data(), m_buf.size(), "r");
			if (!f) { fprintf(stderr, "error: fmemopen: %s\n", strerror(errno)); exit(EXIT_FAILURE); }
			m_seq = new FastaReader(f, path, 0, o);
			return;
		}
		if (plain) {
			m_f = fopen(path.c_str(), "rb");
			if (m_f) {
				// a regular file whose first record is a 4-line FASTQ record ('@' header, '+' on the third
				// line); a SAM file also begins with '@' (its @HD / @SQ header) but fails the second test
				struct stat st;
				bool fastq = fstat(fileno(m_f), &st) == 0 && S_ISREG(st.st_mode);
				if (fastq) {
					m_buf.resize(1u << 16);
					m_buf.resize(fread(&m_buf[0], 1, m_buf.size(), m_f)); // (stays in the buffer: the first window starts with it)
					const size_t l1 = m_buf.find('\n');
					const size_t l2 = l1 == std::string::npos ? l1 : m_buf.find('\n', l1 + 1);
					fastq = !m_buf.empty() && m_buf[0] == '@' && l2 != std::string::npos && l2 + 1 < m_buf.size() && m_buf[l2 + 1] == '+';
				}
				if (!fastq) { fclose(m_f); m_f = nullptr; }
				else {
					// The file mapped: the parser threads read the page cache where it lies (no copy into a window buffer first --
					// 256 MB of pread by 16 threads was the longest serial stretch of a window, ~4 of the reader's ~4 GB/s).
					// ABG_READER_MMAP=0, or a mapping that fails, falls back to windows filled by pread.
					// What the mapping asks of the file: it keeps the size it had when it was opened -- bytes appended later are not
					// read (the pread path reads to the end of the file as it is then), and a file truncated or rewritten under the
					// run takes the process down with SIGBUS where pread sees a short read.  Reads being written by another process
					// go through ABG_READER_MMAP=0 or a pipe.  Windows already parsed are given back (MADV_DONTNEED, next_window).
					const char* e = getenv("ABG_READER_MMAP");
					if (st.st_size > 0 && !(e && atoi(e) == 0)) {
						void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(m_f), 0);
						if (m != MAP_FAILED) { m_map = (const char*)m; m_map_size = (size_t)st.st_size; madvise(m, m_map_size, MADV_SEQUENTIAL); }
					}
					if (!m_map) { m_raw.grow(m_buf.size()); memcpy(m_raw.p, m_buf.data(), m_buf.size()); m_raw.n = m_buf.size(); m_pos = m_buf.size(); }
				}
				m_buf.clear();
			}
		}
		if (!m_f) m_seq = new FastaReader(path, o); // (also reports a missing file the reference's way)
		else m_window = std::min<size_t>((size_t)threads * (32u << 20), (size_t)256 << 20);
		if (const char* e = getenv("ABG_READER_WINDOW")) m_window = std::max<size_t>(64, strtoull(e, nullptr, 10)); // tests: many small windows
	}
	~SequenceReader()
	{
		if (m_next.valid()) m_next.wait();
		delete m_seq;
		if (m_map) munmap((void*)m_map, m_map_size);
		if (m_f) fclose(m_f);
	}
	SequenceReader(const SequenceReader&) = delete;
	SequenceReader& operator=(const SequenceReader&) = delete;
	bool read(std::string& id, std::string& comment, std::string& s)
	{
		if (m_seq) return m_seq->read(id, comment, s);
		if (m_mem && m_done) return false;
		for (;;) {
			while (m_block < m_blocks.size()) {
				Block& b = m_blocks[m_block];
				if (m_rec < b.seq_end.size()) {
					const size_t i = m_rec++;
					id.assign(b.ids, i ? b.id_end[i - 1] : 0, b.id_end[i] - (i ? b.id_end[i - 1] : 0));
					comment.assign(b.comments, i ? b.com_end[i - 1] : 0, b.com_end[i] - (i ? b.com_end[i - 1] : 0));
					s.assign(b.seqs, i ? b.seq_end[i - 1] : 0, b.seq_end[i] - (i ? b.seq_end[i - 1] : 0));
					return true;
				}
				m_block++; m_rec = 0;
			}
			// the next window was being parsed while the caller worked on this one (GPU calls included)
			if (!m_next.valid()) m_next = std::async(std::launch::async, [this]() { return parse_window(); });
			Window w = m_next.get();
			if (!w.ok) { m_done = true; return false; }
			m_blocks = std::move(w.blocks); m_block = 0; m_rec = 0;
			m_next = std::async(std::launch::async, [this]() { return parse_window(); });
		}
	}

	// Starts reading and parsing the first window now, on the reader's background thread, instead of at
	// the first read() / next_block(): a caller with start-up work of its own does it meanwhile.
	void prime()
	{
		if (m_seq || (m_mem && m_done) || m_next.valid() || !m_blocks.empty()) return;
		m_next = std::async(std::launch::async, [this]() { return parse_window(); });
	}
	// The records a parser thread produced, as they lie: concatenated strings and their end offsets.
	struct Block {
		std::string ids, comments, seqs;
		std::vector<size_t> id_end, com_end, seq_end;
	};
	// A caller that takes records wholesale asks for the parser threads' blocks instead of single
	// records (false: no block mode for this input, or no more blocks; do not mix with read()).
	bool has_blocks() const { return m_seq == nullptr; }
	bool next_block(Block& out)
	{
		if (m_seq || (m_mem && m_done)) return false;
		for (;;) {
			if (m_block < m_blocks.size()) { out = std::move(m_blocks[m_block++]); m_rec = 0; return true; }
			if (!m_next.valid()) m_next = std::async(std::launch::async, [this]() { return parse_window(); });
			Window w = m_next.get();
			if (!w.ok) { m_done = true; return false; }
			m_blocks = std::move(w.blocks); m_block = 0; m_rec = 0;
			m_next = std::async(std::launch::async, [this]() { return parse_window(); });
		}
	}

	// A block of plain four-line FASTQ records parsed where it lies: what FastaReader::read above does for the records of
	// such a block -- id / comment split, Casava chastity filter and "/1" suffix, masked-end trimming, case folding, -q
	// trimming, -Q masking -- straight from the block's bytes into the Block's strings (no stream, no string per line:
	// the stream route reads ~200 MB/s a thread).  Anything else the stream route treats specially -- a '#' line, a
	// SAM header line, a record that is not '@' ... '+' ..., an empty sequence, sequence and quality of different
	// lengths, a record cut short by the end of the file -- makes it give up (false; `out` is then rubbish): the caller
	// parses the block again through FastaReader, which handles it or reports it with its line number.
	// `nlines`: the block's line count (the getline calls the stream route would have made).
	static bool parse_fastq_block(const char* p, const char* end, const ReaderOptions& o, Block& out, size_t& nlines)
	{
		const auto space = [](unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }; // isspace in the C locale
		size_t lines = 0;
		const char* x = p;
		// the next line [a, b) without its end-of-line characters; false when the block has no bytes left
		const auto line = [&](const char*& a, const char*& b) -> bool {
			if (x >= end) return false;
			a = x;
			const char* nl = (const char*)memchr(x, '\n', (size_t)(end - x));
			b = nl ? nl : end;
			x = nl ? nl + 1 : end;
			lines++;
			while (b > a && b[-1] == '\r') b--;
			return true;
		};
		const unsigned qoff = o.qualityOffset > 0 ? (unsigned)o.qualityOffset : 33u;
		while (x < end) {
			if (*x != '@') return false;
			const char *h0, *h1, *s0, *s1, *p0, *p1, *q0, *q1;
			line(h0, h1);
			if (h1 - h0 > 3 && isalpha((unsigned char)h0[1]) && isalpha((unsigned char)h0[2]) && h0[3] == '\t') return false;
			const char* ie = h0 + 1;
			while (ie < h1 && !space((unsigned char)*ie)) ie++;
			const char* c0 = ie;
			while (c0 < h1 && space((unsigned char)*c0)) c0++;
			const size_t clen = (size_t)(h1 - c0), ilen = (size_t)(ie - (h0 + 1));
			bool suffix = false;
			if (clen > 3 && c0[1] == ':' && c0[3] == ':') { // Casava
				if (o.chastityFilter && c0[2] == 'Y') {
					const char *a, *b;
					line(a, b); line(a, b); line(a, b); // (the stream route does not look at them either)
					continue;
				}
				suffix = ilen > 2 && ie[-2] != '/';
			}
			if (!line(s0, s1) || !line(p0, p1) || p0 >= end || *p0 != '+') return false;
			if (!line(q0, q1)) q0 = q1 = end; // (a last record without its quality line: no quality, as the stream route has it)
			if (s0 == s1) return false;
			const bool hasq = q1 > q0;
			if (hasq && q1 - q0 != s1 - s0) return false;
			if (o.trimMasked) {
				const char *f = s0, *b = s1;
				while (f < s1 && *f >= 'a' && *f <= 'z') f++;
				while (b > s0 && b[-1] >= 'a' && b[-1] <= 'z') b--;
				if (b < f) b = f;
				if (hasq) { q1 = q0 + (b - s0); q0 += f - s0; }
				s0 = f; s1 = b;
			}
			bool one = false; // -q found no base worth keeping: the stream route leaves the first base (and its quality)
			if (o.qualityThreshold > 0 && hasq) {
				const int good = (int)qoff + o.qualityThreshold;
				const size_t n = (size_t)(q1 - q0);
				size_t front = 0, back = n;
				while (front < n && !((unsigned char)q0[front] >= good && (unsigned char)q0[front] <= '~')) front++;
				if (front == n) one = true;
				else {
					while (!((unsigned char)q0[back - 1] >= good && (unsigned char)q0[back - 1] <= '~')) back--;
					s1 = s0 + back; s0 += front; q1 = q0 + back; q0 += front;
				}
			}
			const size_t at = out.seqs.size();
			if (one) { if (s1 - s0 > 1) s1 = s0 + 1; if (q1 - q0 > 1) q1 = q0 + 1; }
			out.seqs.append(s0, (size_t)(s1 - s0));
			char* d = &out.seqs[0] + at;
			const size_t n = (size_t)(s1 - s0);
			if (o.foldCase) for (size_t i = 0; i < n; i++) { const unsigned char ch = (unsigned char)d[i]; d[i] = (char)(ch - ((ch >= 'a' && ch <= 'z') ? 32 : 0)); }
			if (o.internalQThreshold > 0 && q1 > q0) {
				const int good = (int)qoff + o.internalQThreshold;
				const size_t nq = (size_t)(q1 - q0);
				for (size_t i = 0; i < nq; i++) if (!((unsigned char)q0[i] >= good && (unsigned char)q0[i] <= '~')) d[i] = 'N';
			}
			out.seq_end.push_back(out.seqs.size());
			out.ids.append(h0 + 1, ilen);
			if (suffix) { out.ids += '/'; out.ids += c0[0]; }
			out.id_end.push_back(out.ids.size());
			out.comments.append(c0, clen);
			out.com_end.push_back(out.comments.size());
		}
		nlines = lines;
		return true;
	}
  private:
	// start of the first record at or after `from` (a line start), or `end` if there is none
	static size_t next_record(const char* p, size_t from, size_t end)
	{
		size_t a = from;
		if (a > 0 && p[a - 1] != '\n') { // move to a line start
			const char* nl = (const char*)memchr(p + a, '\n', end - a);
			if (!nl) return end;
			a = (size_t)(nl - p) + 1;
		}
		while (a < end) {
			const char* n1 = (const char*)memchr(p + a, '\n', end - a);
			if (!n1) return end;
			if (p[a] == '@') {
				const size_t l1 = (size_t)(n1 - p) + 1;
				const char* n2 = l1 < end ? (const char*)memchr(p + l1, '\n', end - l1) : nullptr;
				if (!n2) return end;
				const size_t l2 = (size_t)(n2 - p) + 1;
				if (l2 < end && p[l2] == '+') return a;
				if (l2 >= end) return end;
			}
			a = (size_t)(n1 - p) + 1;
		}
		return end;
	}
	// start of the LAST record that begins in p[0, end) (0 if none begins after offset 0)
	static size_t last_record(const char* p, size_t end)
	{
		size_t probe = end > (1u << 20) ? end - (1u << 20) : 0;
		for (;;) {
			size_t r = next_record(p, probe, end), last = 0;
			while (r < end) { last = r; r = next_record(p, r + 1, end); }
			if (last > 0 || probe == 0) return last;
			probe = probe > (16u << 20) ? probe - (16u << 20) : 0;
		}
	}
	struct Window { std::vector<Block> blocks; bool ok = false; };
	// reads and parses the next window (runs on a background thread; touches only m_f, m_buf, m_eof, m_lines)
	Window parse_window()
	{
		Window w;
		size_t end = 0;
		if (m_map) {
			// the next window of the mapping: up to the last record that starts in it (the whole rest at the end of the file)
			const size_t avail = m_map_size - m_pos;
			if (!avail) return w;
			for (size_t take = std::min(avail, m_window);; take = std::min(avail, take + m_window)) {
				end = take;
				if (take == avail) break;
				const size_t cut = last_record(m_map + m_pos, take);
				if (cut > 0) { end = cut; break; }
				// (no second record start in sight yet: look further)
			}
#ifdef MADV_POPULATE_READ
			{
				// one pass over the page tables instead of a fault per parser thread and page -- by a few threads, a part of the
				// window each: a single call takes 20 ms for 256 MB of page cache, four side by side 11 (and it stands in front of
				// every window's parse)
				// (the kernel's page size, not 4096: madvise wants page-aligned addresses; a kernel before 5.14 knows no
				// MADV_POPULATE_READ and says EINVAL -- the first call's answer switches the whole thing off, threads included)
				static const uintptr_t page = []() { const long v = sysconf(_SC_PAGESIZE); return (uintptr_t)(v > 0 ? v : 4096); }();
				static std::atomic<bool> populate_ok{ true };
				const uintptr_t a0 = (uintptr_t)(m_map + m_pos) & ~(page - 1), a1 = (uintptr_t)(m_map + m_pos) + end;
				static const unsigned max_parts = []() { const char* e = getenv("ABG_READER_POPULATE_THREADS"); return e ? (unsigned)std::max(1, atoi(e)) : 8u; }();
				const unsigned parts = (unsigned)std::min<size_t>(std::min(m_threads, max_parts), (a1 - a0) / ((size_t)8 << 20) + 1);
				const uintptr_t step = (((a1 - a0) / parts) + page - 1) & ~(page - 1);
				if (populate_ok.load(std::memory_order_relaxed)) {
					if (madvise((void*)a0, std::min(a1, a0 + step) - a0, MADV_POPULATE_READ) != 0 && (errno == EINVAL || errno == ENOSYS))
						populate_ok.store(false, std::memory_order_relaxed);
					else {
						std::vector<std::thread> pop;
						for (unsigned q = 1; q < parts; q++) {
							const uintptr_t b0 = a0 + q * step, b1 = std::min(a1, b0 + step);
							if (b0 < b1) pop.emplace_back([b0, b1]() { madvise((void*)b0, b1 - b0, MADV_POPULATE_READ); });
						}
						for (auto& t : pop) t.join();
					}
				}
			}
#endif
			// the pages before the PREVIOUS window's start are not needed again (that window may still be with its parser threads:
			// dropping a page under a reader would only cost a fault, but there is no reason to)
			{
				static const uintptr_t pg = []() { const long v = sysconf(_SC_PAGESIZE); return (uintptr_t)(v > 0 ? v : 4096); }();
				const uintptr_t lo = (uintptr_t)m_map + m_released, hi = ((uintptr_t)(m_map + m_prev_start)) & ~(pg - 1);
				if (hi > lo) { madvise((void*)lo, hi - lo, MADV_DONTNEED); m_released = hi - (uintptr_t)m_map; }
				m_prev_start = m_pos;
			}
		} else
		for (;;) {
			if (!m_eof) { // the unparsed tail of the previous window, then fresh bytes
				const size_t have = m_raw.n;
				m_raw.grow(have + m_window);
				const size_t got = read_parallel(m_raw.p + have, m_window);
				m_raw.n = have + got;
				if (got < m_window) m_eof = true;
			}
			if (buf_size() == 0) return w;
			end = buf_size();
			if (m_eof) break; // everything that is left is parsed
			// the last record that starts in the buffer may be cut short: it waits for the next window
			const size_t cut = last_record(buf_data(), end);
			if (cut > 0) { end = cut; break; }
			// (no second record start in sight yet: read on)
		}
		const char* p = buf_data();
		std::vector<size_t> start{ 0 };
		for (unsigned t = 1; t < m_threads; t++) {
			const size_t r = next_record(p, end / m_threads * t, end);
			if (r > start.back() && r < end) start.push_back(r);
		}
		start.push_back(end);
		const size_t nb = start.size() - 1;
		w.blocks.resize(nb);
		std::vector<size_t> nlines(nb, 0);
		std::vector<unsigned> line0(nb, m_lines);
		std::vector<char> fast(nb, 0);
		std::vector<std::thread> pool;
		// every block through parse_fastq_block first ...
		static const bool fast_on = []() { const char* e = getenv("ABG_READER_FAST"); return !(e && atoi(e) == 0); }();
		if (fast_on) {
			for (size_t b = 0; b < nb; b++)
				pool.emplace_back([&, b]() {
					const size_t len = start[b + 1] - start[b];
					Block out; // (filled locally: neighbouring Blocks share cache lines)
					out.seqs.reserve(len / 2 + 64); out.ids.reserve(len / 16 + 64);
					out.seq_end.reserve(len / 256 + 16); out.id_end.reserve(len / 256 + 16); out.com_end.reserve(len / 256 + 16);
					size_t n = 0;
					if (parse_fastq_block(p + start[b], p + start[b + 1], m_opt, out, n)) { nlines[b] = n; fast[b] = 1; w.blocks[b] = std::move(out); }
				});
			for (auto& t : pool) t.join();
			pool.clear();
		}
		// ... and the blocks it gave up on through FastaReader, which wants to know the number of its first line for its messages
		bool all_fast = true;
		for (size_t b = 0; b < nb; b++) all_fast = all_fast && fast[b];
		if (!all_fast) {
			for (size_t b = 0; b < nb; b++) {
				if (fast[b]) continue;
				pool.emplace_back([&, b]() {
					const char* q = p + start[b];
					const char* e = p + start[b + 1];
					size_t n = 0;
					for (const char* x = q; x < e && (x = (const char*)memchr(x, '\n', (size_t)(e - x))) != nullptr; x++) n++;
					nlines[b] = n;
				});
			}
			for (auto& t : pool) t.join();
			pool.clear();
		}
		for (size_t b = 1; b < nb; b++) line0[b] = line0[b - 1] + (unsigned)nlines[b - 1];
		for (size_t b = 0; b < nb && !all_fast; b++) {
			if (fast[b]) continue;
			pool.emplace_back([&, b]() {
				const size_t len = start[b + 1] - start[b];
				if (!len) return;
				FILE* f = fmemopen((void*)(p + start[b]), len, "r");
				if (!f) { fprintf(stderr, "error: fmemopen: %s\n", strerror(errno)); exit(EXIT_FAILURE); }
				FastaReader r(f, m_path, line0[b], m_opt);
				r.on_worker_thread();
				Block out; // (filled locally: neighbouring Blocks share cache lines)
				out.seqs.reserve(len / 2);
				std::string id, comment, s;
				while (r.read(id, comment, s)) {
					out.ids += id; out.id_end.push_back(out.ids.size());
					out.comments += comment; out.com_end.push_back(out.comments.size());
					out.seqs += s; out.seq_end.push_back(out.seqs.size());
				}
				w.blocks[b] = std::move(out);
			});
		}
		for (auto& t : pool) t.join();
		m_lines = line0[nb - 1] + (unsigned)nlines[nb - 1];
		// what was not parsed stays for the next window
		if (m_map) m_pos += end;
		else if (m_mem) m_buf.erase(0, end);
		else { memmove(m_raw.p, m_raw.p + end, m_raw.n - end); m_raw.n -= end; }
		w.ok = true;
		return w;
	}
	std::string m_path;
	ReaderOptions m_opt;
	unsigned m_threads;
	FastaReader* m_seq = nullptr;
	FILE* m_f = nullptr;
	size_t m_window = 0;
	bool m_eof = false, m_mem = false, m_done = false; // (m_mem: the whole inflated file is in m_buf)
	std::string m_buf;
	// The windows of a plain file: a buffer that is never zero-filled (a std::string's resize would
	// write the whole window once before fread writes it again), filled by several threads with
	// pread -- one thread copying out of the page cache is what the reader used to wait for.
	struct RawBuf {
		char* p = nullptr; size_t n = 0, cap = 0;
		~RawBuf() { free(p); }
		void grow(size_t want)
		{
			if (want <= cap) return;
			const size_t c = std::max(want, cap + cap / 2);
			char* q = (char*)realloc(p, c);
			if (!q) { fprintf(stderr, "error: out of memory reading sequences\n"); exit(EXIT_FAILURE); }
			p = q; cap = c;
		}
	} m_raw;
	size_t m_pos = 0; // file offset of the next byte to read into m_raw
	const char* m_map = nullptr; size_t m_map_size = 0; // the file mapped (m_pos: where the next window starts)
	size_t m_released = 0, m_prev_start = 0;            // ... its pages below m_released given back; where the window before this one started
	const char* buf_data() const { return m_map ? m_map + m_pos : m_mem ? m_buf.data() : m_raw.p; }
	size_t buf_size() const { return m_map ? m_map_size - m_pos : m_mem ? m_buf.size() : m_raw.n; }
	size_t read_parallel(char* dst, size_t want)
	{
		const int fd = fileno(m_f);
		const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(m_threads, 16u), want >> 22));
		std::vector<size_t> got(T, 0);
		std::vector<std::thread> pool;
		const size_t slice = (want + T - 1) / T;
		for (unsigned t = 0; t < T; t++)
			pool.emplace_back([&, t]() {
				const size_t a = std::min(want, t * slice), b = std::min(want, a + slice);
				size_t done = 0;
				while (a + done < b) {
					const ssize_t r = pread(fd, dst + a + done, b - a - done, (off_t)(m_pos + a + done));
					if (r < 0 && errno == EINTR) continue;
					if (r <= 0) break; // end of file (or an error: the parser then sees a truncated record)
					done += (size_t)r;
				}
				got[t] = done;
			});
		for (auto& th : pool) th.join();
		// the bytes read form a prefix: a slice that came up short ends the file
		size_t total = 0;
		for (unsigned t = 0; t < T; t++) {
			total += got[t];
			const size_t a = std::min(want, t * slice), b = std::min(want, a + slice);
			if (got[t] < b - a) break;
		}
		m_pos += total;
		return total;
	}
	unsigned m_lines = 0;
	std::vector<Block> m_blocks;
	size_t m_block = 0, m_rec = 0;
	std::future<Window> m_next;
};

} // namespace abghost
