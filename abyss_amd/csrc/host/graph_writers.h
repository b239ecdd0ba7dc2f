// graph_writers.h -- the six contig-graph output formats of the reference (opt::format), shared by the drop-in AdjList and
// abyss-rresolver-short.  Restates the writers of
//   Graph/AdjIO.h:32-66  Graph/DotIO.h:14-101  Graph/GfaIO.h:15-211  Graph/AsqgIO.h:13-70  Graph/SAMIO.h:18-70
// over any graph G that offers
//   unsigned k;  uint64_t nv();            vertex u = 2 * contig + sense
//   bool removed(u);                       (vertex_removed: skipped everywhere)
//   const std::string& cname(u);           contig name;  vname(u) = cname + '+' / '-'
//   unsigned len(u), cov(u);               ContigProperties of the vertex
//   for_out(u, f(v, d));                   out-edges in adjacency order with their distance
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace abgio {

enum Format { ADJ = 0, ASQG, DOT, GFA1, GFA2, SAM }; // Graph/Options.h (the ones these programs offer)

#define ABG_IO_VERSION "2.3.10"

struct Out {
	std::string buf;
	FILE* f;
	explicit Out(FILE* f) : f(f) { buf.reserve(1u << 20); }
	~Out() { flush(); }
	void flush() { if (!buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
	void room() { if (buf.size() > (1u << 20) - 4096) flush(); }
	Out& operator<<(const std::string& s) { buf += s; room(); return *this; }
	Out& operator<<(const char* s) { buf += s; return *this; }
	Out& operator<<(char c) { buf += c; return *this; }
	Out& operator<<(unsigned long long x) { buf += std::to_string(x); return *this; }
	Out& operator<<(unsigned long x) { buf += std::to_string(x); return *this; }
	Out& operator<<(unsigned x) { buf += std::to_string(x); return *this; }
	Out& operator<<(int x) { buf += std::to_string(x); return *this; }
};

template <class G> std::string vname(const G& g, uint64_t u) { return g.cname(u) + ((u & 1) ? '-' : '+'); }

template <class G> void write_adj(Out& out, const G& g) // Graph/AdjIO.h:32-66
{
	const int def = -(int)(g.k - 1);
	for (uint64_t u = 0; u < g.nv(); u++) {
		if (g.removed(u)) continue;
		const unsigned sense = u & 1;
		if (!sense) out << g.cname(u) << ' ' << g.len(u) << ' ' << g.cov(u);
		out << "\t;";
		g.for_out(u, [&](uint32_t v, int d) {
			out << ' ' << vname(g, v ^ sense);
			if (d != def) out << " [d=" << d << ']';
		});
		if (sense) out << '\n';
	}
}
template <class G> void write_dot(Out& out, const G& g) // Graph/DotIO.h:14-101
{
	const int def = -(int)(g.k - 1);
	out << "digraph adj {\n";
	if (g.k > 0) out << "graph [k=" << g.k << "]\nedge [d=" << def << "]\n";
	for (uint64_t u = 0; u < g.nv(); u++) {
		if (g.removed(u)) continue;
		out << '"' << vname(g, u) << "\" [l=" << g.len(u) << " C=" << g.cov(u) << "]\n";
	}
	for (uint64_t u = 0; u < g.nv(); u++) {
		if (g.removed(u)) continue;
		g.for_out(u, [&](uint32_t v, int d) {
			out << '"' << vname(g, u) << "\" -> \"" << vname(g, v) << '"';
			if (d != def) out << " [d=" << d << ']';
			out << '\n';
		});
	}
	out << "}\n";
}
template <class G> void write_gfa1(Out& out, const G& g) // Graph/GfaIO.h:15-66
{
	out << "H\tVN:Z:1.0\n";
	for (uint64_t u = 0; u < g.nv(); u += 2) {
		if (g.removed(u)) continue;
		out << "S\t" << g.cname(u) << "\t*\tLN:i:" << g.len(u);
		if (g.cov(u) > 0) out << "\tKC:i:" << g.cov(u);
		out << '\n';
	}
	for (uint64_t u = 0; u < g.nv(); u++) {
		if (g.removed(u)) continue;
		g.for_out(u, [&](uint32_t v, int d) {
			if (u > (uint64_t)(v ^ 1u)) return; // only the canonical edge
			out << "L\t" << g.cname(u) << '\t' << ((u & 1) ? '-' : '+') << '\t' << g.cname(v) << '\t' << ((v & 1) ? '-' : '+');
			if (d <= 0) out << '\t' << -d << "M\n"; else out << "\t*\n";
		});
	}
}
template <class G> void write_gfa2(Out& out, const G& g) // Graph/GfaIO.h:69-118,129-155,191-211
{
	out << "H\tVN:Z:2.0\n";
	for (uint64_t u = 0; u < g.nv(); u += 2) {
		if (g.removed(u)) continue;
		out << "S\t" << g.cname(u) << '\t' << g.len(u) << "\t*";
		if (g.cov(u) > 0) out << "\tKC:i:" << g.cov(u);
		out << '\n';
	}
	for (uint64_t u = 0; u < g.nv(); u++) {
		if (g.removed(u)) continue;
		g.for_out(u, [&](uint32_t v, int d) {
			if (u > (uint64_t)(v ^ 1u)) return;
			const unsigned overlap = (unsigned)-d, ulen = g.len(u), vlen = g.len(v);
			const bool us = u & 1, vs = v & 1;
			const unsigned ustart = us ? 0 : ulen - overlap, uend = us ? overlap : ulen;
			const unsigned vstart = !vs ? 0 : vlen - overlap, vend = !vs ? overlap : vlen;
			out << "E\t*\t" << vname(g, u) << '\t' << vname(g, v);
			out << '\t' << ustart; if (ustart == ulen) out << '$';
			out << '\t' << uend; if (uend == ulen) out << '$';
			out << '\t' << vstart; if (vstart == vlen) out << '$';
			out << '\t' << vend; if (vend == vlen) out << '$';
			out << '\t' << overlap << "M\n";
		});
	}
}
template <class G> void write_asqg(Out& out, const G& g) // Graph/AsqgIO.h:13-70
{
	out << "HT\tVN:i:1\n";
	for (uint64_t u = 0; u < g.nv(); u += 2) {
		if (g.removed(u)) continue;
		out << "VT\t" << g.cname(u) << "\t*\tLN:i:" << g.len(u);
		if (g.cov(u) > 0) out << "\tKC:i:" << g.cov(u);
		out << '\n';
	}
	for (uint64_t u = 0; u < g.nv(); u++) {
		if (g.removed(u)) continue;
		g.for_out(u, [&](uint32_t v, int d) {
			if (u > (uint64_t)(v ^ 1u)) return;
			const unsigned overlap = (unsigned)-d, ulen = g.len(u), vlen = g.len(v);
			const bool us = u & 1, vs = v & 1;
			out << "ED\t" << g.cname(u) << ' ' << g.cname(v)
			    << ' ' << (us ? 0u : ulen - overlap) << ' ' << (int)((us ? overlap : ulen) - 1) << ' ' << ulen
			    << ' ' << (!vs ? 0u : vlen - overlap) << ' ' << (int)((!vs ? overlap : vlen) - 1) << ' ' << vlen
			    << ' ' << (us != vs ? 1 : 0) << " -1\n";
		});
	}
}
template <class G> void write_sam(Out& out, const G& g, const std::string& program, const std::string& commandLine) // Graph/SAMIO.h:18-70
{
	out << "@HD\tVN:1.0\n@PG\tID:" << program << "\tVN:" ABG_IO_VERSION "\tCL:" << commandLine << '\n';
	for (uint64_t u = 0; u < g.nv(); u += 2) {
		if (g.removed(u)) continue;
		out << "@SQ\tSN:" << g.cname(u) << "\tLN:" << g.len(u);
		if (g.cov(u) > 0) out << "\tXC:" << g.cov(u);
		out << '\n';
	}
	for (uint64_t u = 0; u < g.nv(); u++) {
		if (g.removed(u)) continue;
		g.for_out(u, [&](uint32_t v, int d) {
			if (d > 0) return;
			const bool us = u & 1, vs = v & 1;
			const unsigned alen = (unsigned)-d, ulen = g.len(u), vlen = g.len(v);
			const unsigned pos = 1 + (us ? 0 : ulen - alen), clip = vlen - alen;
			out << g.cname(v) << '\t' << (us == vs ? 0 : 0x10) << '\t' << g.cname(u) << '\t' << pos << "\t255\t";
			if (us) out << clip << 'H' << alen << "M\t"; else out << alen << 'M' << clip << "H\t";
			out << "*\t0\t0\t*\t*\n";
		});
	}
}
// write_graph, Graph/GraphIO.h:19-43
template <class G> void write_graph(Out& out, const G& g, int format, const std::string& program, const std::string& commandLine)
{
	switch (format) {
	case ADJ: write_adj(out, g); break;
	case DOT: write_dot(out, g); break;
	case GFA1: write_gfa1(out, g); break;
	case GFA2: write_gfa2(out, g); break;
	case ASQG: write_asqg(out, g); break;
	case SAM: write_sam(out, g, program, commandLine); break;
	}
}

// printGraphStats (Graph/GraphUtil.h:43-64) with Histogram::barplot (Common/Histogram.cpp:45-95): v vertices that are not
// removed, e edges, h = histogram of their out-degrees
inline void print_graph_stats(FILE* out, unsigned v, unsigned e, const std::map<int, uint64_t>& h)
{
	auto sig = [](float x, int prec) { // operator<<(float) under setprecision(prec): %g
		char b[64];
		snprintf(b, sizeof b, "%.*g", prec, x);
		return std::string(b);
	};
	fprintf(out, "V=%u E=%u E/V=%s\n", v, e, sig((float)e / v, 3).c_str());
	if (h.empty()) return;
	const int mn = h.begin()->first, mx = h.rbegin()->first;
	std::vector<uint64_t> bins;
	{
		const unsigned nb = (unsigned)mx + 1;
		const int per = (int)ceilf((float)(mx - mn) / nb);
		int next = mn + per;
		uint64_t count = 0;
		for (auto& kv : h) {
			if (kv.first >= next) { bins.push_back(count); count = 0; next += per; }
			count += kv.second;
		}
		if (count > 0) bins.push_back(count);
	}
	static const char* bars[10] = { " ", "_", "\342\226\201", "\342\226\202", "\342\226\203", "\342\226\204", "\342\226\205", "\342\226\206", "\342\226\207", "\342\226\210" };
	std::vector<std::string> cells;
	const uint64_t top = 1 + *std::max_element(bins.begin(), bins.end());
	for (uint64_t b : bins) cells.push_back(bars[10 * b / top]);
	while (!cells.empty() && cells.back() == " ") cells.pop_back();
	std::string plot;
	for (auto& c : cells) plot += c;
	uint64_t n = 0, n0 = 0, n1 = 0, n234 = 0;
	for (auto& kv : h) {
		n += kv.second;
		if (kv.first == 0) n0 += kv.second;
		else if (kv.first == 1) n1 += kv.second;
		else if (kv.first <= 4) n234 += kv.second;
	}
	const uint64_t n5 = n - (n0 + n1 + n234);
	fprintf(out, "Degree: %s\n        01234\n0: %s%% 1: %s%% 2-4: %s%% 5+: %s%% max: %d\n", plot.c_str(),
	    sig((float)100 * n0 / n, 2).c_str(), sig((float)100 * n1 / n, 2).c_str(), sig((float)100 * n234 / n, 2).c_str(),
	    sig((float)100 * n5 / n, 2).c_str(), mx);
}

} // namespace abgio
