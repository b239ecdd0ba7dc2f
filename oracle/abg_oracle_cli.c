/*
 * abg_oracle_cli.c -- command-line front end of the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see abg_oracle.h).  Mimics the reference's
 * `abyss-bloom-dbg -k -b -H --kc -t [-K|-s|--qr-seed] [--read-log F] [-T F] files...`
 * (BloomDBG/bloom-dbg.cc:389-558) for plain 4-line FASTQ / 2-line FASTA input
 * with no quality trimming, so that its stdout / read log / trace can be
 * diffed byte-for-byte against oracle/_ref/abyss-bloom-dbg -j1.
 */
#define _GNU_SOURCE
#include "abg_oracle.h"
#include <ctype.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	char* seqs; uint64_t* offs; char** ids; uint64_t n, cap, bytes, bcap;
} reads_t;

static void reads_add(reads_t* r, const char* id, const char* seq, size_t len)
{
	if (r->n + 2 > r->cap) {
		r->cap = r->cap ? r->cap * 2 : 1024;
		r->offs = realloc(r->offs, (r->cap + 1) * sizeof(uint64_t));
		r->ids = realloc(r->ids, r->cap * sizeof(char*));
	}
	if (r->bytes + len + 1 > r->bcap) {
		r->bcap = (r->bcap + len + 1) * 2;
		r->seqs = realloc(r->seqs, r->bcap);
	}
	if (r->n == 0) r->offs[0] = 0;
	/* FastaReader::FOLD_CASE: upper-case the sequence (DataLayer/FastaReader.cpp) */
	for (size_t i = 0; i < len; i++) r->seqs[r->bytes + i] = (char)toupper((unsigned char)seq[i]);
	r->bytes += len;
	r->ids[r->n] = strdup(id);
	r->n++;
	r->offs[r->n] = r->bytes;
}

static void read_file(reads_t* r, const char* path)
{
	FILE* f = fopen(path, "r");
	if (!f) { fprintf(stderr, "error: `%s': cannot open\n", path); exit(1); }
	char* line = NULL; size_t cap = 0; ssize_t n;
	while ((n = getline(&line, &cap, f)) > 0) {
		if (line[0] != '@' && line[0] != '>') continue;
		int fq = line[0] == '@';
		char id[1024]; size_t j = 0;
		for (ssize_t i = 1; i < n && !isspace((unsigned char)line[i]) && j < sizeof id - 1; i++) id[j++] = line[i];
		id[j] = 0;
		n = getline(&line, &cap, f);
		if (n <= 0) break;
		while (n > 0 && (line[n-1] == '\n' || line[n-1] == '\r')) n--;
		reads_add(r, id, line, (size_t)n);
		if (fq) { getline(&line, &cap, f); getline(&line, &cap, f); }
	}
	free(line);
	fclose(f);
}

static const char* ext_str(int c)
{
	switch (c) { case ORC_ER_AMBI_IN: return "AMBI_IN"; case ORC_ER_AMBI_OUT: return "AMBI_OUT";
	case ORC_ER_DEAD_END: return "DEAD_END"; case ORC_ER_CYCLE: return "CYCLE"; default: return "LENGTH_LIMIT"; }
}
static const char* rr_str(int r)
{
	static const char* s[] = { "NA", "SHORTER_THAN_K", "NON_ACGT", "BLUNT_END", "NOT_SOLID",
		"ALL_KMERS_VISITED", "ALL_BRANCH_KMERS_VISITED", "GENERATED_CONTIGS" };
	return s[r];
}

typedef struct { reads_t* r; FILE* trace; unsigned k; } out_t;

static void on_contig(void* user, const orc_contig* c)
{
	out_t* o = (out_t*)user;
	const char* rid = o->r->ids[c->read_index];
	if (!c->redundant) /* printContig, bloom-dbg.h:455-487 */
		printf(">%llu %u %u read:%s\n%s\n", (unsigned long long)c->contig_id, c->length, c->coverage, rid, c->seq);
	if (o->trace) { /* ContigRecord operator<<, bloom-dbg.h:229-254 */
		if (c->redundant) fprintf(o->trace, "NA\t"); else fprintf(o->trace, "%llu\t", (unsigned long long)c->contig_id);
		/* note: rec.length is left uninitialised by the reference for redundant contigs */
		fprintf(o->trace, "%u\t%d\t%s\t", c->length, c->redundant, rid);
		if (c->left_ext > 0) fprintf(o->trace, "%s\t%u\t", ext_str(c->left_code), c->left_ext); else fprintf(o->trace, "NA\tNA\t");
		if (c->right_ext > 0) fprintf(o->trace, "%s\t%u\t", ext_str(c->right_code), c->right_ext); else fprintf(o->trace, "NA\tNA\t");
		fprintf(o->trace, "READ\t%u\t%.*s\n", o->k, (int)o->k, c->seed);
	}
}

/* SIToBytes, Common/StringUtil.h:181-219 (k/M/G = 2^10/20/30) */
static uint64_t si_to_bytes(const char* s)
{
	char* end; double x = strtod(s, &end);
	switch (toupper((unsigned char)*end)) { case 'K': x *= 1024.0; break; case 'M': x *= 1048576.0; break; case 'G': x *= 1073741824.0; break; default: break; }
	return (uint64_t)x;
}

int main(int argc, char** argv)
{
	unsigned k = 0, H = 4, kc = 2, trim = ~0u, K = 0, qr = 0;
	uint64_t B = 0, counters = 0;
	const char* seed = NULL; const char* readlog = NULL; const char* tracep = NULL; const char* dump = NULL;
	static const struct option lo[] = { {"kc", 1, 0, 1000}, {"qr-seed", 1, 0, 1001}, {"read-log", 1, 0, 1002},
		{"counters", 1, 0, 1003}, {"dump-counters", 1, 0, 1004}, {0,0,0,0} };
	int ch;
	while ((ch = getopt_long(argc, argv, "k:b:H:t:K:s:T:j:q:v", lo, NULL)) != -1) switch (ch) {
		case 'k': k = atoi(optarg); break; case 'b': B = si_to_bytes(optarg); break;
		case 'H': H = atoi(optarg); break; case 't': trim = atoi(optarg); break;
		case 'K': K = atoi(optarg); break; case 's': seed = optarg; break; case 'T': tracep = optarg; break;
		case 1000: kc = atoi(optarg); break; case 1001: qr = atoi(optarg); break; case 1002: readlog = optarg; break;
		case 1003: counters = strtoull(optarg, 0, 10); break; case 1004: dump = optarg; break;
		default: break; }
	if (!k || (!B && !counters) || optind >= argc) { fprintf(stderr, "usage: abg_oracle -k K -b B [-H n] [--kc n] [-t n] reads...\n"); return 1; }
	if (trim == ~0u) trim = k;
	char mask[ORC_MAX_KMER + 1] = "";
	if (K) orc_seed_kmer_pair(k, K, mask); else if (qr) orc_seed_qr_pair(k, qr, mask); else if (seed) strncpy(mask, seed, ORC_MAX_KMER);
	if (!counters) counters = orc_counters_for_budget(B);
	orc_ctx* c = orc_create(k, H, kc, trim, counters, mask);
	if (!c) { fprintf(stderr, "abg_oracle: bad parameters\n"); return 1; }
	reads_t load = {0}, asmr = {0};
	int i = optind;
	for (; i < argc; i++) { if (!strcmp(argv[i], ":")) { i++; break; } read_file(&load, argv[i]); } /* BloomIO.h:102-115 */
	if (i < argc) { for (; i < argc; i++) read_file(&asmr, argv[i]); } else asmr = load;
	orc_load_seqs(c, load.seqs, load.offs, load.n);
	fprintf(stderr, "counters=%llu popcount=%llu filtered=%llu\n", (unsigned long long)orc_size(c),
		(unsigned long long)orc_popcount(c), (unsigned long long)orc_filtered_popcount(c));
	if (dump) { FILE* f = fopen(dump, "wb"); fwrite(orc_counters(c), 1, orc_size(c), f); fclose(f); }
	uint8_t* results = malloc(asmr.n ? asmr.n : 1);
	out_t o = { &asmr, NULL, k };
	if (tracep) { o.trace = fopen(tracep, "w"); fprintf(o.trace, "contig_id\tlength\tredundant\tread_id\tleft_result\tleft_extension\tright_result\tright_extension\tseed_type\tseed_length\tseed\n"); }
	orc_assemble(c, asmr.seqs, asmr.offs, asmr.n, results, on_contig, &o);
	if (o.trace) fclose(o.trace);
	if (readlog) { FILE* f = fopen(readlog, "w"); fprintf(f, "read_id\tresult\n"); for (uint64_t j = 0; j < asmr.n; j++) fprintf(f, "%s\t%s\n", asmr.ids[j], rr_str(results[j])); fclose(f); }
	uint64_t solid, vis, proc, bases, next;
	orc_counters_get(c, &solid, &vis, &proc, &bases, &next);
	fprintf(stderr, "Processed %llu reads, solid %llu, visited %llu; assembled %llu bp in %llu contigs\n",
		(unsigned long long)proc, (unsigned long long)solid, (unsigned long long)vis, (unsigned long long)bases, (unsigned long long)next);
	orc_destroy(c);
	return 0;
}
