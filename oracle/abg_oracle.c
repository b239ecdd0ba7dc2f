/*
 * abg_oracle.c -- CPU restatement of the abyss-bloom-dbg hot path (ABySS 2.3.10).
 *
 * TEST INFRASTRUCTURE ONLY: see abg_oracle.h.  Plain C, single thread, value
 * semantics; it restates the reference's algorithm (file:line cited at every
 * function) and is pinned against the reference itself (oracle/_ref).  It is
 * never linked into, called from, or used as a fallback by the product.
 */
#define _GNU_SOURCE
#include "abg_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { FORWARD = 0, REVERSE = 1 };    /* Graph/Path.h:37 */
enum { SENSE = 0, ANTISENSE = 1 };    /* Common/Sense.h */
static const char BASE_CHARS[4] = { 'A', 'C', 'G', 'T' }; /* RollingBloomDBG.h:26 */

/* ------------------------------------------------------------------ ntHash */

/* vendor/nthash/nthash.hpp:18-29 */
#define MULTISHIFT 27
static const uint64_t MULTISEED = 0x90b45d39fb6da1faULL;
static const uint64_t SEED_A = 0x3c8bfbb395c60474ULL;
static const uint64_t SEED_C = 0x3193c18562a02b4cULL;
static const uint64_t SEED_G = 0x20323ed082572324ULL;
static const uint64_t SEED_T = 0x295549f54be24456ULL;

/* seedTab, nthash.hpp:31-64: the non-zero entries only. */
static inline uint64_t
seed_tab(unsigned char c)
{
	switch (c) {
	case 1: return SEED_T;
	case 3: return SEED_G;
	case 4: case 5: return SEED_A;
	case 7: return SEED_C;
	case 'A': case 'a': return SEED_A;
	case 'C': case 'c': return SEED_C;
	case 'G': case 'g': return SEED_G;
	case 'T': case 't': case 'U': case 'u': return SEED_T;
	default: return 0;
	}
}
/* complement seed: seedTab[c & cpOff], nthash.hpp:16,236 */
static inline uint64_t
seed_rc(unsigned char c)
{
	return seed_tab(c & 0x07);
}

/* rol1 + swapbits033, nthash.hpp:186-207: rotate the low 33 bits and the high
 * 31 bits left by one, independently. */
static inline uint64_t
srol1(uint64_t v)
{
	uint64_t h = (v << 1) | (v >> 63);
	uint64_t x = (h ^ (h >> 33)) & 1;
	return h ^ (x | (x << 33));
}
/* ror1 + swapbits3263, nthash.hpp:191-217 */
static inline uint64_t
sror1(uint64_t v)
{
	uint64_t h = (v >> 1) | (v << 63);
	uint64_t x = ((h >> 32) ^ (h >> 63)) & 1;
	return h ^ ((x << 32) | (x << 63));
}
/* msTab31l[c][n%31] | msTab33r[c][n%33], nthash.hpp:66-183 (rol31/rol33 :196-205) */
static inline uint64_t
srol_n(uint64_t v, unsigned n)
{
	uint64_t lo = v & 0x1FFFFFFFFULL, hi = v >> 33;
	unsigned a = n % 33, b = n % 31;
	if (a)
		lo = ((lo << a) | (lo >> (33 - a))) & 0x1FFFFFFFFULL;
	if (b)
		hi = ((hi << b) | (hi >> (31 - b))) & 0x7FFFFFFFULL;
	return (hi << 33) | lo;
}

/* NTF64 / NTR64 base-kmer forms, nthash.hpp:220-239 */
static uint64_t
ntf64(const char* s, unsigned k)
{
	uint64_t h = 0;
	for (unsigned i = 0; i < k; i++)
		h = srol1(h) ^ seed_tab((unsigned char)s[i]);
	return h;
}
static uint64_t
ntr64(const char* s, unsigned k)
{
	uint64_t h = 0;
	for (unsigned i = 0; i < k; i++)
		h = srol1(h) ^ seed_rc((unsigned char)s[k - 1 - i]);
	return h;
}
/* NTE64, nthash.hpp:337-342 (note precedence: i ^ (k * multiSeed)) */
static inline uint64_t
nte64(uint64_t h, unsigned k, unsigned i)
{
	uint64_t t = h;
	t *= ((uint64_t)i ^ ((uint64_t)k * MULTISEED));
	t ^= t >> MULTISHIFT;
	return t;
}

/* ------------------------------------------------------------------ context */

typedef struct {
	uint64_t fh, rh, h; /* RollingHash m_hash1, m_rcHash1, m_hash (RollingHash.h:211-219) */
	char s[ORC_MAX_KMER + 1];
} vtx; /* RollingBloomDBGVertex, RollingBloomDBG.h:33-38, with value semantics */

typedef struct {
	vtx* slots;
	uint8_t* used;
	size_t cap, n;
} vset;

struct orc_ctx {
	unsigned k, nh, kc, trim;
	uint64_t m;        /* counters == visited bits */
	uint8_t* cnt;      /* CountingBloomFilter<uint8_t>::m_filter */
	uint8_t* vis;      /* BloomFilter::m_filter (assembledKmerSet) */
	char mask[ORC_MAX_KMER + 1];
	int has_mask;
	/* AssemblyCounters.h:15-31 */
	uint64_t solid_reads, visited_reads, reads_processed, bases_assembled, contig_id;
	vset* contig_end_kmers; /* bloom-dbg.h:992 */
};

/* maskHash, nthash.hpp:537-547 */
static uint64_t
mask_hash(const orc_ctx* c, uint64_t fk, uint64_t rk, const char* kmer)
{
	uint64_t fs = fk, rs = rk;
	unsigned k = c->k;
	for (unsigned i = 0; i < k; i++) {
		if (c->mask[i] != '1') {
			fs ^= srol_n(seed_tab((unsigned char)kmer[i]), k - 1 - i);
			rs ^= srol_n(seed_rc((unsigned char)kmer[i]), i);
		}
	}
	return (rs < fs) ? rs : fs;
}

/* canonicalHash + optional maskHash, RollingHash.h:28-31,74-79 */
static inline void
vtx_finish(const orc_ctx* c, vtx* v)
{
	v->h = (v->rh < v->fh) ? v->rh : v->fh;
	if (c->has_mask)
		v->h = mask_hash(c, v->fh, v->rh, v->s);
}

/* RollingHash::reset, RollingHash.h:69-80 */
static void
vtx_init(const orc_ctx* c, vtx* v, const char* kmer)
{
	memcpy(v->s, kmer, c->k);
	v->s[c->k] = 0;
	v->fh = ntf64(v->s, c->k);
	v->rh = ntr64(v->s, c->k);
	vtx_finish(c, v);
}

/* Vertex::shift, RollingBloomDBG.h:55-63 = RollingHash::rollRight/rollLeft
 * (RollingHash.h:88-124; NTC64 nthash.hpp:242-257,275-279; NTC64L :282-304)
 * followed by LightweightKmer::shift (LightweightKmer.h:52-62). */
static void
vtx_shift(const orc_ctx* c, vtx* v, int sense, char in)
{
	unsigned k = c->k;
	if (sense == SENSE) {
		unsigned char out = (unsigned char)v->s[0];
		v->fh = srol1(v->fh) ^ seed_tab((unsigned char)in) ^ srol_n(seed_tab(out), k);
		v->rh = sror1(v->rh ^ srol_n(seed_rc((unsigned char)in), k) ^ seed_rc(out));
		memmove(v->s, v->s + 1, k - 1);
		v->s[k - 1] = in;
	} else {
		unsigned char out = (unsigned char)v->s[k - 1];
		v->fh = sror1(v->fh ^ srol_n(seed_tab((unsigned char)in), k) ^ seed_tab(out));
		v->rh = srol1(v->rh) ^ seed_rc((unsigned char)in) ^ srol_n(seed_rc(out), k);
		memmove(v->s + 1, v->s, k - 1);
		v->s[0] = in;
	}
	vtx_finish(c, v);
}

/* u.clone() + shift(dir) + setLastBase(dir, base): the neighbour enumerated by
 * out_edge_iterator / in_edge_iterator (RollingBloomDBG.h:302-427).  setLastBase
 * (RollingHash.h:175-193) rolls 'A' out and `base` in, which is the hash of the
 * shifted k-mer ending (starting) in `base`. */
static void
vtx_neighbour(const orc_ctx* c, const vtx* u, int sense, char base, vtx* out)
{
	*out = *u;
	vtx_shift(c, out, sense, base);
}

static char
complement_base(char ch)
{
	/* complementBaseChar, Common/Sequence.cpp:20-47 (upper-case subset + N) */
	switch (toupper((unsigned char)ch)) {
	case 'A': return 'T';
	case 'C': return 'G';
	case 'G': return 'C';
	case 'T': return 'A';
	case 'N': return 'N';
	default:
		fprintf(stderr, "oracle: unexpected character `%c'\n", ch);
		abort();
	}
}

/* Vertex::reverseComplement, RollingBloomDBG.h:71-75; LightweightKmer.h:114-129;
 * RollingHash::reverseComplement RollingHash.h:202-205 */
static void
vtx_revcomp(const orc_ctx* c, vtx* v)
{
	unsigned k = c->k;
	for (unsigned i = 0; i < k / 2; i++) {
		char tmp = complement_base(v->s[i]);
		v->s[i] = complement_base(v->s[k - i - 1]);
		v->s[k - i - 1] = tmp;
	}
	if (k % 2 == 1 && k > 1)
		v->s[k / 2] = complement_base(v->s[k / 2]);
	uint64_t t = v->fh;
	v->fh = v->rh;
	v->rh = t;
}

/* LightweightKmer::isCanonical, LightweightKmer.h:88-101 */
static int
kmer_is_canonical(const char* s, unsigned k)
{
	for (unsigned i = 0; i < k / 2; i++) {
		char c1 = (char)toupper((unsigned char)s[i]);
		char c2 = complement_base((char)toupper((unsigned char)s[k - i - 1]));
		if (c1 > c2)
			return 0;
		else if (c1 < c2)
			return 1;
	}
	return 1;
}

/* RollingBloomDBGVertex::compare, RollingBloomDBG.h:114-159 */
static int
vtx_compare(const orc_ctx* c, const vtx* a, const vtx* b)
{
	int k = (int)c->k;
	int rc1 = !kmer_is_canonical(a->s, c->k);
	int rc2 = !kmer_is_canonical(b->s, c->k);
	int end1 = rc1 ? -1 : k, end2 = rc2 ? -1 : k;
	int inc1 = rc1 ? -1 : 1, inc2 = rc2 ? -1 : 1;
	int pos1 = rc1 ? k - 1 : 0, pos2 = rc2 ? k - 1 : 0;
	for (; pos1 != end1 && pos2 != end2; pos1 += inc1, pos2 += inc2) {
		char c1 = (char)toupper((unsigned char)a->s[pos1]);
		char c2 = (char)toupper((unsigned char)b->s[pos2]);
		if (c->has_mask && c->mask[pos1] != '1')
			continue;
		if (rc1)
			c1 = complement_base(c1);
		if (rc2)
			c2 = complement_base(c2);
		if (c1 > c2)
			return 1;
		if (c1 < c2)
			return -1;
	}
	return 0;
}

/* RollingBloomDBGVertex::operator==, RollingBloomDBG.h:92-99 (RollingHash::operator==
 * compares k and the canonical hash seed, RollingHash.h:150-160) */
static int
vtx_eq(const orc_ctx* c, const vtx* a, const vtx* b)
{
	if (a->h != b->h)
		return 0;
	return vtx_compare(c, a, b) == 0;
}

/* LightweightKmer::operator==, LightweightKmer.h:131-146 */
static int
kmer_eq(const orc_ctx* c, const char* a, const char* b)
{
	if (!c->has_mask)
		return !memcmp(a, b, c->k);
	for (unsigned i = 0; i < c->k; i++)
		if (c->mask[i] != '0' && a[i] != b[i])
			return 0;
	return 1;
}

/* ------------------------------------------------- unordered_set<Vertex> */
/* Keyed by hash<Vertex> = canonical hash seed (RollingBloomDBG.h:163-171) with
 * equality vtx_eq.  Linear probing with backward-shift deletion. */

static void
vset_init(vset* s)
{
	s->slots = NULL;
	s->used = NULL;
	s->cap = 0;
	s->n = 0;
}
static void
vset_free(vset* s)
{
	free(s->slots);
	free(s->used);
	vset_init(s);
}
static size_t
vset_find_slot(const orc_ctx* c, const vset* s, const vtx* v, int* found)
{
	size_t mask = s->cap - 1;
	size_t i = (size_t)(v->h * 0x9E3779B97F4A7C15ULL >> 20) & mask;
	while (s->used[i]) {
		if (vtx_eq(c, &s->slots[i], v)) {
			*found = 1;
			return i;
		}
		i = (i + 1) & mask;
	}
	*found = 0;
	return i;
}
static int
vset_contains(const orc_ctx* c, const vset* s, const vtx* v)
{
	int found;
	if (s->n == 0)
		return 0;
	vset_find_slot(c, s, v, &found);
	return found;
}
static void vset_grow(const orc_ctx* c, vset* s);
/* returns 1 if inserted, 0 if already present */
static int
vset_insert(const orc_ctx* c, vset* s, const vtx* v)
{
	int found;
	if (s->cap == 0 || (s->n + 1) * 2 > s->cap)
		vset_grow(c, s);
	size_t i = vset_find_slot(c, s, v, &found);
	if (found)
		return 0;
	s->slots[i] = *v;
	s->used[i] = 1;
	s->n++;
	return 1;
}
static void
vset_grow(const orc_ctx* c, vset* s)
{
	vset old = *s;
	s->cap = old.cap ? old.cap * 2 : 16;
	s->slots = (vtx*)malloc(s->cap * sizeof(vtx));
	s->used = (uint8_t*)calloc(s->cap, 1);
	s->n = 0;
	if (!s->slots || !s->used) {
		fprintf(stderr, "oracle: out of memory\n");
		abort();
	}
	for (size_t i = 0; i < old.cap; i++)
		if (old.used[i])
			vset_insert(c, s, &old.slots[i]);
	free(old.slots);
	free(old.used);
}
static void
vset_erase(const orc_ctx* c, vset* s, const vtx* v)
{
	int found;
	if (s->n == 0)
		return;
	size_t i = vset_find_slot(c, s, v, &found);
	if (!found)
		return;
	size_t mask = s->cap - 1;
	s->used[i] = 0;
	s->n--;
	size_t j = i;
	for (;;) {
		j = (j + 1) & mask;
		if (!s->used[j])
			break;
		size_t home = (size_t)(s->slots[j].h * 0x9E3779B97F4A7C15ULL >> 20) & mask;
		/* move j back to i if its home is cyclically outside (i, j] */
		if ((i <= j) ? (home <= i || home > j) : (home <= i && home > j)) {
			s->slots[i] = s->slots[j];
			s->used[i] = 1;
			s->used[j] = 0;
			i = j;
		}
	}
}

/* ------------------------------------------------------- Path<Vertex> deque */

typedef struct {
	vtx* a;
	size_t cap, head, n;
} vpath;

static void
vpath_init(vpath* p)
{
	p->cap = 64;
	p->a = (vtx*)malloc(p->cap * sizeof(vtx));
	p->head = p->cap / 2;
	p->n = 0;
}
static void
vpath_free(vpath* p)
{
	free(p->a);
	p->a = NULL;
}
static void
vpath_regrow(vpath* p)
{
	size_t ncap = p->cap * 2 + 64;
	vtx* na = (vtx*)malloc(ncap * sizeof(vtx));
	size_t nhead = (ncap - p->n) / 2;
	if (!na) {
		fprintf(stderr, "oracle: out of memory\n");
		abort();
	}
	memcpy(na + nhead, p->a + p->head, p->n * sizeof(vtx));
	free(p->a);
	p->a = na;
	p->cap = ncap;
	p->head = nhead;
}
static void
vpath_push_back(vpath* p, const vtx* v)
{
	if (p->head + p->n == p->cap)
		vpath_regrow(p);
	p->a[p->head + p->n] = *v;
	p->n++;
}
static void
vpath_push_front(vpath* p, const vtx* v)
{
	if (p->head == 0)
		vpath_regrow(p);
	p->head--;
	p->a[p->head] = *v;
	p->n++;
}
static void
vpath_pop_back(vpath* p)
{
	p->n--;
}
static void
vpath_pop_front(vpath* p)
{
	p->head++;
	p->n--;
}
#define VP_AT(p, i) (&(p)->a[(p)->head + (i)])
#define VP_FRONT(p) VP_AT(p, 0)
#define VP_BACK(p) VP_AT(p, (p)->n - 1)

/* ------------------------------------------------------------ Bloom filters */

/* RollingHash::getHashes, RollingHash.h:141-146 */
static inline void
get_hashes(const orc_ctx* c, uint64_t h, uint64_t* out)
{
	out[0] = h;
	for (unsigned i = 1; i < c->nh; i++)
		out[i] = nte64(h, c->k, i);
}

/* CountingBloomFilter::minCount, CountingBloomFilter.hpp:53-64 */
static inline uint8_t
cbf_min_count(const orc_ctx* c, const uint64_t* hashes)
{
	uint8_t min = c->cnt[hashes[0] % c->m];
	for (unsigned i = 1; i < c->nh; i++) {
		uint64_t pos = hashes[i] % c->m;
		if (c->cnt[pos] < min)
			min = c->cnt[pos];
	}
	return min;
}
/* CountingBloomFilter::contains, :190-196 */
static inline int
cbf_contains(const orc_ctx* c, const uint64_t* hashes)
{
	return cbf_min_count(c, hashes) >= c->kc;
}
/* CountingBloomFilter::insert -> incrementMin, :135-162,198-204, executed by one
 * thread: every compare-and-swap sees the value it expects unless an earlier
 * iteration of this same loop already bumped that counter (two hashes of the
 * k-mer falling on one counter), so each minimal counter is incremented once. */
static inline void
cbf_insert(orc_ctx* c, const uint64_t* hashes)
{
	uint8_t min_val = cbf_min_count(c, hashes);
	uint8_t new_val = (uint8_t)(min_val + 1);
	if (min_val > new_val)
		return; /* saturated at 255 */
	for (unsigned i = 0; i < c->nh; i++) {
		uint64_t pos = hashes[i] % c->m;
		if (c->cnt[pos] == min_val)
			c->cnt[pos] = new_val;
	}
}
/* BloomFilter::insert, BloomFilter.hpp:182-191 */
static inline void
bf_insert(orc_ctx* c, const uint64_t* hashes)
{
	for (unsigned i = 0; i < c->nh; i++) {
		uint64_t p = hashes[i] % c->m;
		c->vis[p / 8] |= (uint8_t)(1u << (p % 8));
	}
}
/* BloomFilter::contains, BloomFilter.hpp:249-259 */
static inline int
bf_contains(const orc_ctx* c, const uint64_t* hashes)
{
	for (unsigned i = 0; i < c->nh; i++) {
		uint64_t p = hashes[i] % c->m;
		if (!(c->vis[p / 8] & (1u << (p % 8))))
			return 0;
	}
	return 1;
}

/* vertex_exists, RollingBloomDBG.h:436-446 */
static inline int
vertex_exists(const orc_ctx* c, const vtx* v)
{
	uint64_t hashes[ORC_MAX_HASHES];
	get_hashes(c, v->h, hashes);
	return cbf_contains(c, hashes);
}

/* ------------------------------------------------------ RollingHashIterator */

/* Callback returns non-zero to stop the iteration early. */
typedef int (*kmer_fn)(void* u, size_t pos, const vtx* v);

/* RollingHashIterator (RollingHashIterator.h:35-97,127-143): upper-case the
 * sequence, record non-ACGT positions, visit every k-mer with no non-ACGT char
 * under an unmasked position, computing hashes by reset() after a gap and by
 * rollRight() otherwise.  Returns the number of k-mers visited. */
static size_t
foreach_kmer(const orc_ctx* c, const char* seq_in, size_t len, kmer_fn fn, void* u)
{
	unsigned k = c->k;
	if (len < k)
		return 0;
	char* seq = (char*)malloc(len + 1);
	for (size_t i = 0; i < len; i++)
		seq[i] = (char)toupper((unsigned char)seq_in[i]);
	seq[len] = 0;
	/* next bad position at or after i: nb[i] (len if none) */
	size_t* nb = (size_t*)malloc((len + 1) * sizeof(size_t));
	nb[len] = len;
	for (size_t i = len; i-- > 0;) {
		char ch = seq[i];
		int good = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
		nb[i] = good ? nb[i + 1] : i;
	}
	size_t count = 0, pos = 0;
	int roll = 0;
	vtx v;
	while (pos < len - k + 1) {
		size_t bad = nb[pos];
		if (bad < pos + k) {
			if (!c->has_mask) {
				roll = 0;
				pos = bad + 1;
				continue;
			}
			int good_kmer = 1;
			for (size_t b = bad; b < pos + k; b = nb[b + 1]) {
				if (c->mask[b - pos] == '1') {
					good_kmer = 0;
					break;
				}
			}
			if (!good_kmer) {
				roll = 0;
				++pos;
				continue;
			}
		}
		if (!roll) {
			vtx_init(c, &v, seq + pos);
			roll = 1;
		} else {
			vtx_shift(c, &v, SENSE, seq[pos + k - 1]);
		}
		count++;
		if (fn(u, pos, &v))
			break;
		++pos;
	}
	free(nb);
	free(seq);
	return count;
}

/* ----------------------------------------------------------- graph search */

/* lookAhead, Graph/ExtendPath.h:100-139 */
static int
look_ahead_rec(const orc_ctx* c, const vtx* u, int dir, unsigned depth, unsigned limit,
    vset* visited)
{
	vset_insert(c, visited, u);
	if (depth >= limit)
		return 1;
	int sense = (dir == FORWARD) ? SENSE : ANTISENSE;
	for (int b = 0; b < 4; b++) {
		vtx v;
		vtx_neighbour(c, u, sense, BASE_CHARS[b], &v);
		if (!vertex_exists(c, &v))
			continue;
		if (!vset_contains(c, visited, &v)) {
			if (look_ahead_rec(c, &v, dir, depth + 1, limit, visited))
				return 1;
		}
	}
	return 0;
}
/* lookAhead wrapper, ExtendPath.h:153-161 */
static int
look_ahead(const orc_ctx* c, const vtx* start, int dir, unsigned depth)
{
	vset visited;
	vset_init(&visited);
	int r = look_ahead_rec(c, start, dir, 0, depth, &visited);
	vset_free(&visited);
	return r;
}

/* trueBranch, ExtendPath.h:174-244.  The edge is (u,v) walked in direction dir:
 * u is the vertex we come from, v the next vertex. */
static int
true_branch_rec(const orc_ctx* c, const vtx* u, const vtx* v, unsigned depth, int dir,
    unsigned trim, unsigned fp_trim, vset* visited)
{
	if (vset_contains(c, visited, v))
		return 1;
	if (depth >= trim)
		return 1;
	vset_insert(c, visited, v);

	int sense = (dir == FORWARD) ? SENSE : ANTISENSE;
	int other_dir = (dir == FORWARD) ? REVERSE : FORWARD;
	int other_sense = (dir == FORWARD) ? ANTISENSE : SENSE;

	for (int b = 0; b < 4; b++) {
		vtx w;
		vtx_neighbour(c, v, sense, BASE_CHARS[b], &w);
		if (!vertex_exists(c, &w))
			continue;
		if (true_branch_rec(c, v, &w, depth + 1, dir, trim, fp_trim, visited))
			return 1;
	}
	if (depth >= fp_trim || look_ahead(c, v, dir, fp_trim)) {
		for (int b = 0; b < 4; b++) {
			vtx w;
			vtx_neighbour(c, v, other_sense, BASE_CHARS[b], &w);
			if (!vertex_exists(c, &w))
				continue;
			if (vtx_eq(c, &w, u))
				continue;
			if (true_branch_rec(c, v, &w, 0, other_dir, trim, fp_trim, visited))
				return 1;
		}
	}
	vset_erase(c, visited, v);
	return 0;
}
/* trueBranch wrapper, ExtendPath.h:253-261 */
static int
true_branch(const orc_ctx* c, const vtx* u, const vtx* v, int dir, unsigned trim,
    unsigned fp_trim)
{
	vset visited;
	vset_init(&visited);
	int r = true_branch_rec(c, u, v, 0, dir, trim, fp_trim, &visited);
	vset_free(&visited);
	return r;
}

/* successor, ExtendPath.h:314-362 */
static int
successor(const orc_ctx* c, const vtx* u, int dir, unsigned trim, unsigned fp_trim, vtx* vout)
{
	int sense = (dir == FORWARD) ? SENSE : ANTISENSE;
	*vout = *u;
	for (unsigned i = 0;; i = (i == 0) ? 1 : (trim < 2 * i ? trim : 2 * i)) {
		unsigned true_branches = 0;
		for (int b = 0; b < 4; b++) {
			vtx w;
			vtx_neighbour(c, u, sense, BASE_CHARS[b], &w);
			if (!vertex_exists(c, &w))
				continue;
			if (true_branch(c, u, &w, dir, i, fp_trim)) {
				*vout = w;
				++true_branches;
				if (true_branches >= 2)
					break;
			}
		}
		if (true_branches == 0)
			return ORC_ER_DEAD_END;
		else if (true_branches == 1)
			return ORC_ER_LENGTH_LIMIT;
		else if (i == trim)
			return ORC_ER_AMBI_OUT;
	}
}

/* ambiguous(u, dir), ExtendPath.h:368-374 */
static int
ambiguous(const orc_ctx* c, const vtx* u, int dir, unsigned trim, unsigned fp_trim)
{
	vtx v;
	return successor(c, u, dir, trim, fp_trim, &v) == ORC_ER_AMBI_OUT;
}
/* ambiguous(u, expected, dir), ExtendPath.h:383-397 */
static int
ambiguous_expected(const orc_ctx* c, const vtx* u, const vtx* expected, int dir,
    unsigned trim, unsigned fp_trim)
{
	vtx v;
	int result = successor(c, u, dir, trim, fp_trim, &v);
	return result == ORC_ER_AMBI_OUT ||
	       (result == ORC_ER_LENGTH_LIMIT && !vtx_eq(c, &v, expected));
}

/* extendPathBySingleVertex, ExtendPath.h:403-459 */
static int
extend_path_by_single_vertex(const orc_ctx* c, vpath* path, int dir, unsigned trim,
    unsigned fp_trim, int look_behind)
{
	vtx t, v;
	int result;
	/* copy: the deque may reallocate below */
	vtx head = (dir == FORWARD) ? *VP_BACK(path) : *VP_FRONT(path);

	if (look_behind) {
		int other_dir = (dir == FORWARD) ? REVERSE : FORWARD;
		result = successor(c, &head, other_dir, trim, fp_trim, &t);
		if (result == ORC_ER_AMBI_OUT)
			return ORC_ER_AMBI_IN;
		if (path->n > 1) {
			if (result == ORC_ER_DEAD_END) {
				return ORC_ER_AMBI_IN;
			} else {
				const vtx* prev = (dir == FORWARD) ? VP_AT(path, path->n - 2) : VP_AT(path, 1);
				if (!vtx_eq(c, prev, &t))
					return ORC_ER_AMBI_IN;
			}
		}
	}
	result = successor(c, &head, dir, trim, fp_trim, &v);
	if (result != ORC_ER_LENGTH_LIMIT)
		return result;
	if (dir == FORWARD)
		vpath_push_back(path, &v);
	else
		vpath_push_front(path, &v);
	return ORC_ER_LENGTH_LIMIT;
}

/* ExtendPathParams as set by processRead, bloom-dbg.h:845-850: trimLen = trim,
 * fpTrim = 5, maxLen = NO_LIMIT, lookBehind = true, lookBehindStartVertex = false.
 * extendPath, ExtendPath.h:620-706 (both overloads). */
static int
extend_path(const orc_ctx* c, vpath* path, int dir, unsigned trim, unsigned fp_trim,
    unsigned* extension)
{
	vset visited;
	vset_init(&visited);
	for (size_t i = 0; i < path->n; i++)
		vset_insert(c, &visited, VP_AT(path, i));
	size_t orig = path->n;
	int result = ORC_ER_DEAD_END;
	int look_behind = 0; /* lookBehindStartVertex */
	for (;;) { /* maxLen == NO_LIMIT */
		result = extend_path_by_single_vertex(c, path, dir, trim, fp_trim, look_behind);
		if (result != ORC_ER_LENGTH_LIMIT)
			break;
		const vtx* head = (dir == FORWARD) ? VP_BACK(path) : VP_FRONT(path);
		if (!vset_insert(c, &visited, head)) {
			result = ORC_ER_CYCLE;
			if (dir == FORWARD)
				vpath_pop_back(path);
			else
				vpath_pop_front(path);
			break;
		}
		look_behind = 1; /* params.lookBehind */
	}
	vset_free(&visited);
	*extension = (unsigned)(path->n - orig);
	return result;
}

/* ------------------------------------------------------ sequence utilities */

static int
cb_all_in_counting(void* u, size_t pos, const vtx* v)
{
	(void)pos;
	const orc_ctx* c = (const orc_ctx*)((void**)u)[0];
	int* ok = (int*)((void**)u)[1];
	uint64_t hashes[ORC_MAX_HASHES];
	get_hashes(c, v->h, hashes);
	if (!cbf_contains(c, hashes)) {
		*ok = 0;
		return 1;
	}
	return 0;
}
static int
cb_all_in_visited(void* u, size_t pos, const vtx* v)
{
	(void)pos;
	const orc_ctx* c = (const orc_ctx*)((void**)u)[0];
	int* ok = (int*)((void**)u)[1];
	uint64_t hashes[ORC_MAX_HASHES];
	get_hashes(c, v->h, hashes);
	if (!bf_contains(c, hashes)) {
		*ok = 0;
		return 1;
	}
	return 0;
}
/* allKmersInBloom, bloom-dbg.h:58-77.  which: 0 = solid (counting), 1 = visited */
static int
all_kmers_in_bloom(const orc_ctx* c, const char* seq, size_t len, int which)
{
	int ok = 1;
	void* u[2] = { (void*)c, &ok };
	size_t valid = foreach_kmer(c, seq, len, which ? cb_all_in_visited : cb_all_in_counting, u);
	if (!ok)
		return 0;
	if (valid < len - c->k + 1)
		return 0;
	return 1;
}
static int
cb_add_to_visited(void* u, size_t pos, const vtx* v)
{
	(void)pos;
	orc_ctx* c = (orc_ctx*)u;
	uint64_t hashes[ORC_MAX_HASHES];
	get_hashes(c, v->h, hashes);
	bf_insert(c, hashes);
	return 0;
}
/* addKmersToBloom, bloom-dbg.h:79-90 */
static void
add_kmers_to_visited(orc_ctx* c, const char* seq, size_t len)
{
	foreach_kmer(c, seq, len, cb_add_to_visited, c);
}
static int
cb_coverage(void* u, size_t pos, const vtx* v)
{
	(void)pos;
	const orc_ctx* c = (const orc_ctx*)((void**)u)[0];
	unsigned* cov = (unsigned*)((void**)u)[1];
	uint64_t hashes[ORC_MAX_HASHES];
	get_hashes(c, v->h, hashes);
	*cov += cbf_min_count(c, hashes);
	return 0;
}
/* getSeqAbsoluteKmerCoverage, bloom-dbg.h:92-109 */
static unsigned
seq_coverage(const orc_ctx* c, const char* seq, size_t len)
{
	unsigned cov = 0;
	void* u[2] = { (void*)c, &cov };
	foreach_kmer(c, seq, len, cb_coverage, u);
	return cov;
}
static int
cb_to_path(void* u, size_t pos, const vtx* v)
{
	(void)pos;
	vpath_push_back((vpath*)u, v);
	return 0;
}
/* seqToPath, bloom-dbg.h:111-124 */
static void
seq_to_path(const orc_ctx* c, const char* seq, size_t len, vpath* path)
{
	foreach_kmer(c, seq, len, cb_to_path, path);
}
/* pathToSeq, bloom-dbg.h:126-158 (inconsistency warnings omitted: stderr only) */
static char*
path_to_seq(const orc_ctx* c, const vpath* path, size_t* len_out)
{
	unsigned k = c->k;
	size_t len = path->n + k - 1;
	char* seq = (char*)malloc(len + 1);
	memset(seq, 'N', len);
	seq[len] = 0;
	for (size_t i = 0; i < path->n; i++) {
		const char* kmer = VP_AT(path, i)->s;
		for (unsigned j = 0; j < k; j++)
			if (!c->has_mask || c->mask[j] == '1')
				seq[i + j] = kmer[j];
	}
	*len_out = len;
	return seq;
}
/* reverseComplement(Sequence), Common/Sequence.cpp:49-57 */
static void
revcomp_str(const char* s, size_t len, char* out)
{
	for (size_t i = 0; i < len; i++)
		out[i] = complement_base(s[len - 1 - i]);
	out[len] = 0;
}
/* canonicalize(Sequence&), Common/Sequence.h:39-44 */
static void
canonicalize_str(char* s, size_t len)
{
	char* rc = (char*)malloc(len + 1);
	revcomp_str(s, len, rc);
	if (strcmp(rc, s) < 0)
		memcpy(s, rc, len);
	free(rc);
}

/* ------------------------------------------------------------- pass 2 logic */

/* leftIsBluntEnd, bloom-dbg.h:489-509 */
static int
left_is_blunt_end(const orc_ctx* c, const char* seq, size_t len)
{
	if (len < c->k)
		return 0;
	vpath path;
	vpath_init(&path);
	seq_to_path(c, seq, c->k, &path);
	int r = !look_ahead(c, VP_FRONT(&path), REVERSE, 5);
	vpath_free(&path);
	return r;
}
/* hasBluntEnd, bloom-dbg.h:519-532 */
static int
has_blunt_end(const orc_ctx* c, const char* seq, size_t len)
{
	if (left_is_blunt_end(c, seq, len))
		return 1;
	char* rc = (char*)malloc(len + 1);
	revcomp_str(seq, len, rc);
	int r = left_is_blunt_end(c, rc, len);
	free(rc);
	return r;
}

/* isTip, bloom-dbg.h:759-776 */
static int
is_tip(unsigned length, int left, int right, unsigned trim)
{
	if (length > trim)
		return 0;
	if (left == ORC_ER_DEAD_END && (right == ORC_ER_DEAD_END || right == ORC_ER_AMBI_IN))
		return 1;
	if (right == ORC_ER_DEAD_END && (left == ORC_ER_DEAD_END || left == ORC_ER_AMBI_IN))
		return 1;
	return 0;
}

enum { CT_LINEAR, CT_CIRCULAR, CT_HAIRPIN };

/* edge(u, v, g), RollingBloomDBG.h:558-574 */
static int
edge_exists(const orc_ctx* c, const vtx* u, const vtx* v)
{
	for (int b = 0; b < 4; b++) {
		vtx w;
		vtx_neighbour(c, u, SENSE, BASE_CHARS[b], &w);
		if (!vertex_exists(c, &w))
			continue;
		if (vtx_eq(c, &w, v))
			return 1;
	}
	return 0;
}
/* getContigType, bloom-dbg.h:629-645 */
static int
get_contig_type(const orc_ctx* c, const vpath* path)
{
	if (edge_exists(c, VP_BACK(path), VP_FRONT(path))) {
		vtx v = *VP_FRONT(path);
		vtx_shift(c, &v, ANTISENSE, VP_BACK(path)->s[0]);
		if (kmer_eq(c, v.s, VP_BACK(path)->s))
			return CT_CIRCULAR;
		else
			return CT_HAIRPIN;
	}
	return CT_LINEAR;
}
/* preprocessCircularContig, bloom-dbg.h:648-702 */
static void
preprocess_circular_contig(const orc_ctx* c, vpath* path, unsigned trim)
{
	int type = get_contig_type(c, path);
	if (path->n <= 2)
		return;
	const unsigned fp_trim = 5;
	vtx front = *VP_FRONT(path), back = *VP_BACK(path);
	int branch_start = ambiguous(c, &front, FORWARD, trim, fp_trim) ||
	                   ambiguous(c, &front, REVERSE, trim, fp_trim);
	int branch_end = ambiguous(c, &back, FORWARD, trim, fp_trim) ||
	                 ambiguous(c, &back, REVERSE, trim, fp_trim);
	if (branch_start && !branch_end) {
		if (type == CT_CIRCULAR) {
			vpath_push_back(path, &front);
		} else {
			vtx rc = front;
			vtx_revcomp(c, &rc);
			vpath_push_back(path, &rc);
		}
	} else if (!branch_start && branch_end) {
		if (type == CT_CIRCULAR) {
			vpath_push_front(path, &back);
		} else {
			vtx rc = back;
			vtx_revcomp(c, &rc);
			vpath_push_front(path, &rc);
		}
	}
}
/* trimBranchKmers, bloom-dbg.h:723-757 */
static void
trim_branch_kmers(const orc_ctx* c, vpath* path, unsigned trim)
{
	if (path->n == 1)
		return;
	int type = get_contig_type(c, path);
	if (type == CT_CIRCULAR || type == CT_HAIRPIN)
		preprocess_circular_contig(c, path, trim);
	size_t l = path->n;
	const unsigned fp_trim = 5;
	vtx p0 = *VP_AT(path, 0), p1 = *VP_AT(path, 1);
	vtx pl1 = *VP_AT(path, l - 1), pl2 = *VP_AT(path, l - 2);
	int ambiguous1 = ambiguous_expected(c, &p0, &p1, FORWARD, trim, fp_trim);
	int ambiguous2 = ambiguous_expected(c, &pl1, &pl2, REVERSE, trim, fp_trim);
	if (ambiguous1)
		vpath_pop_front(path);
	if (ambiguous2)
		vpath_pop_back(path);
}

typedef struct {
	orc_contig_cb cb;
	void* user;
} sink;

/* outputContig, bloom-dbg.h:538-620 */
static void
output_contig(orc_ctx* c, const vpath* path, orc_contig* rec, const sink* out)
{
	const unsigned fp_look_ahead = 5;
	unsigned k = c->k;
	size_t len;
	char* seq = path_to_seq(c, path, &len);

	char kmer1[ORC_MAX_KMER + 1], kmer2[ORC_MAX_KMER + 1];
	memcpy(kmer1, seq, k);
	kmer1[k] = 0;
	canonicalize_str(kmer1, k);
	vtx v1;
	vtx_init(c, &v1, kmer1);
	memcpy(kmer2, seq + len - k, k);
	kmer2[k] = 0;
	canonicalize_str(kmer2, k);
	vtx v2;
	vtx_init(c, &v2, kmer2);

	int redundant = 0;
	if (len < k + fp_look_ahead - 1) {
		if (vset_contains(c, c->contig_end_kmers, &v1) &&
		    vset_contains(c, c->contig_end_kmers, &v2)) {
			redundant = 1;
		} else {
			vset_insert(c, c->contig_end_kmers, &v1);
			vset_insert(c, c->contig_end_kmers, &v2);
		}
	} else if (all_kmers_in_bloom(c, seq, len, 1)) {
		redundant = 1;
	}
	if (!redundant)
		add_kmers_to_visited(c, seq, len);
	rec->redundant = redundant;
	rec->seq = seq;
	rec->length = (uint32_t)len;
	rec->coverage = 0;
	rec->contig_id = UINT64_MAX;
	if (!redundant) {
		rec->coverage = seq_coverage(c, seq, len);
		rec->contig_id = c->contig_id;
		c->contig_id++;
		c->bases_assembled += len;
	}
	if (out->cb)
		out->cb(out->user, rec);
	free(seq);
}

/* processRead, bloom-dbg.h:781-882 */
static int
process_read(orc_ctx* c, const char* seq, size_t len, uint64_t read_index, const sink* out)
{
	unsigned k = c->k;
	if (len < k)
		return ORC_RR_SHORTER_THAN_K;
	/* allACGT, Common/Sequence.h:31-34 */
	for (size_t i = 0; i < len; i++) {
		char ch = seq[i];
		if (!(ch == 'a' || ch == 'c' || ch == 'g' || ch == 't' || ch == 'A' || ch == 'C' ||
		      ch == 'G' || ch == 'T'))
			return ORC_RR_NON_ACGT;
	}
	if (has_blunt_end(c, seq, len))
		return ORC_RR_BLUNT_END;
	if (!all_kmers_in_bloom(c, seq, len, 0))
		return ORC_RR_NOT_SOLID;
	c->solid_reads++;
	if (all_kmers_in_bloom(c, seq, len, 1)) {
		c->visited_reads++;
		return ORC_RR_ALL_KMERS_VISITED;
	}

	vset assembled; /* assembledKmers, bloom-dbg.h:837 */
	vset_init(&assembled);
	vpath path;
	vpath_init(&path);
	seq_to_path(c, seq, len, &path);
	for (size_t it = 0; it < path.n; it++) {
		const vtx* seed = VP_AT(&path, it);
		if (vset_contains(c, &assembled, seed))
			continue;
		orc_contig rec;
		memset(&rec, 0, sizeof rec);
		rec.read_index = read_index;
		rec.seed = seed->s;

		vpath contig;
		vpath_init(&contig);
		vpath_push_back(&contig, seed);
		const unsigned fp_trim = 5;
		rec.left_code = extend_path(c, &contig, REVERSE, c->trim, fp_trim, &rec.left_ext);
		rec.right_code = extend_path(c, &contig, FORWARD, c->trim, fp_trim, &rec.right_ext);

		if (!is_tip((unsigned)contig.n, rec.left_code, rec.right_code, c->trim)) {
			trim_branch_kmers(c, &contig, c->trim);
			output_contig(c, &contig, &rec, out);
		}
		for (size_t i = 0; i < contig.n; i++)
			vset_insert(c, &assembled, VP_AT(&contig, i));
		vpath_free(&contig);
	}
	vpath_free(&path);
	vset_free(&assembled);
	return ORC_RR_GENERATED_CONTIGS;
}

/* ----------------------------------------------------------------- public */

/* roundUpToMultiple(round(B / 1.125 / sizeof(uint8_t)), 64), bloom-dbg.cc:365-367,
 * BloomIO.h:14-24 */
uint64_t
orc_counters_for_budget(uint64_t bloom_bytes)
{
	double sz = (double)bloom_bytes / 1.125 / 1.0;
	uint64_t n = (uint64_t)round(sz);
	uint64_t rem = n % 64;
	if (rem)
		n += 64 - rem;
	return n;
}

orc_ctx*
orc_create(unsigned k, unsigned num_hashes, unsigned min_cov, unsigned trim, uint64_t counters,
    const char* mask)
{
	if (k < 2 || k > ORC_MAX_KMER || num_hashes < 1 || num_hashes > ORC_MAX_HASHES || !counters)
		return NULL;
	orc_ctx* c = (orc_ctx*)calloc(1, sizeof *c);
	c->k = k;
	c->nh = num_hashes;
	c->kc = min_cov;
	c->trim = trim;
	/* CountingBloomFilter ctor: round size in bytes up to a multiple of 8,
	 * CountingBloomFilter.hpp:40-50 */
	uint64_t rem = counters % 8;
	c->m = rem ? counters + 8 - rem : counters;
	c->cnt = (uint8_t*)calloc(c->m, 1);
	/* BloomFilter(filterSize = solid.size(), ...) -> initSize: bytes = size / 8
	 * (precondition: multiple of 64), BloomFilter.hpp:63-76 */
	c->vis = (uint8_t*)calloc(c->m / 8 + 1, 1);
	if (mask && mask[0]) {
		/* MaskedKmer::setMask checks, MaskedKmer.h:25-48 */
		if (strlen(mask) != k || strspn(mask, "01") != k || mask[0] != '1' ||
		    mask[k - 1] != '1') {
			free(c->cnt);
			free(c->vis);
			free(c);
			return NULL;
		}
		memcpy(c->mask, mask, k);
		c->has_mask = 1;
	}
	c->contig_end_kmers = (vset*)malloc(sizeof(vset));
	vset_init(c->contig_end_kmers);
	if (!c->cnt || !c->vis) {
		orc_destroy(c);
		return NULL;
	}
	return c;
}

void
orc_destroy(orc_ctx* c)
{
	if (!c)
		return;
	free(c->cnt);
	free(c->vis);
	if (c->contig_end_kmers) {
		vset_free(c->contig_end_kmers);
		free(c->contig_end_kmers);
	}
	free(c);
}

uint64_t orc_size(const orc_ctx* c) { return c->m; }
uint8_t* orc_counters(orc_ctx* c) { return c->cnt; }
uint8_t* orc_visited(orc_ctx* c) { return c->vis; }

uint64_t
orc_popcount(const orc_ctx* c)
{
	uint64_t n = 0;
	for (uint64_t i = 0; i < c->m; i++)
		n += c->cnt[i] != 0;
	return n;
}
uint64_t
orc_filtered_popcount(const orc_ctx* c)
{
	uint64_t n = 0;
	for (uint64_t i = 0; i < c->m; i++)
		n += c->cnt[i] >= c->kc;
	return n;
}

typedef struct {
	const orc_ctx* c;
	uint32_t* pos;
	uint64_t* hashes;
	uint64_t cap, n;
} hash_out;
static int
cb_hash_out(void* u, size_t pos, const vtx* v)
{
	hash_out* o = (hash_out*)u;
	if (o->n < o->cap) {
		if (o->pos)
			o->pos[o->n] = (uint32_t)pos;
		if (o->hashes)
			get_hashes(o->c, v->h, o->hashes + o->n * o->c->nh);
	}
	o->n++;
	return 0;
}
uint64_t
orc_hash_seq(const orc_ctx* c, const char* seq, size_t len, uint32_t* pos_out,
    uint64_t* hashes_out, uint64_t cap)
{
	hash_out o = { c, pos_out, hashes_out, cap, 0 };
	foreach_kmer(c, seq, len, cb_hash_out, &o);
	return o.n;
}

static int
cb_load(void* u, size_t pos, const vtx* v)
{
	(void)pos;
	orc_ctx* c = (orc_ctx*)u;
	uint64_t hashes[ORC_MAX_HASHES];
	get_hashes(c, v->h, hashes);
	cbf_insert(c, hashes);
	return 0;
}
/* loadSeq for each sequence, BloomIO.h:32-41 */
void
orc_load_seqs(orc_ctx* c, const char* seqs, const uint64_t* offsets, uint64_t n)
{
	for (uint64_t i = 0; i < n; i++)
		foreach_kmer(c, seqs + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), cb_load, c);
}

void
orc_insert_hashes(orc_ctx* c, const uint64_t* hashes, uint64_t n)
{
	for (uint64_t i = 0; i < n; i++)
		cbf_insert(c, hashes + i * c->nh);
}
void
orc_min_count(const orc_ctx* c, const uint64_t* hashes, uint64_t n, uint8_t* out)
{
	for (uint64_t i = 0; i < n; i++)
		out[i] = cbf_min_count(c, hashes + i * c->nh);
}
void
orc_visited_insert_hashes(orc_ctx* c, const uint64_t* hashes, uint64_t n)
{
	for (uint64_t i = 0; i < n; i++)
		bf_insert(c, hashes + i * c->nh);
}
void
orc_visited_contains(const orc_ctx* c, const uint64_t* hashes, uint64_t n, uint8_t* out)
{
	for (uint64_t i = 0; i < n; i++)
		out[i] = (uint8_t)bf_contains(c, hashes + i * c->nh);
}

typedef struct {
	orc_ctx* c;
	const char* seqs;
	const uint64_t* offsets;
	uint64_t n;
	uint8_t* results;
	sink out;
} assemble_args;

static void*
assemble_thread(void* p)
{
	assemble_args* a = (assemble_args*)p;
	for (uint64_t i = 0; i < a->n; i++) {
		/* the batch loop of assemble(), bloom-dbg.h:1012-1066, at -j1 */
		int r = process_read(a->c, a->seqs + a->offsets[i],
		    (size_t)(a->offsets[i + 1] - a->offsets[i]), a->c->reads_processed, &a->out);
		if (a->results)
			a->results[i] = (uint8_t)r;
		a->c->reads_processed++;
	}
	return NULL;
}

uint64_t
orc_assemble(orc_ctx* c, const char* seqs, const uint64_t* offsets, uint64_t n,
    uint8_t* results_out, orc_contig_cb cb, void* user)
{
	/* trueBranch recursion is deep (bin/abyss-pe:17-25 runs the reference under
	 * abyss-stack-size 65536): run on a thread with a large stack. */
	assemble_args a = { c, seqs, offsets, n, results_out, { cb, user } };
	pthread_attr_t attr;
	pthread_t th;
	pthread_attr_init(&attr);
	pthread_attr_setstacksize(&attr, (size_t)1 << 30);
	if (pthread_create(&th, &attr, assemble_thread, &a) != 0) {
		assemble_thread(&a);
	} else {
		pthread_join(th, NULL);
	}
	pthread_attr_destroy(&attr);
	return c->contig_id;
}

void
orc_counters_get(const orc_ctx* c, uint64_t* solid_reads, uint64_t* visited_reads,
    uint64_t* reads_processed, uint64_t* bases_assembled, uint64_t* next_contig_id)
{
	if (solid_reads) *solid_reads = c->solid_reads;
	if (visited_reads) *visited_reads = c->visited_reads;
	if (reads_processed) *reads_processed = c->reads_processed;
	if (bases_assembled) *bases_assembled = c->bases_assembled;
	if (next_contig_id) *next_contig_id = c->contig_id;
}

int
orc_look_ahead(const orc_ctx* c, const char* kmer, int dir, unsigned depth)
{
	vtx v;
	vtx_init(c, &v, kmer);
	return look_ahead(c, &v, dir, depth);
}
int
orc_successor(const orc_ctx* c, const char* kmer, int dir, unsigned trim, unsigned fp_trim,
    char* succ_out)
{
	vtx u, v;
	vtx_init(c, &u, kmer);
	int r = successor(c, &u, dir, trim, fp_trim, &v);
	if (succ_out) {
		memcpy(succ_out, v.s, c->k);
		succ_out[c->k] = 0;
	}
	return r;
}
unsigned
orc_out_mask(const orc_ctx* c, const char* kmer)
{
	vtx u, w;
	unsigned m = 0;
	vtx_init(c, &u, kmer);
	for (int b = 0; b < 4; b++) {
		vtx_neighbour(c, &u, SENSE, BASE_CHARS[b], &w);
		if (vertex_exists(c, &w))
			m |= 1u << b;
	}
	return m;
}
unsigned
orc_in_mask(const orc_ctx* c, const char* kmer)
{
	vtx u, w;
	unsigned m = 0;
	vtx_init(c, &u, kmer);
	for (int b = 0; b < 4; b++) {
		vtx_neighbour(c, &u, ANTISENSE, BASE_CHARS[b], &w);
		if (vertex_exists(c, &w))
			m |= 1u << b;
	}
	return m;
}

/* ------------------------------------------------------------ -g: outputGraph */

/* trimSeq (bloom-dbg.h:399-451): the longest run of consecutive k-mers contained in the solid
 * filter; the first such run wins a tie (strictly greater).  RollingHashIterator skips k-mers
 * over non-ACGT characters, and a jump in position ends a run (:422). */
typedef struct {
	const orc_ctx* c;
	size_t prev, start, len, best_start, best_len;
} trim_state;
#define TRIM_UNSET ((size_t)-1)
static int
cb_trim(void* u, size_t pos, const vtx* v)
{
	trim_state* t = (trim_state*)u;
	int good = vertex_exists(t->c, v);
	if (!good || (t->prev != TRIM_UNSET && pos - t->prev > 1)) {
		if (t->start != TRIM_UNSET && t->len > t->best_len) {
			t->best_len = t->len;
			t->best_start = t->start;
		}
		t->start = TRIM_UNSET;
		t->len = 0;
	}
	if (good) {
		if (t->start == TRIM_UNSET)
			t->start = pos;
		t->len++;
	}
	t->prev = pos;
	return 0;
}

typedef struct {
	char* buf;
	size_t n, cap;
	orc_text_cb cb;
	void* user;
} text_out;
static void
text_flush(text_out* o)
{
	if (o->n && o->cb)
		o->cb(o->user, o->buf, o->n);
	o->n = 0;
}
static void
text_put(text_out* o, const char* s, size_t n)
{
	if (o->n + n > o->cap)
		text_flush(o);
	memcpy(o->buf + o->n, s, n);
	o->n += n;
}

/* breadthFirstSearchImpl (Graph/BreadthFirstSearch.h:93-167) with one start vertex, a colour map
 * shared by all searches (DefaultColorMap: white unless present) and GraphvizBFSVisitor
 * (bloom-dbg.h:1097-1159): examine_edge prints "\tU -> V;\n" for every out-edge of a dequeued
 * vertex, discover_vertex prints "\tV;\n" when the target was white.  A search runs until its
 * queue is empty, so every vertex it discovered is black afterwards: "non-white" is all the
 * colour map has to remember, and a start vertex that is not white is not even enqueued. */
static void
graph_bfs(const orc_ctx* c, vset* seen, const vtx* start, text_out* o, uint64_t* nodes, uint64_t* edges)
{
	unsigned k = c->k;
	if (vset_contains(c, seen, start))
		return; /* black (:126-128) */
	size_t qcap = 1024, qh = 0, qn = 0;
	vtx* q = (vtx*)malloc(qcap * sizeof(vtx));
	vset_insert(c, seen, start);
	(*nodes)++;
	text_put(o, "\t", 1); text_put(o, start->s, k); text_put(o, ";\n", 2);
	q[qn++] = *start;
	while (qh < qn) {
		vtx u = q[qh++];
		for (int b = 0; b < 4; b++) { /* out_edge_iterator, RollingBloomDBG.h:299-330 */
			vtx v;
			vtx_neighbour(c, &u, SENSE, BASE_CHARS[b], &v);
			if (!vertex_exists(c, &v))
				continue;
			(*edges)++;
			text_put(o, "\t", 1); text_put(o, u.s, k); text_put(o, " -> ", 4); text_put(o, v.s, k); text_put(o, ";\n", 2);
			if (vset_insert(c, seen, &v)) {
				(*nodes)++;
				text_put(o, "\t", 1); text_put(o, v.s, k); text_put(o, ";\n", 2);
				if (qn == qcap) {
					if (qh > qcap / 2) {
						memmove(q, q + qh, (qn - qh) * sizeof(vtx));
						qn -= qh; qh = 0;
					} else {
						qcap *= 2;
						q = (vtx*)realloc(q, qcap * sizeof(vtx));
					}
				}
				q[qn++] = v;
			}
		}
	}
	free(q);
}

/* outputGraph (bloom-dbg.h:1171-1242), without the "digraph g {" / "}" frame the visitor's
 * constructor and destructor print.  Sequences as FastaReader(FOLD_CASE) hands them over. */
void
orc_output_graph(const orc_ctx* c, const char* seqs, const uint64_t* offsets, uint64_t n,
    orc_text_cb cb, void* user, uint64_t* nodes_out, uint64_t* edges_out)
{
	unsigned k = c->k;
	vset seen;
	vset_init(&seen);
	text_out o = { (char*)malloc(1 << 20), 0, 1 << 20, cb, user };
	uint64_t nodes = 0, edges = 0;
	for (uint64_t i = 0; i < n; i++) {
		const char* seq = seqs + offsets[i];
		size_t len = (size_t)(offsets[i + 1] - offsets[i]);
		if (len < k)
			continue; /* trimSeq :406-409 */
		trim_state t = { c, TRIM_UNSET, TRIM_UNSET, 0, TRIM_UNSET, 0 };
		foreach_kmer(c, seq, len, cb_trim, &t);
		if (t.start != TRIM_UNSET && t.len > t.best_len) { /* :439-442 */
			t.best_len = t.len;
			t.best_start = t.start;
		}
		if (t.best_len == 0)
			continue;
		/* seq = seq.substr(maxMatchStart, maxMatchLen + k - 1); FastaReader folded the case */
		char first[ORC_MAX_KMER + 1], last[ORC_MAX_KMER + 1];
		for (unsigned j = 0; j < k; j++) {
			first[j] = (char)toupper((unsigned char)seq[t.best_start + j]);
			last[j] = (char)toupper((unsigned char)seq[t.best_start + t.best_len - 1 + j]);
		}
		first[k] = last[k] = 0;
		vtx start, rc;
		vtx_init(c, &start, first);
		graph_bfs(c, &seen, &start, &o, &nodes, &edges);
		/* reverseComplement(seq).substr(0, k) = reverse complement of the last k-mer, hashed afresh (:1223-1225) */
		char rcs[ORC_MAX_KMER + 1];
		for (unsigned j = 0; j < k; j++)
			rcs[j] = complement_base(last[k - 1 - j]);
		rcs[k] = 0;
		vtx_init(c, &rc, rcs);
		graph_bfs(c, &seen, &rc, &o, &nodes, &edges);
	}
	text_flush(&o);
	free(o.buf);
	vset_free(&seen);
	if (nodes_out) *nodes_out = nodes;
	if (edges_out) *edges_out = edges;
}

/* ---------------------------------------------------- HashAgnosticCascadingBloom */
struct orc_cascade {
	orc_ctx* hasher;   /* k, H and the k-mer iterator (its own filters are unused) */
	unsigned levels;
	uint64_t bits;
	uint8_t** data;
};
orc_cascade*
orc_cascade_create(unsigned k, unsigned num_hashes, unsigned levels, uint64_t level_bits)
{
	if (!levels || !level_bits || level_bits % 64) return NULL;
	orc_cascade* c = (orc_cascade*)calloc(1, sizeof *c);
	c->hasher = orc_create(k, num_hashes, 0, k, 64, NULL);
	if (!c->hasher) { free(c); return NULL; }
	c->hasher->m = level_bits; /* hash positions are taken modulo the level size */
	c->levels = levels;
	c->bits = level_bits;
	c->data = (uint8_t**)calloc(levels, sizeof(uint8_t*));
	for (unsigned i = 0; i < levels; i++) c->data[i] = (uint8_t*)calloc(level_bits / 8, 1);
	return c;
}
void
orc_cascade_destroy(orc_cascade* c)
{
	if (!c) return;
	for (unsigned i = 0; i < c->levels; i++) free(c->data[i]);
	free(c->data);
	c->hasher->m = 64;
	orc_destroy(c->hasher);
	free(c);
}
static int
cb_cascade_insert(void* u, size_t pos, const vtx* v)
{
	(void)pos;
	orc_cascade* c = (orc_cascade*)u;
	uint64_t hashes[ORC_MAX_HASHES];
	get_hashes(c->hasher, v->h, hashes);
	/* HashAgnosticCascadingBloom::insert, HashAgnosticCascadingBloom.h:124-133 */
	for (unsigned l = 0; l < c->levels; l++) {
		int contains = 1; /* BloomFilter::contains, BloomFilter.hpp:249-259 */
		for (unsigned i = 0; i < c->hasher->nh; i++) {
			uint64_t p = hashes[i] % c->bits;
			if (!(c->data[l][p / 8] & (1u << (p % 8)))) { contains = 0; break; }
		}
		if (!contains) {
			for (unsigned i = 0; i < c->hasher->nh; i++) { /* BloomFilter::insert, :182-191 */
				uint64_t p = hashes[i] % c->bits;
				c->data[l][p / 8] |= (uint8_t)(1u << (p % 8));
			}
			break;
		}
	}
	return 0;
}
void
orc_cascade_load_seqs(orc_cascade* c, const char* seqs, const uint64_t* offsets, uint64_t n)
{
	for (uint64_t i = 0; i < n; i++)
		foreach_kmer(c->hasher, seqs + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), cb_cascade_insert, c);
}
uint8_t*
orc_cascade_level(orc_cascade* c, unsigned level)
{
	return level < c->levels ? c->data[level] : NULL;
}

/* SpacedSeed::kmerPair, SpacedSeed.h:18-25 */
void
orc_seed_kmer_pair(unsigned k, unsigned K, char* out)
{
	memset(out, '0', k);
	memset(out, '1', K);
	memset(out + k - K, '1', K);
	out[k] = 0;
}
/* SpacedSeed::qrSeed, SpacedSeed.h:39-52 */
void
orc_seed_qr(unsigned len, char* out)
{
	memset(out, '1', len);
	out[len] = 0;
	for (size_t i = 0; i < len; ++i) {
		for (size_t j = 1; j < len; ++j) {
			if (j * j % len == i) {
				out[i] = '0';
				break;
			}
		}
	}
}
/* SpacedSeed::qrSeedPair, SpacedSeed.h:65-75 */
void
orc_seed_qr_pair(unsigned k, unsigned qr_len, char* out)
{
	char qr[ORC_MAX_KMER + 1];
	memset(out, '0', k);
	out[k] = 0;
	orc_seed_qr(qr_len, qr);
	memcpy(out, qr, qr_len);
	/* std::reverse(qr); std::copy(qr.rbegin(), qr.rend(), seed.rbegin()):
	 * the tail of the seed read backwards equals the reversed QR read backwards,
	 * i.e. seed[k-1-i] = qr[i] */
	for (unsigned i = 0; i < qr_len; i++)
		out[k - 1 - i] = qr[i];
}
