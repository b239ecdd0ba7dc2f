// Simulation: which ops of a batch could be settled without the reservation rounds under the extended rule,
// and a check that settling them by formula gives the sequential result.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint64_t rs = 88172645463325252ULL;
static uint64_t rnd(void){ rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static uint64_t mix(uint64_t x){ x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
#define K 31
#define H 4
typedef struct { uint64_t h; uint32_t t; } Op;
static int cmp_op(const void* a, const void* b){ const Op* x=a; const Op* y=b; if (x->h != y->h) return x->h < y->h ? -1 : 1; return x->t < y->t ? -1 : x->t > y->t; }
typedef struct { uint64_t pos; uint32_t q; uint8_t j; } Pr;
static int cmp_pr(const void* a, const void* b){ const Pr* x=a; const Pr* y=b; if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1; return x->q < y->q ? -1 : x->q > y->q; }
static uint64_t M;
static uint64_t pos_of(uint64_t h, int j){ return mix(h ^ ((uint64_t)j * 0xD6E8FEB86659FD93ULL)) % M; }
static void seq_op(uint8_t* c, uint64_t h){ uint64_t p[H]; unsigned mn = 255; for (int j=0;j<H;j++){ p[j]=pos_of(h,j); if (c[p[j]]<mn) mn=c[p[j]]; } if (mn==255) return; for (int j=0;j<H;j++) if (c[p[j]]==mn) c[p[j]]=mn+1; /* duplicates: second sees mn+1 */ }
int main(int argc, char** argv)
{
	uint64_t G = argc > 1 ? atoll(argv[1]) : 400000; int cov = argc > 5 ? atoi(argv[5]) : 50; int RL = 150; double err = 0.005; int NB = argc > 2 ? atoi(argv[2]) : 59;
	double bpb = argc > 3 ? atof(argv[3]) : 71.6; int lead_js = 2; int mode = argc > 4 ? atoi(argv[4]) : 1;
	M = (uint64_t)(G * bpb);
	uint8_t* g = malloc(G); for (uint64_t i=0;i<G;i++) g[i]=rnd()&3;
	uint64_t nreads = G * cov / RL;
	uint64_t per = RL - K + 1, T = nreads * per;
	uint64_t* hs = malloc(T * 8);
	uint8_t rd[512];
	for (uint64_t r=0;r<nreads;r++){
		uint64_t s = rnd() % (G - RL); int rc = rnd()&1;
		for (int i=0;i<RL;i++){ uint8_t b = g[s+i]; if ((rnd() % 100000) < err*100000) b = (b + 1 + rnd()%3)&3; rd[i]=b; }
		if (rc) { uint8_t t2[512]; for (int i=0;i<RL;i++) t2[i]=3-rd[RL-1-i]; memcpy(rd,t2,RL); }
		for (uint64_t i=0;i<per;i++){
			uint64_t f=0, v=0; for (int x=0;x<K;x++){ f=(f<<2)|rd[i+x]; v=(v<<2)|(3-rd[i+K-1-x]); }
			hs[r*per+i] = mix(f<v?f:v);
		}
	}
	uint8_t* cs = calloc(M,1); // sequential truth
	uint8_t* cm = calloc(M,1); // mixed evaluation
	uint64_t bsz = (T + NB - 1)/NB;
	Op* ops = malloc(bsz*sizeof(Op)); Pr* prs = malloc(bsz*H*sizeof(Pr));
	uint64_t tot=0, pend_old=0, pend_new=0, iters_max=0, pend_rule=0;
	for (int b=0;b<NB;b++){
		uint64_t t0=b*bsz, t1=t0+bsz<T?t0+bsz:T, n=t1-t0;
		for (uint64_t i=0;i<n;i++){ ops[i].h=hs[t0+i]; ops[i].t=i; }
		qsort(ops,n,sizeof(Op),cmp_op);
		// distinct k-mers
		uint32_t nq=0; uint32_t* qn=malloc(n*4); uint64_t* qh=malloc(n*8); uint32_t* opq=malloc(n*4);
		for (uint64_t i=0;i<n;i++){ if (!i || ops[i].h!=ops[i-1].h){ qh[nq]=ops[i].h; qn[nq]=0; nq++; } qn[nq-1]++; opq[ops[i].t]=nq-1; }
		uint64_t np=0; for (uint32_t q=0;q<nq;q++) for (int j=0;j<H;j++){ prs[np].pos=pos_of(qh[q],j); prs[np].q=q; prs[np].j=j; np++; }
		qsort(prs,np,sizeof(Pr),cmp_pr);
		uint8_t* shared = calloc(nq,1); // bit j
		for (uint64_t i=0;i<np;){ uint64_t e=i; while (e<np && prs[e].pos==prs[i].pos) e++; if (e-i>1) for (uint64_t x=i;x<e;x++) shared[prs[x].q] |= 1u<<prs[x].j; i=e; }
		// per k-mer verdicts
		uint8_t* st = calloc(nq,1); // 0 rounds, 1 settled-old(pure/benign), 2 candidate new
		uint8_t* tg = calloc(nq,1); uint8_t* wmask = calloc(nq,1);
		for (uint32_t q=0;q<nq;q++){
			unsigned fl=shared[q], mp=256, ms=256; int known = 0;
			for (int j=0;j<H;j++){ unsigned c=cm[pos_of(qh[q],j)]; if ((fl>>j)&1){ if(c<ms)ms=c; } else { if(c<mp)mp=c; if (j<lead_js) known=1; } }
			if (!fl){ st[q]=1; tg[q]= mp+qn[q]>255?255:mp+qn[q]; continue; }
			if (!known || qn[q]>=254) { st[q]=0; continue; }
			unsigned t = mp+qn[q]>255?255:mp+qn[q]; tg[q]=t;
			if (ms>=t) { st[q]=1; continue; }
			if (ms>=mp && mode) { st[q]=2; for (int j=0;j<H;j++) if (((fl>>j)&1) && cm[pos_of(qh[q],j)]<t) wmask[q]|=1u<<j; pend_rule+=qn[q]; }
			else st[q]=0;
		}
		for (uint32_t q=0;q<nq;q++) if (st[q]!=1) pend_old+=qn[q];
		// fixpoint
		uint64_t it=0; int changed=1;
		while (changed){ changed=0; it++;
			for (uint64_t i=0;i<np;){ uint64_t e=i; while (e<np && prs[e].pos==prs[i].pos) e++;
				if (e-i>1){ int bad=0; for (uint64_t x=i;x<e;x++) if (st[prs[x].q]==0) bad=1;
					if (bad) for (uint64_t x=i;x<e;x++){ uint32_t q=prs[x].q; if (st[q]==2 && ((wmask[q]>>prs[x].j)&1)) { st[q]=0; changed=1; } } }
				i=e; }
		}
		if (it>iters_max) iters_max=it;
		for (uint32_t q=0;q<nq;q++) if (st[q]==0) pend_new+=qn[q];
		tot+=n;
		// mixed evaluation: formula for settled (max), then rounds ops in order
		for (uint32_t q=0;q<nq;q++) if (st[q]) { if (tg[q]) for (int j=0;j<H;j++){ uint64_t p=pos_of(qh[q],j); /* raise only counters below tg... */ } }
		// need initial values for all formula decisions: compute raises into a temp then apply
		{
			// raises: for settled k-mer, every counter of it: c = max(c, tg) BUT only if mp<255 (tg computed) -- per formula all counters raised to at least tg? no: only counters < tg are raised to tg
			Pr* rz = malloc(np*sizeof(Pr)); uint64_t nr=0;
			for (uint32_t q=0;q<nq;q++) if (st[q]) for (int j=0;j<H;j++){ rz[nr].pos=pos_of(qh[q],j); rz[nr].q=tg[q]; nr++; }
			for (uint64_t i=0;i<nr;i++) if (cm[rz[i].pos] < rz[i].q) cm[rz[i].pos]=(uint8_t)rz[i].q;
			free(rz);
		}
		for (uint64_t i=0;i<n;i++) if (st[opq[i]]==0) seq_op(cm, hs[t0+i]);
		for (uint64_t i=0;i<n;i++) seq_op(cs, hs[t0+i]);
		if (memcmp(cs,cm,M)) { uint64_t d=0; for (uint64_t i=0;i<M;i++) d+=cs[i]!=cm[i]; printf("batch %d: MISMATCH in %llu counters\n", b, (unsigned long long)d); return 1; }
		free(qn);free(qh);free(opq);free(shared);free(st);free(tg);free(wmask);
	}
	printf("ops %llu: pending old rule %.2f%%, candidates by the new rule %.2f%%, pending after closure %.2f%%, max fixpoint iterations %llu; counters equal to sequential\n",
	    (unsigned long long)tot, 100.0*pend_old/tot, 100.0*pend_rule/tot, 100.0*pend_new/tot, (unsigned long long)iters_max);
	return 0;
}
