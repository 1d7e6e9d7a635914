// shim_fill.h -- the marshalling half of the reference-side binding: the reference's objects (Seq, PwdB,
// Exinon, the parameter globals) -> the plain structs of include/spdp.h.  This is the code a maintainer of ogotoh/spaln
// adds to the reference's tree (it compiles against the reference's headers, src/aln.h & co., and include/spdp.h; nothing
// of the reference is in it).  Used by spaln_gpu_shim.cc (the reference's CLI on the library), and by the checkers
// oracle/ref_build/shim_check.cc (one pair, reference vs library) and seed_bench.cc (a batch of pairs through the seeded path).
#ifndef SHIM_FILL_H_
#define SHIM_FILL_H_
#include <vector>
#include <string>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "aln.h"		// the reference's headers (src/)
#include "utilseq.h"
#include "wln.h"
#include "vmf.h"
#include "gsinfo.h"
#include "spdp.h"
extern	int	MaxVmfSpace;

static void fill_scoring(SpdpScoring& sc, const PwdB* pwd, const Seq* b) {
	memset(&sc, 0, sizeof sc);
const	Simmtx* sm = pwd->simmtx;			// src/simmtx.h:35-63
	sc.mtx_dim = sm->dim;
	for (int i = 0; i < sm->dim; ++i)
	    for (int j = 0; j < sm->dim; ++j) sc.mtx[i * sm->dim + j] = sm->mtx[i][j];
	sc.gop = pwd->BasicGOP;  sc.gep = pwd->BasicGEP;	// src/aln.h:243-244
	sc.lgop = pwd->LongGOP;  sc.lgep = pwd->LongGEP;  sc.noll = pwd->Noll;  sc.codonk1 = pwd->codonk1;
	sc.spj = b->inex.intr;
	sc.llmt = IntronPrm.llmt;			// src/codepot.h:207-214
	sc.ipen = pwd->IntPen->Penalty();		// GapWI, src/codepot.h:241
	sc.nquant = IntronPrm.nquant;			// 1 when -A3 (src/fwd2s1.cc:125)
	for (int j = 0; pwd->IntPen->qm && j < sc.nquant; ++j) {	// (the quantised table exists under -A2 / -A3 only, src/codepot.cc:161)
	    sc.qm_len[j] = pwd->IntPen->qm[j].len;	// src/codepot.h:218-221,232
	    sc.qm_pen[j] = pwd->IntPen->qm[j].pen;
	}
	sc.local = algmode.lcl & 16;  sc.sh = alprm.sh;  sc.ubh = alprm.ubh;
	sc.max_vmf_space = MaxVmfSpace;  sc.ref_nelem = 16;
}

static void fill_problem(SpdpProblem& p, const Seq* a, const Seq* b,
			 std::vector<int16_t>& s5, std::vector<int16_t>& s3) {
	memset(&p, 0, sizeof p);
	p.a = a->at(0);  p.a_len = a->len;		// residue codes, src/seq.h:329
	p.b = b->at(0);  p.b_len = b->len;
	s5.assign(b->len + 1, 0);  s3.assign(b->len + 1, 0);
	for (int n = b->left; n <= b->right; ++n) {	// Exinon::data_n, src/codepot.h:104
const	    SGPT2* g = b->exin->score_n(n);
	    s5[n] = g->sig5;  s3[n] = g->sig3;
	}
	p.sig5 = s5.data();  p.sig3 = s3.data();
	p.a_left = a->left;  p.a_right = a->right;  p.b_left = b->left;  p.b_right = b->right;
	p.a_exgl = a->inex.exgl;  p.a_exgr = a->inex.exgr;
	p.b_exgl = b->inex.exgl;  p.b_exgr = b->inex.exgr;
}

// the exact-model inputs (-A0 / -A1 engines, and the seeded walk's joins): IntronPenalty::Penalty(len) materialised, the
// junction table behind Exinon::sig53(.., IE53), dinucleotide classes, site flags, splice-phase marks
struct SeedCols { std::vector<int16_t> s5, s3, ip; std::vector<uint8_t> c5, c3, dc; std::vector<int8_t> p5, p3; std::vector<int32_t> flat; };
// (with_table = false: a batch shares one IntPen table, its caller fills sc.intpen once)
static void fill_exact_s(SpdpScoring& sc, SpdpProblem& p, const Seq* b, const PwdB* pwd, SeedCols& c, bool with_table = true) {
	if (with_table) {
	    c.ip.resize(b->len + 2);
	    for (int l = 0; l < (int) c.ip.size(); ++l) c.ip[l] = pwd->IntPen->Penalty(l);
	    sc.intpen = c.ip.data();  sc.intpen_len = (int) c.ip.size();
	}
	sc.minl = IntronPrm.minl;
	sc.scalar_engines = algmode.alg == 0? 1: (algmode.alg == 1? 2: 0);
	c.c5.assign(b->len + 3, 0); c.c3.assign(b->len + 3, 0); c.dc.assign(b->len + 3, 0);
	c.p5.assign(b->len + 3, -2); c.p3.assign(b->len + 3, -2);
	std::vector<unsigned char> d5(b->len + 3, 0), d3(b->len + 3, 0);
	int	nc = 1;
	for (int i = b->left; i < b->right; ++i) {		// the dinucleotide classes of Exinon::intron53_c
	    int ch = ncredctab[*b->at(i)];
	    if (ch >= 4) ch = 1;
	    nc = ((nc << 2) + ch) & 0xf;
	    if (i - 1 >= 0) d5[i - 1] = nc;
	    d3[i + 1] = nc;
	}
	int	mrep[16], nrep[16];
	for (int u = 0; u < 16; ++u) mrep[u] = nrep[u] = -1;
	for (int n = b->left; n <= b->right; ++n) {
const	    SGPT2* g = b->exin->score_n(n);
	    c.p5[n] = g->phs5; c.p3[n] = g->phs3;
	    c.c5[n] = b->exin->isDonor(n); c.c3[n] = b->exin->isAccpt(n);
	    c.dc[n] = (uint8_t) (d5[n] << 4 | d3[n]);
	    if (n < b->right - 1 && mrep[d5[n]] < 0) mrep[d5[n]] = n;
	    if (n >= b->left + 2 && nrep[d3[n]] < 0) nrep[d3[n]] = n;
	}
	for (int u = 0; u < 16; ++u)
	    for (int v = 0; v < 16; ++v)
		sc.t53[16 * u + v] = (mrep[u] >= 0 && nrep[v] >= 0)?
		    (int16_t) (b->exin->sig53(mrep[u], nrep[v], IE53) - b->exin->score_n(nrep[v])->sig3): 0;
	p.cano5 = c.c5.data(); p.cano3 = c.c3.data(); p.dinc = c.dc.data(); p.phs5 = c.p5.data(); p.phs3 = c.p3.data();
	p.exin_left = b->left; p.exin_right = b->right;
}

static void fill_scoring_h(SpdpScoringH& sc, const PwdB* pwd, const Seq* b) {
	memset(&sc, 0, sizeof sc);
const	Simmtx* sm = pwd->simmtx;			// aa x tron matrix: rows x dim
	sc.mtx_rows = sm->rows;  sc.mtx_cols = sm->dim;
	for (int i = 0; i < sm->rows; ++i)
	    for (int j = 0; j < sm->dim; ++j) sc.mtx[i * sm->dim + j] = sm->mtx[i][j];
	sc.gop = pwd->BasicGOP;  sc.gep = pwd->BasicGEP;  sc.lgep = pwd->LongGEP;
	sc.codonk1 = pwd->codonk1;			// GapExtPen3, src/aln.h:302
	sc.noll = pwd->Noll;  sc.lgop = pwd->LongGOP;	// double affine gaps (-yl3): GapW3L = lgop + lgep
	sc.gapw1 = pwd->GapW1;  sc.gapw2 = pwd->GapW2;  sc.gapw3 = pwd->GapW3;
	sc.spj = b->inex.intr;  sc.llmt = IntronPrm.llmt;  sc.ipen = pwd->IntPen->Penalty();
	sc.nquant = IntronPrm.nquant;			// 1 under -A3 (src/fwd2h1.cc:127)
	for (int j = 0; pwd->IntPen->qm && j < sc.nquant; ++j) { sc.qm_len[j] = pwd->IntPen->qm[j].len; sc.qm_pen[j] = pwd->IntPen->qm[j].pen; }
	sc.local = algmode.lcl & 16;  sc.term_codon = (algmode.lcl & 2) != 0;
	sc.sh = alprm.sh;  sc.max_vmf_space = MaxVmfSpace;  sc.ubh = alprm.ubh;  sc.ref_nelem = 16;
}

struct HCols { std::vector<int16_t> s5, s3, sS, sT, sE; std::vector<int8_t> p5, p3; };
static void fill_problem_h(SpdpProblemH& p, const Seq* a, const Seq* b, HCols& c, int exin_left, int exin_right) {
	memset(&p, 0, sizeof p);
	p.a = a->at(0);  p.a_len = a->len;		// amino-acid codes
	p.b = b->at(0);  p.b_len = b->len;		// tron codes; at(len) is readable (terminator)
const	int N = b->len + 3;
	c.s5.assign(N, 0); c.s3.assign(N, 0); c.sS.assign(N, 0); c.sT.assign(N, 0); c.sE.assign(N, 0);
	c.p5.assign(N, -2); c.p3.assign(N, -2);
	for (int n = std::max(0, exin_left - 1); n <= exin_right + 1; ++n) {	// what Exinon allocated
const	    SGPT6* g = b->exin->score_p(n);		// src/codepot.h:105
	    c.s5[n] = g->sig5; c.s3[n] = g->sig3; c.sS[n] = g->sigS; c.sT[n] = g->sigT; c.sE[n] = g->sigE;
	    c.p5[n] = g->phs5; c.p3[n] = g->phs3;
	}
	p.sig5 = c.s5.data(); p.sig3 = c.s3.data(); p.sigS = c.sS.data(); p.sigT = c.sT.data(); p.sigE = c.sE.data();
	p.phs5 = c.p5.data(); p.phs3 = c.p3.data();
	p.exin_left = exin_left;  p.exin_right = exin_right;	// b->left / right when the Exinon was built
	p.a_left = a->left; p.a_right = a->right; p.b_left = b->left; p.b_right = b->right;
	p.a_exgl = a->inex.exgl; p.a_exgr = a->inex.exgr; p.b_exgl = b->inex.exgl; p.b_exgr = b->inex.exgr;
}

static void fill_exact_h(SpdpScoringH& sc, SpdpProblemH& p, const Seq* b, const PwdB* pwd, SeedCols& sx, bool with_table = true) {
	// the exact-model inputs the walk prices its joins with (and the scalar engine behind its small DP calls)
	if (with_table) {
	    sx.ip.resize(b->len + 2);
	    for (int l = 0; l < (int) sx.ip.size(); ++l) sx.ip[l] = pwd->IntPen->Penalty(l);
	    sc.intpen = sx.ip.data();  sc.intpen_len = (int) sx.ip.size();
	}
	sc.lgop = pwd->LongGOP; sc.gape1 = pwd->GapE1; sc.gape2 = pwd->GapE2; sc.extragop = pwd->ExtraGOP;
	sc.diffu = pwd->diffu; sc.k1 = alprm.k1; sc.minl = IntronPrm.minl;
	sx.dc.assign(b->len + 3, 0);
	std::vector<unsigned char> d5(b->len + 3, 0), d3(b->len + 3, 0);
	int	nc = 1;
	for (int i = b->left; i < b->right; ++i) {		// intron53_c on the tron sequence
	    int ch = tnredctab[*b->at(i)];
	    if (ch >= 4) ch = 1;
	    nc = ((nc << 2) + ch) & 0xf;
	    if (i - 1 >= 0) d5[i - 1] = nc;
	    d3[i + 1] = nc;
	}
	int	mrep[16], nrep[16];
	for (int u = 0; u < 16; ++u) mrep[u] = nrep[u] = -1;
	for (int n = b->left; n <= b->right; ++n) {
	    sx.dc[n] = (uint8_t) (d5[n] << 4 | d3[n]);
	    if (n < b->right - 1 && mrep[d5[n]] < 0) mrep[d5[n]] = n;
	    if (n >= b->left + 2 && nrep[d3[n]] < 0) nrep[d3[n]] = n;
	}
	for (int u = 0; u < 16; ++u)
	    for (int v = 0; v < 16; ++v)
		sc.t53[16 * u + v] = (mrep[u] >= 0 && nrep[v] >= 0)?
		    (int16_t) (b->exin->sig53(mrep[u], nrep[v], IE53) - b->exin->score_p(nrep[v])->sig3): 0;
	p.dinc = sx.dc.data();
	sc.scalar_engines = algmode.alg == 0? 1: (algmode.alg == 1? 2: 0);
}


// the parameters of the seeded walk from the globals they live in (src/wln.h, src/codepot.h, src/aln.h)
static void fill_seed_params(SpdpSeedParams& sp, const PwdB* pwd, const Seq* b) {
	memset(&sp, 0, sizeof sp);
	sp.qck = algmode.qck;
	for (int l = 0; l < 4; ++l) sp.wl_width[l] = setwlprm(l)->width;
	sp.elmt = IntronPrm.elmt; sp.minl = IntronPrm.minl; sp.vthr = (int) pwd->Vthr; sp.desert = alprm2.desert;
	sp.maxsp = alprm.maxsp; sp.crs = algmode.crs; sp.smn4 = getsmn(4); sp.w2 = alprm2.w;
	sp.gc_sig5 = b->exin->gc_sig5; sp.lcl = algmode.lcl; sp.codonk1 = pwd->codonk1;
	sp.any = algmode.any; sp.both_ori = 0; sp.ip_maxl = IntronPrm.maxl; sp.ip_mode = IntronPrm.mode;
}

// the model of the library's own HSP search (SpdpSeedParams.wilip; spdp_wilip.h restates Wilip): the word parameters of the
// three levels, the HSP-search matrix and the scalars Wlp reads.  wlparams / hspprm are file statics of src/wln.cc: what is
// needed of them follows from their definitions (EndBonus = (VTYPE) AvTrc() / 2, src/wln.cc:146); AvrSig of the intron
// penalty is private: PenaltyPlus(n) - Penalty(n)
static void fill_wilip_model(SpdpWilipModel& m, const PwdB* pwd) {
	memset(&m, 0, sizeof m);
	for (INT l = 0; l < MaxWlpLevel; ++l) {
const	    WLPRM* p = setwlprm(l);
	    SpdpWilipLevel& L = m.level[l];
	    L.elem = p->elem; L.tpl = p->tpl; L.mask = p->mask; L.width = p->width; L.gain = p->gain; L.gain1 = p->gain1;
	    L.thr = p->thr; L.xdrp = p->xdrp; L.cutoff = p->cutoff; L.vthr = p->vthr;
	    L.bitpat_len = p->bitpat? (int) strlen(p->bitpat): 0;
	    for (int i = 0; i < L.bitpat_len && i < 32; ++i) L.bitpat[i] = p->bitpat[i] == '1';
	    for (int c = 0; c < 32; ++c) L.convtab[c] = (c <= ZZZ && p->ConvTab)? (uint8_t) std::min<INT>(p->ConvTab[c], 255): 127;
	}
const	Simmtx*	sm = getSimmtx(WlnPamNo);
	m.mtx_rows = sm->rows? sm->rows: sm->dim;  m.mtx_cols = sm->dim;
	for (int i = 0; i < m.mtx_rows; ++i)
	    for (int j = 0; j < m.mtx_cols; ++j) m.mtx[i * m.mtx_cols + j] = sm->mtx[i][j];
	m.dvsp = pwd->DvsP;  m.end_bonus = (VTYPE) sm->AvTrc() / 2;
	m.crs = algmode.crs;  m.lsg = algmode.lsg;  m.mlt = algmode.mlt;
	m.hard_minl = IntronPrm.hard_minl;  m.hard_maxl = IntronPrm.hard_maxl;  m.minl = IntronPrm.minl;  m.maxl = IntronPrm.maxl;
	m.llmt = IntronPrm.llmt;
	for (int n = IntronPrm.llmt; n < IntronPrm.llmt + 64; ++n)
	    if (pwd->IntPen->Penalty(n) > SHRT_MIN) { m.avrsig = pwd->IntPen->PenaltyPlus(n) - pwd->IntPen->Penalty(n); break; }
	m.shortquery = shortquery;  m.min_hit = 3;  m.met = MET;  m.ser = SER;  m.ser2 = SER2;
}

// one Wilip search on a sub-range, flattened as SpdpHspSource::units wants it
static void wilip_flat(Seq** seqs, const PwdB* pwd, int level, const int32_t span[8], std::vector<int32_t>& L) {
	Seq*	a = seqs[0];
	Seq*	b = seqs[1];
const	RANGE	ra = {a->left, a->right}, rb = {b->left, b->right};
const	INEX	ia = a->inex, ib = b->inex;
	a->left = span[0]; a->right = span[1]; b->left = span[2]; b->right = span[3];
	a->inex.exgl = span[4]; a->inex.exgr = span[5]; b->inex.exgl = span[6]; b->inex.exgr = span[7];	// (Wlp's end bonus reads them)
	{
	    Wilip	wl((const Seq**) seqs, pwd, level);		// src/wln.cc:980
	    L.clear();
const	    WLUNIT*	wlu = wl.begin();
const	    int	nw = wlu? wl.size(): 0;
	    L.push_back(nw);
	    for (int u = 0; u < nw; ++u) {
const		WLUNIT& x = wlu[u];
const		int uh[6] = {x.num, x.nid, x.tlen, x.llmt, x.ulmt, (int) x.scr};
		L.insert(L.end(), uh, uh + 6);
		for (int j = 0; j <= x.num; ++j) {		// num HSPs + the slot behind them
const		    JUXT& t = x.jxt[j];
const		    int jr[5] = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		    L.insert(L.end(), jr, jr + 5);
		}
	    }
	}
	a->left = ra.left; a->right = ra.right; b->left = rb.left; b->right = rb.right;
	a->inex = ia; b->inex = ib;
}
#endif
