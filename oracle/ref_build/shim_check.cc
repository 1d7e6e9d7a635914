// shim_check -- the reference-side binding of INTEGRATION.md, compiled against the reference's own
// headers and linked with libspdp_hip.so.  TEST INFRASTRUCTURE ONLY (built by oracle/ref_build/Makefile
// into oracle/_ref/, where /root/reference exists; runs wherever a GPU is).
//
// It is the proof that the C ABI is a drop-in from the reference's side: the program sets a
// (genomic window, query) pair up exactly as the reference does (Seq, PwdB, Exinon), runs the
// reference's own HomScoreS_ng / alignS_ng (or HomScoreH_ng / alignH_ng for a protein query) under
// -A2, then the *_gpu replacements below -- which see nothing but the reference's objects -- and
// compares scores and SKL corner lists.  Exit status 0 = identical.
//
// usage: shim_check genome.fa query.fa      (cDNA / EST or protein query; -Q0 semantics)

#include "ref_dump_common.h"
#include "spdp.h"

static SpdpContext* g_ctx = 0;

// ======================================================================================
// The shim: ~80 lines a maintainer adds to sblib
// ======================================================================================
static void fill_scoring(SpdpScoring& sc, const PwdB* pwd, const Seq* b) {
	memset(&sc, 0, sizeof sc);
const	Simmtx* sm = pwd->simmtx;			// src/simmtx.h:35-63
	sc.mtx_dim = sm->dim;
	for (int i = 0; i < sm->dim; ++i)
	    for (int j = 0; j < sm->dim; ++j) sc.mtx[i * sm->dim + j] = sm->mtx[i][j];
	sc.gop = pwd->BasicGOP;  sc.gep = pwd->BasicGEP;	// src/aln.h:243-244
	sc.lgop = pwd->LongGOP;  sc.lgep = pwd->LongGEP;  sc.noll = pwd->Noll;
	sc.spj = b->inex.intr;
	sc.llmt = IntronPrm.llmt;			// src/codepot.h:207-214
	sc.ipen = pwd->IntPen->Penalty();		// GapWI, src/codepot.h:241
	sc.nquant = IntronPrm.nquant;			// 1 when -A3 (src/fwd2s1.cc:125)
	for (int j = 0; j < sc.nquant; ++j) {
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

static VTYPE HomScoreS_gpu(const Seq* seqs[], const PwdB* pwd) {	// == HomScoreS_ng, -A2/-A3
	SpdpScoring sc;  SpdpProblem p;  std::vector<int16_t> s5, s3;  int32_t scr;
	fill_scoring(sc, pwd, seqs[1]);  fill_problem(p, seqs[0], seqs[1], s5, s3);
	if (spdp_homscore_s(g_ctx, &sc, &p, 1, &scr)) fatal("%s\n", spdp_last_error(g_ctx));
	return scr;
}

static SKL* alignS_gpu(Seq* seqs[], const PwdB* pwd, Gsinfo* gsi) {	// == alignS_ng(.., ori = 1), -Q0/-Q4
	SpdpScoring sc;  SpdpProblem p;  std::vector<int16_t> s5, s3;  SpdpAlignment al;
	fill_scoring(sc, pwd, seqs[1]);  fill_problem(p, seqs[0], seqs[1], s5, s3);
	if (spdp_align_s(g_ctx, &sc, &p, 1, &al) < 0) fatal("%s\n", spdp_last_error(g_ctx));
	gsi->scr = al.score;
	if (!al.n_skl) return 0;			// "no alignment", as the reference
	SKL* skl = new SKL[al.n_skl + 1];		// caller (~Gsinfo) delete[]s it
	memcpy(skl, al.skl, sizeof(SKL) * al.n_skl);	// SpdpSkl is layout-identical to SKL
	skl[al.n_skl].m = skl[al.n_skl].n = EOS;
	if (seqs[0]->inex.sens) skl->m |= A_RevCom;
	spdp_free_alignments(&al, 1);
	return skl;
}

static void fill_scoring_h(SpdpScoringH& sc, const PwdB* pwd, const Seq* b) {
	memset(&sc, 0, sizeof sc);
const	Simmtx* sm = pwd->simmtx;			// aa x tron matrix: rows x dim
	sc.mtx_rows = sm->rows;  sc.mtx_cols = sm->dim;
	for (int i = 0; i < sm->rows; ++i)
	    for (int j = 0; j < sm->dim; ++j) sc.mtx[i * sm->dim + j] = sm->mtx[i][j];
	sc.gop = pwd->BasicGOP;  sc.gep = pwd->BasicGEP;  sc.lgep = pwd->LongGEP;
	sc.codonk1 = pwd->codonk1;			// GapExtPen3, src/aln.h:302
	sc.gapw1 = pwd->GapW1;  sc.gapw2 = pwd->GapW2;  sc.gapw3 = pwd->GapW3;
	sc.spj = b->inex.intr;  sc.llmt = IntronPrm.llmt;  sc.ipen = pwd->IntPen->Penalty();
	sc.nquant = IntronPrm.nquant;			// 1 under -A3 (src/fwd2h1.cc:127)
	for (int j = 0; j < sc.nquant; ++j) { sc.qm_len[j] = pwd->IntPen->qm[j].len; sc.qm_pen[j] = pwd->IntPen->qm[j].pen; }
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

static SKL* alignH_gpu(const Seq* seqs[], const PwdB* pwd, Gsinfo* gsi, int exin_left, int exin_right) {	// == alignH_ng, -Q0/-Q4
	SpdpScoringH sc;  SpdpProblemH p;  HCols c;  SpdpAlignment al;
	fill_scoring_h(sc, pwd, seqs[1]);  fill_problem_h(p, seqs[0], seqs[1], c, exin_left, exin_right);
const	int rc = spdp_align_h(g_ctx, &sc, &p, 1, &al);
	if (rc < 0) fatal("%s\n", spdp_last_error(g_ctx));
	if (rc == 1 || al.n_skl < 0) return alignH_ng(seqs, pwd, gsi);	// engine not built / reference-undefined input
	gsi->scr = al.score;
	if (!al.n_skl) return 0;
	SKL* skl = new SKL[al.n_skl + 1];
	memcpy(skl, al.skl, sizeof(SKL) * al.n_skl);
	skl[al.n_skl].m = skl[al.n_skl].n = EOS;
	spdp_free_alignments(&al, 1);
	return skl;
}

// ======================================================================================
// driver: reference vs shim on one pair
// ======================================================================================
static bool same_skl(const SKL* x, const SKL* y)
{
	if (!x || !y) return x == y;
	if (x->n != y->n || x->m != y->m) return false;
	for (int i = 1; i <= x->n; ++i)
	    if (x[i].m != y[i].m || x[i].n != y[i].n) return false;
	return true;
}

static void print_skl(const char* tag, const SKL* s)
{
	printf("%s:", tag);
	if (!s) { printf(" (none)\n"); return; }
	for (int i = 1; i <= s->n; ++i) printf(" (%d,%d)", s[i].m, s[i].n);
	printf("\n");
}

int main(int argc, const char** argv)
{
	if (argc != 3) { fprintf(stderr, "usage: shim_check genome.fa query.fa\n"); return 2; }
	g_ctx = spdp_create(0);
	if (!g_ctx) { fprintf(stderr, "shim_check: no HIP device\n"); return 3; }
const	char*	files[2] = {argv[1], argv[2]};
	set_default_params();
	optimize(GLOBAL, MAXIMUM);
	algmode.qck = 0;		// -Q0
	algmode.blk = 0;
	alprm.ls = 2;
	OutPrm.all_out = 1;
	Seq*	seqs[4];
	initseq(seqs, 4);
	Seq*&	a = seqs[0];
	Seq*&	b = seqs[1];
	SeqServer	svr(2, files, IM_SNGL, 0, UNKNOWN, UNKNOWN);
	if (svr.nextseq(b, 1) == IS_END) { fprintf(stderr, "no genome\n"); return 2; }
	if (svr.nextseq(a, 0) != IS_OK) { fprintf(stderr, "no query\n"); return 2; }
const	bool	protein = a->isprotein();
	b->inex.intr = algmode.lsg;
	makeWlprms(prePwd((const Seq**) seqs));
	algmode.alg = 2;		// -A2: the `_wip` engines
	PwdB*	pwd = new PwdB((const Seq**) seqs);
	makeStdSig53();
	a->inex.intr = 0;
	if (!protein) a->inex.ori = 1;
	if (protein) b->nuc2tron();
const	int	exin_left = b->left, exin_right = b->right;
	if (protein) b->exin = new Exinon(b, pwd, false);
	a->exg_seq(algmode.lcl & 4, algmode.lcl & 8);
	b->exg_seq(algmode.lcl & 1, algmode.lcl & 2);
	if (!protein) b->exin = new Exinon(b, pwd, false);
const	RANGE	ra = {a->left, a->right}, rb = {b->left, b->right};
const	INEX	ia = a->inex, ib = b->inex;
	auto restore = [&]() {
	    a->left = ra.left; a->right = ra.right; b->left = rb.left; b->right = rb.right;
	    a->inex = ia; b->inex = ib;
	};
	bool	ok = true;
	if (!protein) {
	    VTYPE	h_ref = HomScoreS_ng((const Seq**) seqs, pwd);
	    restore();
	    VTYPE	h_gpu = HomScoreS_gpu((const Seq**) seqs, pwd);
	    restore();
	    Gsinfo	g_ref, g_gpu;
	    g_ref.skl = alignS_ng(seqs, pwd, &g_ref, 1);
	    restore();
	    g_gpu.skl = alignS_gpu(seqs, pwd, &g_gpu);
	    ok = h_ref == h_gpu && g_ref.scr == g_gpu.scr && same_skl(g_ref.skl, g_gpu.skl);
	    printf("cDNA query %d nt, window %d nt\nHomScoreS: reference %d, GPU %d\nalignS score: reference %d, GPU %d\n",
		a->len, b->len, (int) h_ref, (int) h_gpu, (int) g_ref.scr, (int) g_gpu.scr);
	    print_skl("reference SKL", g_ref.skl);
	    print_skl("GPU       SKL", g_gpu.skl);
	} else {
	    Gsinfo	g_ref, g_gpu;
	    g_ref.skl = alignH_ng((const Seq**) seqs, pwd, &g_ref);
	    restore();
	    g_gpu.skl = alignH_gpu((const Seq**) seqs, pwd, &g_gpu, exin_left, exin_right);
	    ok = g_ref.scr == g_gpu.scr && same_skl(g_ref.skl, g_gpu.skl);
	    printf("protein query %d aa, window %d nt\nalignH score: reference %d, GPU %d\n",
		a->len, b->len, (int) g_ref.scr, (int) g_gpu.scr);
	    print_skl("reference SKL", g_ref.skl);
	    print_skl("GPU       SKL", g_gpu.skl);
	}
	printf(ok? "IDENTICAL\n": "DIFFERENT\n");
	spdp_destroy(g_ctx);
	return ok? 0: 1;
}
