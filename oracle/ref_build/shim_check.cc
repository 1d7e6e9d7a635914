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
// usage: shim_check [-Q n] genome.fa query.fa      (cDNA / EST or protein query; -Q0 semantics)
//   -Q n (cDNA): algmode.qck = n -- the seeded path.  The reference's own geneorient() finds the HSPs and its own Wilip
//   answers the recursion levels through SpdpHspSource; alignS_ng of the reference against spdp_align_s_seeded.

#include "ref_dump_common.h"
#include "shim_fill.h"		// integration/shim_fill.h: the reference-side binding under test

static SpdpContext* g_ctx = 0;
static bool g_undefined = false;		// the library flagged the input as undefined in the reference (n_skl < 0)

// ======================================================================================
// The shim: ~80 lines a maintainer adds to sblib
// ======================================================================================
static VTYPE HomScoreS_gpu(const Seq* seqs[], const PwdB* pwd) {	// == HomScoreS_ng, -A2/-A3
	SpdpScoring sc;  SpdpProblem p;  std::vector<int16_t> s5, s3;  int32_t scr;  SeedCols c;
	fill_scoring(sc, pwd, seqs[1]);  fill_problem(p, seqs[0], seqs[1], s5, s3);
	fill_exact_s(sc, p, seqs[1], pwd, c);	// the exact intron-length engines: -A0 / -A1, and sub-problems below 8 query rows under every -A
	if (spdp_homscore_s(g_ctx, &sc, &p, 1, &scr)) fatal("%s\n", spdp_last_error(g_ctx));
	return scr;
}

static SKL* alignS_gpu(Seq* seqs[], const PwdB* pwd, Gsinfo* gsi) {	// == alignS_ng(.., ori = 1), -Q0/-Q4
	SpdpScoring sc;  SpdpProblem p;  std::vector<int16_t> s5, s3;  SpdpAlignment al;  SeedCols c;
	fill_scoring(sc, pwd, seqs[1]);  fill_problem(p, seqs[0], seqs[1], s5, s3);
	fill_exact_s(sc, p, seqs[1], pwd, c);		// (a ladder under -A2 / -A3 meets sub-problems below 8 rows too: small MaxVmfSpace, deep recursion)
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

static SKL* alignH_gpu(const Seq* seqs[], const PwdB* pwd, Gsinfo* gsi, int exin_left, int exin_right) {	// == alignH_ng, -Q0/-Q4
	SpdpScoringH sc;  SpdpProblemH p;  HCols c;  SpdpAlignment al;  SeedCols sx;
	fill_scoring_h(sc, pwd, seqs[1]);  fill_problem_h(p, seqs[0], seqs[1], c, exin_left, exin_right);
	p.a_pad = *seqs[0]->at(seqs[0]->len);
	fill_exact_h(sc, p, seqs[1], pwd, sx);		// -A0 / -A1, and sub-problems below 8 rows under every -A
const	int rc = spdp_align_h(g_ctx, &sc, &p, 1, &al);
	if (rc < 0) fatal("%s\n", spdp_last_error(g_ctx));
	if (rc == 1 || al.n_skl < 0) {			// engine not built / reference-undefined input: a production shim falls
	    g_undefined = true;				// back to the host here (the reference's result is garbage that differs
	    return alignH_ng(seqs, pwd, gsi);		// from run to run on such inputs); the checker reports the case as such
	}
	gsi->scr = al.score;
	if (!al.n_skl) return 0;
	SKL* skl = new SKL[al.n_skl + 1];
	memcpy(skl, al.skl, sizeof(SKL) * al.n_skl);
	skl[al.n_skl].m = skl[al.n_skl].n = EOS;
	spdp_free_alignments(&al, 1);
	return skl;
}

// ---- the seeded path (-Q5 .. -Q7): globalS_ng with algmode.qck != 0 ------------------------------------------------
// What a maintainer adds beside alignS_gpu: the exact-model inputs the walk prices its joins with, the parameters of
// SpdpSeedParams from the globals they live in, b->jxt as the HSP list, and the reference's own Wilip behind the
// SpdpHspSource callback (called from the walk's thread; one query here, so the shared Seq ranges are safe to move).
struct SeedSrc { Seq** seqs; const PwdB* pwd; SeedCols* cols; };

static int wilip_units(void* user, int32_t, int32_t level, const int32_t span[8], const int32_t** flat, int32_t* n_flat)
{
	SeedSrc* S = (SeedSrc*) user;
	wilip_flat(S->seqs, S->pwd, level, span, S->cols->flat);
	*flat = S->cols->flat.data(); *n_flat = (int32_t) S->cols->flat.size();
	return 0;
}

static SKL* alignS_seeded_gpu(Seq* seqs[], const PwdB* pwd, Gsinfo* gsi) {	// == alignS_ng(.., ori = 1), -Q5 .. -Q7
	Seq*	a = seqs[0];
	Seq*	b = seqs[1];
	SpdpScoring sc;  SpdpProblem p;  std::vector<int16_t> s5, s3;  SpdpAlignment al;  SeedCols c;
	fill_scoring(sc, pwd, b);  fill_problem(p, a, b, s5, s3);
	fill_exact_s(sc, p, b, pwd, c);
	SpdpSeedParams sp;
	fill_seed_params(sp, pwd, b);
	std::vector<SpdpJuxt> jx;
	for (int j = 0; b->jxt && j <= b->CdsNo; ++j) {		// CdsNo HSPs + the free slot behind them
const	    JUXT& t = b->jxt[j];
	    SpdpJuxt q = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
	    jx.push_back(q);
	}
const	SpdpJuxt* lists[1] = {jx.empty()? 0: jx.data()};
const	int32_t	counts[1] = {b->jxt? b->CdsNo: 0};
const	int32_t	lowest[1] = {b->wllvl};
	SeedSrc	ss = {seqs, pwd, &c};
	SpdpHspSource src = {&ss, wilip_units, 0};
const	int rc = spdp_align_s_seeded(g_ctx, &sc, &sp, &p, 1, lists, counts, lowest, &src, &al);
	if (rc != 0) fatal("spdp_align_s_seeded: %s\n", spdp_last_error(g_ctx));
	gsi->scr = al.score;
	if (!al.n_skl) return 0;
	SKL* skl = new SKL[al.n_skl + 1];
	memcpy(skl, al.skl, sizeof(SKL) * al.n_skl);
	skl[al.n_skl].m = skl[al.n_skl].n = EOS;
	spdp_free_alignments(&al, 1);
	return skl;
}

static SKL* alignH_seeded_gpu(Seq* seqs[], const PwdB* pwd, Gsinfo* gsi, int exin_left, int exin_right) {	// == alignH_ng, -Q5 .. -Q7
	Seq*	a = seqs[0];
	Seq*	b = seqs[1];
	SpdpScoringH sc;  SpdpProblemH p;  HCols c;  SpdpAlignment al;  SeedCols sx;
	fill_scoring_h(sc, pwd, b);  fill_problem_h(p, a, b, c, exin_left, exin_right);
	p.a_pad = *a->at(a->len);			// what exg_seq left behind the query
	fill_exact_h(sc, p, b, pwd, sx);
	SpdpSeedParams sp;
	fill_seed_params(sp, pwd, b);
	std::vector<SpdpJuxt> jx;
	for (int j = 0; b->jxt && j <= b->CdsNo; ++j) {
const	    JUXT& t = b->jxt[j];
	    SpdpJuxt q = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
	    jx.push_back(q);
	}
const	SpdpJuxt* lists[1] = {jx.empty()? 0: jx.data()};
const	int32_t	counts[1] = {b->jxt? b->CdsNo: 0};
const	int32_t	lowest[1] = {b->wllvl};
	SeedSrc	ss = {seqs, pwd, &sx};
	SpdpHspSource src = {&ss, wilip_units, 0};
const	int rc = spdp_align_h_seeded(g_ctx, &sc, &sp, &p, 1, lists, counts, lowest, &src, &al);
	if (rc < 0) fatal("spdp_align_h_seeded: %s\n", spdp_last_error(g_ctx));
	if (rc == 1) return (SKL*) -1;			// a DP call the reference itself leaves undefined: not comparable
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
	int	seeded_q = 0, crs = -1, local = 0, alg = 2, ls = 2;
	long	vmfspace = 0;
	while (argc > 3 && argv[1][0] == '-') {		// -Q n, -X crs, -L, -C, -V space: as ref_dump's options of the same name
	    const char c = argv[1][1];
	    if (c == 'L') { local = 1; ++argv; --argc; continue; }
	    if (c == 'C') { local = 3; ++argv; --argc; continue; }
	    if (argc < 5) break;
	    if (c == 'Q') seeded_q = atoi(argv[2]) & 3;
	    else if (c == 'X') crs = atoi(argv[2]);
	    else if (c == 'V') vmfspace = atol(argv[2]);
	    else if (c == 'A') alg = atoi(argv[2]);
	    else if (c == 'l') ls = atoi(argv[2]);		// alprm.ls: 3 = double affine gaps (-yl3)
	    else break;
	    argv += 2; argc -= 2;
	}
	if (argc != 3) { fprintf(stderr, "usage: shim_check [-Q n] [-A n] [-l ls] [-X crs] [-L | -C] [-V space] genome.fa query.fa\n"); return 2; }
	g_ctx = spdp_create(0);
	if (!g_ctx) { fprintf(stderr, "shim_check: no HIP device\n"); return 3; }
const	char*	files[2] = {argv[1], argv[2]};
	set_default_params();
	optimize(GLOBAL, MAXIMUM);
	algmode.qck = 0;		// -Q0
	algmode.blk = 0;
	alprm.ls = ls;
	if (local) algmode.lcl |= 16;
	if (local == 3) algmode.lcl |= 32;
	if (crs >= 0) algmode.crs = crs;
	if (vmfspace) MaxVmfSpace = (int) vmfspace;
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
	algmode.alg = alg;		// -A0 / -A1 / -A2 / -A3 (the quantile table above needs alg > 1 at PwdB's construction)
	if (alg == 3) IntronPrm.nquant = 1;
	a->inex.intr = 0;
	if (!protein) a->inex.ori = 1;
	if (protein && seeded_q) b->comrev(seqs + 2);	// (match_2, spaln.cc:748-756: the other strand first, then both to tron codes)
	if (protein) b->nuc2tron();
const	int	exin_left = b->left, exin_right = b->right;
	if (protein) b->exin = new Exinon(b, pwd, false);
	if (protein && seeded_q) { seqs[2]->nuc2tron(); seqs[2]->exin = new Exinon(seqs[2], pwd, false); }
	if (algmode.lcl & 16) { a->exg_seq(1, 1); b->exg_seq(1, 1); }
	else {
	    a->exg_seq(algmode.lcl & 4, algmode.lcl & 8);
	    b->exg_seq(algmode.lcl & 1, algmode.lcl & 2);
	}
	if (!protein) b->exin = new Exinon(b, pwd, false);
const	RANGE	ra = {a->left, a->right}, rb = {b->left, b->right};
const	INEX	ia = a->inex, ib = b->inex;
	auto restore = [&]() {
	    a->left = ra.left; a->right = ra.right; b->left = rb.left; b->right = rb.right;
	    a->inex = ia; b->inex = ib;
	};
	bool	ok = true;
	if (!protein && seeded_q) {
	    // match_2's set-up for the seeded path (spaln.cc:742-776): the other strand beside this one, HSPs from geneorient()
	    algmode.qck = seeded_q;
	    Seq* const	b0 = b;
	    b->comrev(seqs + 2);
	    seqs[2]->exin = new Exinon(seqs[2], pwd, false);
const	    int	np = geneorient(seqs, pwd);
	    if (b != b0) { fprintf(stderr, "shim_check -Q: the reverse strand won geneorient()\n"); return 4; }
	    // the walk edits the phase marks and the slot behind the HSP list: both runs start from the same state
	    std::vector<SGPT2> sg0(b->right - b->left + 1);
	    for (int n = b->left; n <= b->right; ++n) sg0[n - b->left] = *b->exin->score_n(n);
	    std::vector<JUXT> jx0(b->jxt, b->jxt + (b->jxt? b->CdsNo + 1: 0));
	    Gsinfo	g_ref, g_gpu;
	    g_ref.skl = alignS_ng(seqs, pwd, &g_ref, 1);
	    restore();
	    for (int n = b->left; n <= b->right; ++n) *b->exin->score_n(n) = sg0[n - b->left];
	    if (b->jxt) vcopy(b->jxt, jx0.data(), jx0.size());
	    g_gpu.skl = alignS_seeded_gpu(seqs, pwd, &g_gpu);
	    ok = g_ref.scr == g_gpu.scr && same_skl(g_ref.skl, g_gpu.skl);
	    int64_t st[6] = {0};
	    spdp_seeded_stats(g_ctx, st, 6);
	    printf("cDNA query %d nt, window %d nt, -Q%d: %d HSP(s) at level %d\nalignS score: reference %d, GPU %d\n"
		   "device batches %d (lspS_ng calls %d, tracebacks %d), Wilip calls through the callback %d\n",
		a->len, b->len, seeded_q + 4, np? b->CdsNo: 0, b->wllvl, (int) g_ref.scr, (int) g_gpu.scr,
		(int) st[0], (int) st[1], (int) st[2], (int) st[4]);
	    print_skl("reference SKL", g_ref.skl);
	    print_skl("GPU       SKL", g_gpu.skl);
	} else if (!protein) {
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
	} else if (seeded_q) {
	    algmode.qck = seeded_q;
	    Seq* const	b0 = b;
const	    int	np = geneorient(seqs, pwd);
	    if (b != b0) { fprintf(stderr, "shim_check -Q: the reverse strand won geneorient()\n"); return 4; }
	    std::vector<SGPT6> sg0;
	    for (int n = std::max(0, b->left - 1); n <= b->right + 1; ++n) sg0.push_back(*b->exin->score_p(n));
	    std::vector<JUXT> jx0(b->jxt, b->jxt + (b->jxt? b->CdsNo + 1: 0));
	    Gsinfo	g_ref, g_gpu;
	    g_ref.skl = alignH_ng((const Seq**) seqs, pwd, &g_ref);
	    restore();
	    for (int n = std::max(0, b->left - 1), i = 0; n <= b->right + 1; ++n, ++i) *b->exin->score_p(n) = sg0[i];
	    if (b->jxt) vcopy(b->jxt, jx0.data(), jx0.size());
	    g_gpu.skl = alignH_seeded_gpu(seqs, pwd, &g_gpu, exin_left, exin_right);
	    if (g_gpu.skl == (SKL*) -1) { printf("a DP call of this case is undefined in the reference\n"); g_gpu.skl = 0; spdp_destroy(g_ctx); return 5; }
	    ok = g_ref.scr == g_gpu.scr && same_skl(g_ref.skl, g_gpu.skl);
	    int64_t st[6] = {0};
	    spdp_seeded_stats(g_ctx, st, 6);
	    printf("protein query %d aa, window %d nt, -Q%d: %d HSP(s) at level %d\nalignH score: reference %d, GPU %d\n"
		   "device batches %d (lspH_ng calls %d, tracebacks %d, with a cut range %d), Wilip calls through the callback %d\n",
		a->len, b->len, seeded_q + 4, np? b->CdsNo: 0, b->wllvl, (int) g_ref.scr, (int) g_gpu.scr,
		(int) st[0], (int) st[1], (int) st[2], (int) st[3], (int) st[4]);
	    print_skl("reference SKL", g_ref.skl);
	    print_skl("GPU       SKL", g_gpu.skl);
	} else {
	    Gsinfo	g_ref, g_gpu;
	    g_ref.skl = alignH_ng((const Seq**) seqs, pwd, &g_ref);
	    restore();
	    g_gpu.skl = alignH_gpu((const Seq**) seqs, pwd, &g_gpu, exin_left, exin_right);
	    if (g_undefined) { printf("the library reports this input as undefined in the reference\n"); spdp_destroy(g_ctx); return 5; }
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
