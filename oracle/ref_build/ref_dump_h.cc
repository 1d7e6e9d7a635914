// ref_dump_h.cc -- protein x genome half of the golden harness (TEST INFRASTRUCTURE ONLY).
// Separate translation unit: fwd2s1_simd.h and fwd2h1_simd.h cannot share one.
#include <algorithm>
#include "ref_dump_common.h"
#include "fwd2h1_simd.h"

// ---------------------------------------------------------------- protein x genome (Fwd2h1)
// Same idea for the aa x 3-frame path: SimdAln2h1 (fwd2h1_simd.h:69-382) and the
// HomScoreH_ng / alignH_ng surface (fwd2h1.cc:3288, 3310).  Set-up order as in match_2
// (spaln.cc:742-765): nuc2tron, Exinon, then exg_seq.
int dump_protein(Seq** seqs, const char* exg, const std::vector<int>& udh_list, const char* outfn)
{
	Seq*&	a = seqs[0];
	Seq*&	b = seqs[1];
	b->inex.intr = algmode.lsg;
	makeWlprms(prePwd((const Seq**) seqs));
	algmode.alg = 2;
	PwdB*	pwd = new PwdB((const Seq**) seqs);
	makeStdSig53();
	a->inex.intr = 0;
	if (g_seeded_q) b->comrev(seqs + 2);		// (match_2, spaln.cc:748-756: the other strand first, then both to tron codes)
	b->nuc2tron();
	b->exin = new Exinon(b, pwd, false);
	if (g_seeded_q) {
	    seqs[2]->nuc2tron();
	    seqs[2]->exin = new Exinon(seqs[2], pwd, false);
	}
	if (algmode.lcl & 16) {
	    a->exg_seq(1, 1);
	    b->exg_seq(1, 1);
	} else {
	    a->exg_seq(algmode.lcl & 4, algmode.lcl & 8);
	    b->exg_seq(algmode.lcl & 1, algmode.lcl & 2);
	}
	if (exg) {
	    a->exg_seq(exg[0] == '1', exg[1] == '1');
	    b->exg_seq(exg[2] == '1', exg[3] == '1');
	}
	if (g_seeded_q) {
	    // HSPs of the lowest level that finds any, as match_2 obtains them (spaln.cc:773-776)
	    algmode.qck = g_seeded_q;
	    Seq* const	b0 = b;
	    const int np = geneorient(seqs, pwd);
	    if (b != b0) { fprintf(stderr, "ref_dump -Q: the reverse strand won geneorient(); not a fixture\n"); return 3; }
	    if (np == 0) fprintf(stderr, "ref_dump -Q: no HSP at any level\n");
	}
	return dump_protein_body(seqs, pwd, udh_list, outfn);
}

// everything from here on reads the pair as the aligner sees it (ranges, Exinon, HSPs): the part a live run shares
// (spaln_dumpq: the reference's own CLI writes the fixture of one named query from inside its alignH_ng call)
int dump_protein_body(Seq** seqs, PwdB* pwd, const std::vector<int>& udh_list, const char* outfn)
{
	Seq*&	a = seqs[0];
	Seq*&	b = seqs[1];
	Writer	w(outfn);
	w.put_int("is_protein", 1);
	w.put("a_codes", 1, a->at(0), a->len);
	w.put_int("a_pad", *a->at(a->len));
	{
	    // SpJunc::spjseq around every ambiguous position of the window: rows {n5, n3, codon of phase 1, codon of phase 2}
	    SpJunc	spj(b, pwd);
	    std::vector<int>	probe;
	    int	n_amb = 0;
	    for (int x = b->left; x < b->right && n_amb < 64; ++x) {
		if (*b->at(x) != AMB) continue;
		++n_amb;
		for (int d = -3; d <= 4; ++d) {
		    const int pr[2][2] = {{x + d, x + d + 61}, {x + d - 61, x + d}};
		    for (int k = 0; k < 2; ++k) {
			const int n5 = pr[k][0], n3 = pr[k][1];
			if (n5 - 2 < 0 || n3 + 1 > b->len || n3 <= 0) continue;
			const CHAR* cs = spj.spjseq(n5, n3);
			probe.push_back(n5); probe.push_back(n3); probe.push_back(cs[0]); probe.push_back(cs[1]);
		    }
		}
	    }
	    if (!probe.empty()) w.put_i32("spj_probe", probe);
	}		// what exg_seq left behind the query (SpdpProblemH::a_pad)
	w.put("b_codes", 1, b->at(0), b->len + 1);	// + the terminator the engine reads (sm_a at n = right + 2)
	{
	    // the signal model behind the SGPT6 arrays (Exinon::intron53_p, codepot.cc:524-611): the position weight
	    // matrices (splice sites, start / stop context, branch point), the coding / intron potential tables (ExinPot,
	    // utilseq.h:90-168) and the scale factors, so that the arrays can be recomputed from b_codes
	    auto dump_pm = [&](const char* tag, const PatMat* pm) {
		char nmb[40];
		std::vector<int> hd = {pm ? pm->rows : 0, pm ? pm->cols : 0, pm ? pm->offset : 0, pm ? pm->order() : 0, pm ? pm->nalpha : 0};
		snprintf(nmb, sizeof nmb, "%s_hdr", tag); w.put_i32(nmb, hd);
		std::vector<int> fl(pm ? 2 + pm->rows * pm->cols : 0);
		if (pm) { memcpy(&fl[0], &pm->tonic, 4); memcpy(&fl[1], &pm->min_elem, 4); memcpy(&fl[2], pm->mtx, sizeof(float) * pm->rows * pm->cols); }
		snprintf(nmb, sizeof nmb, "%s_f32", tag); w.put_i32(nmb, fl);
	    };
	    dump_pm("pm5", pwd->eijpat->pattern5);
	    dump_pm("pm3", pwd->eijpat->pattern3);
	    dump_pm("pmI", pwd->eijpat->patternI);
	    dump_pm("pmT", pwd->eijpat->patternT);
	    dump_pm("pmB", pwd->eijpat->patternB);
	    auto dump_pot = [&](const char* tag, const ExinPot* ep) {
		char nmb[40];
		const int nd = ep ? ep->size() : 0, tot = ep ? (int) (ep->end() - ep->begin()) : 0;
		std::vector<int> hd = {nd, nd ? tot / nd : 0};
		snprintf(nmb, sizeof nmb, "%s_hdr", tag); w.put_i32(nmb, hd);
		std::vector<int> fl(tot);
		if (tot) memcpy(fl.data(), ep->begin(), sizeof(float) * tot);
		snprintf(nmb, sizeof nmb, "%s_f32", tag); w.put_i32(nmb, fl);
	    };
	    dump_pot("potC", pwd->codepot);
	    dump_pot("potI", pwd->intnpot);
	    dump_pot("potX", pwd->exonpot);
	    const float fact = (float) b->exin->fact, fS = (float) b->exin->fS;
	    const float f[10] = {(float) (alprm2.z * b->exin->fact), (float) (alprm2.Z * b->exin->fact), (float) (alprm2.bti * b->exin->fact),
		(float) (bpprm.factor * b->exin->fact), (float) (-alprm2.o * b->exin->fact), fS, (float) (b->exin->fS * alprm2.sss),
		pwd->eijpat->tonic3, pwd->eijpat->tonic5, pwd->eijpat->tonicB};
	    std::vector<int> fb(10);
	    memcpy(fb.data(), f, sizeof f);
	    w.put_i32("sigmodel_f32", fb);		// fE, fI, fT, fB, fO, fS, fs, tonic3, tonic5, tonicB
	    std::vector<int> im = {(int) algmode.any, (int) pwd->DvsP, (int) bpprm.maxb3d, (int) b->many, (int) sizeof(FTYPE), (int) TRM, (int) TRM2};
	    w.put_i32("sigmodel_i32", im);		// any, DvsP, maxb3d, many, sizeof(FTYPE), TRM, TRM2
	    (void) fact;
	    // the per-dinucleotide terms sig53tab[0 / 1][class] (private to Exinon): what is left of a signal after the scaled
	    // matrix score, which the reference's own PatMat::calcPatMat reproduces (same call and range as intron53_p)
	    std::vector<int> tab(32, INT_MIN);
	    if (pwd->eijpat->pattern5 && pwd->eijpat->pattern3) {
		const float fs = f[6];
		// (with the branch-point term on, sig3 carries it: the table is read off a second Exinon built without it)
		PatMat* const	patB = pwd->eijpat->patternB;
		Exinon*	exq = b->exin;
		if (patB) { pwd->eijpat->patternB = 0; exq = new Exinon(b, pwd, false); pwd->eijpat->patternB = patB; }
		--b->left; ++b->right;
		float* p5 = pwd->eijpat->pattern5->calcPatMat(b);
		float* p3 = pwd->eijpat->pattern3->calcPatMat(b);
		++b->left; --b->right;
		std::vector<unsigned char> e5(b->len + 3, 0), e3(b->len + 3, 0);
		int	nc2 = 1;
		for (int i = b->left; i < b->right; ++i) {
		    int c = tnredctab[*b->at(i)];
		    if (c >= 4) c = 1;
		    nc2 = ((nc2 << 2) + c) & 0xf;
		    if (i - 1 >= 0) e5[i - 1] = nc2;
		    e3[i + 1] = nc2;
		}
		int	clash = 0;
		for (int n = b->left + 2; n < b->right - 1; ++n) {
		    const SGPT6* sg = exq->score_p(n);
		    const int t5 = sg->sig5 - (STYPE) (fs * p5[n - b->left + 1]);
		    const int t3 = sg->sig3 - (STYPE) (fs * p3[n - b->left + 1]);
		    if (tab[e5[n]] == INT_MIN) tab[e5[n]] = t5; else if (tab[e5[n]] != t5) ++clash;
		    if (tab[16 + e3[n]] == INT_MIN) tab[16 + e3[n]] = t3; else if (tab[16 + e3[n]] != t3) ++clash;
		}
		if (clash) fprintf(stderr, "ref_dump: %d signal-table clashes\n", clash);
		if (exq != b->exin) delete exq;
		delete[] p5; delete[] p3;
	    }
	    w.put_i32("sig53tab01", tab);
	}
	{
	    const int N = b->len + 3;
	    std::vector<short> v[6];
	    for (int f = 0; f < 6; ++f) v[f].assign(N, 0);
	    std::vector<signed char> p5(N, -2), p3(N, -2);
	    std::vector<unsigned char> good(N, 0);
	    for (int n = std::max(0, b->left - 1); n <= b->right + 1 && n < N; ++n) {
		const SGPT6* sg = b->exin->score_p(n);
		v[0][n] = sg->sig5; v[1][n] = sg->sig3; v[2][n] = sg->sigS;
		v[3][n] = sg->sigT; v[4][n] = sg->sigE; v[5][n] = sg->sigI;
		p5[n] = sg->phs5; p3[n] = sg->phs3;
		good[n] = b->exin->good(sg);
	    }
	    const char* nm[6] = {"sig5", "sig3", "sigS", "sigT", "sigE", "sigI"};
	    for (int f = 0; f < 6; ++f) w.put(nm[f], 2, v[f].data(), N);
	    w.put("phs5", 4, p5.data(), N);
	    w.put("phs3", 4, p3.data(), N);
	    w.put("good", 1, good.data(), N);
	    // exact-model inputs of the rescoring walk (skl_rngH_ng): dinucleotide classes as
	    // Exinon::intron53_c assigns them on a tron sequence (codepot.cc:437-450), the junction table
	    // behind sig53(m, n, IE53) (codepot.cc:411-415) and the intron-length penalty by length
	    std::vector<unsigned char> d5(N, 0), d3(N, 0);
	    int	nc = 1;
	    for (int i = b->left; i < b->right; ++i) {
		int c = tnredctab[*b->at(i)];
		if (c >= 4) c = 1;
		nc = ((nc << 2) + c) & 0xf;
		if (i - 1 >= 0) d5[i - 1] = nc;
		d3[i + 1] = nc;
	    }
	    w.put("dinc5", 1, d5.data(), b->len + 1);
	    w.put("dinc3", 1, d3.data(), b->len + 1);
	    std::vector<int> t53(256, 0), mrep(16, -1), nrep(16, -1);
	    for (int n = b->left; n <= b->right; ++n) {
		if (n < b->right - 1 && mrep[d5[n]] < 0) mrep[d5[n]] = n;
		if (n >= b->left + 2 && nrep[d3[n]] < 0) nrep[d3[n]] = n;
	    }
	    for (int u = 0; u < 16; ++u)
		for (int v2 = 0; v2 < 16; ++v2)
		    if (mrep[u] >= 0 && nrep[v2] >= 0)
			t53[16 * u + v2] = b->exin->sig53(mrep[u], nrep[v2], IE53) - b->exin->score_p(nrep[v2])->sig3;
	    w.put_i32("t53", t53);
	    int	bad = 0;
	    for (int t = 0; t < 400; ++t) {		// self-check of the restated classes
		int m = b->left + (t * 7919) % std::max(1, b->right - b->left - 2);
		int n = b->left + 2 + (t * 104729) % std::max(1, b->right - b->left - 2);
		int want = b->exin->sig53(m, n, IE53);
		int got = b->exin->score_p(n)->sig3 + t53[16 * d5[m] + d3[n]];
		if (want != got) ++bad;
	    }
	    if (bad) fprintf(stderr, "ref_dump: %d sig53 self-check mismatches (protein)\n", bad);
	    std::vector<short> ip(b->len + 2, 0);
	    for (int l = 0; l <= b->len + 1; ++l) ip[l] = (short) pwd->IntPen->Penalty(l);
	    w.put("intpen", 2, ip.data(), ip.size());
	}
	{
	    const Simmtx* sm = pwd->simmtx;
	    std::vector<int> dims = {sm->dim, sm->rows, sm->cols};
	    w.put_i32("mtx_dims", dims);
	    const int R = sm->rows? sm->rows: sm->dim, Cc = sm->cols? sm->cols: sm->dim;
	    std::vector<int>	mtx(R * Cc);
	    for (int i = 0; i < R; ++i)
		for (int j = 0; j < Cc; ++j) mtx[i * Cc + j] = sm->mtx[i][j];
	    w.put_i32("mtx", mtx);
	    w.put_int("avmch", (int) sm->AvTrc());
	}
	{
	    std::vector<int> prm = {
		(int) pwd->BasicGOP, (int) pwd->BasicGEP, (int) pwd->LongGOP, (int) pwd->LongGEP,
		pwd->Noll, (int) pwd->Vthr, (int) pwd->Vab, pwd->codonk1,
		IntronPrm.llmt, IntronPrm.minl, IntronPrm.rlmt, IntronPrm.mu,
		IntronPrm.maxl, IntronPrm.nquant, (int) IntronPrm.hard_minl, (int) IntronPrm.hard_maxl,
		(int) pwd->IntPen->Penalty(), alprm.sh, (int) (algmode.lcl & 16),
		(int) a->inex.exgl, (int) a->inex.exgr, (int) b->inex.exgl, (int) b->inex.exgr,
		a->left, a->right, b->left, b->right, MaxVmfSpace, alprm.ubh,
		(int) b->inex.intr};
	    w.put_i32("params", prm);
	    std::vector<int> hp = {(int) pwd->GapW1, (int) pwd->GapW2, (int) pwd->GapW3, (int) pwd->GapW3L,
		(int) pwd->GapE1, (int) pwd->GapE2, (int) pwd->ExtraGOP, alprm.k1, alprm2.termk1,
		(int) algmode.lcl, pwd->DvsP};
	    w.put_i32("hparams", hp);
	    std::vector<int> rp = {(int) pwd->diffu, (int) OutPrm.supTcodon, (int) (alprm2.Z * 1000), a->many,
		alprm2.jneibr, (int) algmode.lsg, (int) (alprm2.o * 1000)};
	    w.put_i32("rparams", rp);
	    std::vector<int>	ql, qp;
	    for (int j = 0; pwd->IntPen->qm && j < IntronPrm.nquant; ++j) {	// (none under -A0: the quantiles exist for the `_wip` engines only, codepot.cc:163)
		ql.push_back(pwd->IntPen->qm[j].len);
		qp.push_back(pwd->IntPen->qm[j].pen);
	    }
	    w.put_i32("qm_len", ql);
	    w.put_i32("qm_pen", qp);
	}
const	RANGE	ra = {a->left, a->right};
const	RANGE	rb = {b->left, b->right};
const	INEX	ia = a->inex, ib = b->inex;
const	int	nq0 = IntronPrm.nquant;
	auto restore = [&]() {
	    a->left = ra.left; a->right = ra.right;
	    b->left = rb.left; b->right = rb.right;
	    a->inex = ia; b->inex = ib;
	};
	char	nm[48];
	if (g_seeded_q) {
	    // alignH_ng with seeding on (fwd2h1.cc:3310 -> globalH_ng :3267 -> seededH_ng :3180 -> interpolateH :3022)
	    std::vector<int> jx;
	    for (int j = 0; b->jxt && j <= b->CdsNo; ++j) {
		const JUXT& t = b->jxt[j];
		const int jr[5] = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		jx.insert(jx.end(), jr, jr + 5);
	    }
	    w.put_i32("seed_jxt", jx);		// CdsNo HSPs + the slot behind them, scores before addsigEjxt (:2397)
	    dump_wilip_model(w, pwd);
	    float smn4 = getsmn(4), w2 = alprm2.w, maxsp = alprm.maxsp;
	    int smn4b, w2b, maxspb;
	    memcpy(&smn4b, &smn4, 4); memcpy(&w2b, &w2, 4); memcpy(&maxspb, &maxsp, 4);
	    std::vector<int> sp = {(int) algmode.qck, b->wllvl, b->jxt? b->CdsNo: 0,
		(int) setwlprm(0)->width, (int) setwlprm(1)->width, (int) setwlprm(2)->width, (int) setwlprm(3)->width,
		IntronPrm.elmt, IntronPrm.minl, IntronPrm.tlmt, (int) pwd->Vthr, alprm2.desert, maxspb, (int) algmode.crs,
		smn4b, w2b, (int) b->exin->gc_sig5, (int) algmode.lcl, (int) a->inex.ori, (int) b->exin->at_sig5,
		IntronPrm.maxl, IntronPrm.mode};
	    w.put_i32("seed_params", sp);
	    std::vector<SGPT6> sg0;
	    for (int n = std::max(0, b->left - 1); n <= b->right + 1; ++n) sg0.push_back(*b->exin->score_p(n));
	    std::vector<JUXT> jx0(b->jxt, b->jxt + (b->jxt? b->CdsNo + 1: 0));
	    std::vector<int> algs = g_alg_list;
	    if (algs.empty()) { algs.push_back(0); algs.push_back(2); }
	    for (size_t k = 0; k < algs.size(); ++k) {
const		int	alg = algs[k];
		algmode.alg = alg;
		restore();
		for (int n = std::max(0, b->left - 1), i = 0; n <= b->right + 1; ++n, ++i) *b->exin->score_p(n) = sg0[i];
		if (b->jxt) vcopy(b->jxt, jx0.data(), jx0.size());
		wilip_tap_log.clear();
		wilip_tap_on = true;
		Gsinfo	gsi;
		gsi.skl = alignH_ng((const Seq**) seqs, pwd, &gsi);
		wilip_tap_on = false;
		snprintf(nm, sizeof nm, "seed_scr_A%d", alg);
		w.put_int(nm, (int) gsi.scr);
		snprintf(nm, sizeof nm, "seed_skl_A%d", alg);
		w.put_i32(nm, skl2vec(gsi.skl));
		snprintf(nm, sizeof nm, "seed_wilip_A%d", alg);
		w.put_i32(nm, wilip_tap_log);
		{   // what the walk left in the Exinon: the phases it wrote at the junctions it chose itself (src/fwd2h1.cc:2508-2517),
		    // which skl_rngH_ng reads afterwards (:824-825) -- {position, phs5, phs3} wherever a mark differs from the input
		    std::vector<int> mk;
		    for (int n = std::max(0, b->left - 1), i = 0; n <= b->right + 1; ++n, ++i) {
			const SGPT6* g = b->exin->score_p(n);
			if (g->phs5 != sg0[i].phs5 || g->phs3 != sg0[i].phs3) { mk.push_back(n); mk.push_back(g->phs5); mk.push_back(g->phs3); }
		    }
		    snprintf(nm, sizeof nm, "seed_marks_A%d", alg);
		    w.put_i32(nm, mk);
		}
	    }
	    restore();				// (a live caller -- dumpq.cc -- goes on with the pair)
	    for (int n = std::max(0, b->left - 1), i = 0; n <= b->right + 1; ++n, ++i) *b->exin->score_p(n) = sg0[i];
	    if (b->jxt) vcopy(b->jxt, jx0.data(), jx0.size());
	    return 0;
	}
	// -A list: only those selectors, and none of the engine-level `_wip` runs (double affine gaps, -l 3: the reference's
	// `_wip` flavours are not defined there)
	for (int pass = 0; pass < 2 && g_alg_list.empty(); ++pass) {
	    IntronPrm.nquant = pass? 1: nq0;
const	    char*	tag = pass? "q1": "qn";
	    SpJunc	spjcs(b, pwd);
	    WINDOW	wdw;
	    restore();
	    stripe31((const Seq**) seqs, &wdw, alprm.sh);
	    if (pass == 0) {
		std::vector<int> wv = {wdw.lw, wdw.up, wdw.width};
		w.put_i32("wdw", wv);
	    }
	    {
		SimdAln2h1 eng((const Seq**) seqs, pwd, wdw, &spjcs, 0, 1);
		VTYPE	s = eng.forwardH1_wip();
		snprintf(nm, sizeof nm, "wip_%s_score", tag);
		w.put_int(nm, (int) s);
	    }
	    {
		restore();
		Mfile	mfd(sizeof(SKL));
		SimdAln2h1 eng((const Seq**) seqs, pwd, wdw, &spjcs, 0, 1, 0);
		VTYPE	s = eng.forwardH1_wip(&mfd);
		snprintf(nm, sizeof nm, "wip_%s_fwd_scr", tag);
		w.put_int(nm, (int) s);
		int	nrec = (int) mfd.size();
		SKL*	rec = (SKL*) mfd.flush();
		std::vector<int> v;
		for (int i = 0; i < nrec; ++i) { v.push_back(rec[i].m); v.push_back(rec[i].n); }
		snprintf(nm, sizeof nm, "wip_%s_fwd_skl", tag);
		w.put_i32(nm, v);
		delete[] rec;
	    }
	    for (size_t u = 0; u < udh_list.size(); ++u) {
const		int	n_im = udh_list[u];
		restore();
		stripe31((const Seq**) seqs, &wdw, alprm.sh);
const		int	mode = ((std::max(abs(wdw.lw), wdw.up) + wdw.width) < SHRT_MAX)? 2: 4;
		Dim10*	cpos = new Dim10[n_im + 1];
		for (int i = 0; i <= n_im; ++i)
		    for (int c = 0; c < 10; ++c) cpos[i][c] = end_of_ulk;
		SimdAln2h1 eng((const Seq**) seqs, pwd, wdw, &spjcs, 0, mode);
		VTYPE	s = eng.hirschbergH1_wip(cpos, n_im);
		snprintf(nm, sizeof nm, "wip_%s_udh%d_scr", tag, n_im);
		w.put_int(nm, (int) s);
		std::vector<int> v;
		for (int i = 0; i <= n_im; ++i)
		    for (int c = 0; c < 10; ++c) v.push_back(cpos[i][c]);
		snprintf(nm, sizeof nm, "wip_%s_udh%d_cpos", tag, n_im);
		w.put_i32(nm, v);
		std::vector<int> rngs = {a->left, a->right, b->left, b->right, mode};
		snprintf(nm, sizeof nm, "wip_%s_udh%d_rng", tag, n_im);
		w.put_i32(nm, rngs);
		delete[] cpos;
	    }
	}
	IntronPrm.nquant = nq0;
	static const int alg_order[] = {0, 2, 3, 6, 1};	// 6 = -A2 with the recursive switch (algmode.alg & 4);
	for (int ai = 0; ai < 5; ++ai) {		// -A1 last: it leaves state behind that changes later runs
	    const int alg = alg_order[ai];
	    if (!g_alg_list.empty() && std::find(g_alg_list.begin(), g_alg_list.end(), alg) == g_alg_list.end()) continue;

	    algmode.alg = alg;
	    restore();
	    // HomScoreH_ng under -A1 (forwardH1 without a Vmf) stops with SIGSEGV in the reference itself; the CLI
	    // never takes that path for proteins, so only the alignment is recorded for that mode
	    if (alg != 1) {
		VTYPE	hs = HomScoreH_ng((const Seq**) seqs, pwd);
		snprintf(nm, sizeof nm, "hom_scr_A%d", alg);
		w.put_int(nm, (int) hs);
	    }
	    restore();
	    Gsinfo	gsi;
	    gsi.skl = alignH_ng((const Seq**) seqs, pwd, &gsi);
	    snprintf(nm, sizeof nm, "aln_scr_A%d", alg);
	    w.put_int(nm, (int) gsi.scr);
	    snprintf(nm, sizeof nm, "aln_skl_A%d", alg);
	    w.put_i32(nm, skl2vec(gsi.skl));
	    if (gsi.skl && gsi.skl->n) {
		restore();
		VTYPE	rs = skl_rngH_ng((const Seq**) seqs, &gsi, pwd);
		snprintf(nm, sizeof nm, "rng_scr_A%d", alg);
		w.put_int(nm, (int) rs);
		std::vector<int> fs = {(int) gsi.fstat.mch, (int) gsi.fstat.mmc, (int) gsi.fstat.gap,
		    (int) gsi.fstat.unp, (int) gsi.fstat.val, gsi.noeij, alprm2.jneibr, (int) algmode.lsg};
		snprintf(nm, sizeof nm, "rng_fstat_A%d", alg);
		w.put_i32(nm, fs);
		std::vector<int> ej;
		if (gsi.eijnc) {
		    const EISCR* e = gsi.eijnc->begin();
		    for (int i = 0; i < gsi.eijnc->size(); ++i, ++e) {
			const int rec[21] = {e->left, e->right, e->rleft, e->rright, e->mch, e->mmc, e->gap, e->unp,
			    e->mch5, e->mmc5, e->gap5, e->unp5, e->mch3, e->mmc3, e->gap3, e->unp3, e->phs,
			    (int) e->escr, (int) e->iscr, (int) e->sig3, (int) e->sig5};
			ej.insert(ej.end(), rec, rec + 21);
		    }
		}
		snprintf(nm, sizeof nm, "rng_eij_A%d", alg);
		w.put_i32(nm, ej);
		// the exon-form report of this alignment (Gsinfo::ExonForm, sqpr.cc:820-996: the -O4 lines, the same numbers
		// the -O12 ExonRecord / GeneRecord files carry), and the few values of the run it reads besides the records
		if (alg == 2 || alg == 0) {
		    FILE* tf = tmpfile();
		    if (tf) {
			static bool out_ready = false;
			if (!out_ready) { (void) setup_output(EXN_FORM, 0, false); out_ready = true; }	// sets the printer's out_form (sqpr.cc:95-118)
			if (g_o12_mode) o12_write(gsi, seqs);
			else gsi.printgene(seqs, EXN_FORM, tf);
			const long len = ftell(tf);
			std::vector<unsigned char> txt(len > 0 ? len : 0);
			rewind(tf);
			if (len > 0 && fread(txt.data(), 1, len, tf) != (size_t) len) txt.clear();
			fclose(tf);
			snprintf(nm, sizeof nm, "rng_exn_A%d", alg);
			w.put(nm, 1, txt.data(), (int) txt.size());
			const Seq* gene = seqs[1];
			const Seq* qry = seqs[0];
			float scale = alprm.scale;
			if (gene->exin && gene->exin->fact) scale *= gene->exin->fact;
			int sbits; memcpy(&sbits, &scale, 4);
			int abits; { float as = alprm.scale; memcpy(&abits, &as, 4); }
			std::vector<int> ep = {sbits, abits, gene->SiteNo(0), gene->SiteNo(1), gene->len, (int) gene->inex.sens,
			    qry->SiteNo(0), qry->SiteNo(1), qry->len, (int) qry->inex.sens, qry->many, (int) gsi.scr,
			    qry->left, qry->right};
			snprintf(nm, sizeof nm, "rng_exnprm_A%d", alg);
			w.put_i32(nm, ep);
		    }
		}
		// the edit records behind the Cigar / Vulgar writers (fwd2h1.cc:663-667, 695-924; Vulgar after postproc)
		if (alg == 2 || alg == 0) {
		    const int keep_nsa = algmode.nsa;
		    for (int f = 0; f < 2; ++f) {
			algmode.nsa = f ? VLG_FORM : CIG_FORM;
			restore();
			Gsinfo	g2;
			g2.skl = gsi.skl;
			(void) skl_rngH_ng((const Seq**) seqs, &g2, pwd);
			std::vector<int> ops;
			if (!f && g2.cigar)
			    for (int i = 0; i < g2.cigar->size(); ++i) { ops.push_back(g2.cigar->rec[i].ope); ops.push_back(g2.cigar->rec[i].len); }
			if (f && g2.vlgar)
			    for (int i = 0; i < g2.vlgar->size(); ++i) {
				ops.push_back(g2.vlgar->rec[i].ope); ops.push_back(g2.vlgar->rec[i].alen); ops.push_back(g2.vlgar->rec[i].blen);
			    }
			snprintf(nm, sizeof nm, "rng_%s_A%d", f ? "vulgar" : "cigar", alg);
			w.put_i32(nm, ops);
			g2.skl = 0;
		    }
		    algmode.nsa = keep_nsa;
		}
	    }
	}
	if (g_o12_mode) o12_collect(w, seqs[1]);
	return 0;
}
