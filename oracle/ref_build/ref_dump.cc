// ref_dump -- golden-vector generator.  TEST INFRASTRUCTURE ONLY.
//
// Our own harness (no reference code in it) that links against the compiled
// reference objects (oracle/ref_build/Makefile) and drives the reference's
// public Aln2 surface (aln.h:348-357) and its SimdAln2s1 engine class
// (fwd2s1_simd.h:65-349) on one (genomic window, query) pair, then writes
//   * the complete read-only DP context the engines consumed (codes, splice
//     signals, substitution matrix, gap / intron parameters, band), and
//   * what the reference produced (raw engine scores, SKL corner lists, UDH
//     cpos rows, rescored totals)
// as a flat little-endian "SPDG" container that tests/ read with numpy.
//
// It runs only in the build container (needs /root/reference); the fixtures it
// writes are committed under tests/golden/ together with the driver script.
//
// usage: ref_dump [opts] genome.fa query.fa out.spdg
//   -l N     alprm.ls (2 affine, 3 double affine)      -w N   band shoulder alprm.sh
//   -L       local (-LS)                               -g ab  exg flags: 4 chars 0/1
//                                                              a.exgl a.exgr b.exgl b.exgr
//   -U N     forced #intermediates for UDH (alprm.ubh) -V N   MaxVmfSpace bytes
//   -q N     IntronPrm.nquant override                 -T dir species table (AlnParam not parsed)
//   -r al,ar,bl,br  restrict the active ranges (Seq::left/right) before the tables are built
//   -u list  extra explicit UDH runs with these n_im (comma separated)
//   -A list  only these engine selectors in the Aln2-surface section, and no engine-level section: fixtures beyond
//            1472 nt, where the reference's int16 engines (-A1..3) are erratic and -A0 is the truth
//   -X n     algmode.crs (-yX)                        -C     -LC (local, LocalC)
//   -I list  the query carries conserved intron positions (a SigII as a database entry with gene structure has it,
//            src/dbs.cc:780): Cip_score::cip_score(m) (src/gsinfo.cc:65-79) then gives every intron accepted in row m a
//            bonus under -A0 / -A1; -J w = alprm2.spb (the weight, as -yJ).  The bonus row is dumped as `cip`.
//   -B       the -O12 record files of the -A0 and -A2 alignments instead of their -O4 text (keys o12_grd / o12_erd / o12_qrd)
//   -Q n     seeded path (algmode.qck = n, 1..3): the HSPs of geneorient() as match_2 obtains them (spaln.cc:773-776) are
//            dumped as inputs, alignS_ng(seqs, pwd, gsi, 1) runs with seeding on, and every Wilip the walk constructs on a
//            sub-range (seededS_ng at the higher levels, fwd2s1.cc:2609) is recorded through the tap below
//   -O       alignS_ng(ori = 3) fixture: both strands prepared as spaln.cc:1137-1152 does (genomicseq, ori = 3),
//            the reverse-strand problem dumped under r_*, the result of alignS_ng(seqs, pwd, gsi, 3) under ori3_*

#include "ref_dump_common.h"
#include <algorithm>
#include <climits>
#include "fwd2s1_simd.h"

// ---- Wilip tap.  The Makefile links ref_dump against a copy of the reference's wln.o in which objcopy has renamed the
// two symbols of Wilip::Wilip(const Seq**, const PwdB*, int) (complete / base object constructor) -- the object code is
// the reference's, untouched.  The definition below takes the original name, calls the renamed original and, while the
// tap is on, records what the constructor produced: that is how a fixture carries the HSP units the reference's own
// seeded walk saw at its recursion levels.
extern "C" void ref_wilip_ctor(Wilip* self, const Seq** seqs, const PwdB* pwd, int level);
bool	g_o12_mode = false;
char	g_o12_prefix[256];
int	g_seeded_q = 0;
std::vector<int>	g_alg_list;
bool		wilip_tap_on = false;
std::vector<int>	wilip_tap_log;
Wilip::Wilip(const Seq* seqs[], const PwdB* pwd, const int level)
{
	ref_wilip_ctor(this, seqs, pwd, level);
	if (!wilip_tap_on) return;
	std::vector<int>& L = wilip_tap_log;
	// (+ 16 on the reverse-strand pass of alignS_ng(.., 3): the query has been reverse-complemented there)
	const int hd[6] = {level + (seqs[0]->inex.sens? 16: 0), seqs[0]->left, seqs[0]->right, seqs[1]->left, seqs[1]->right, wlu? nwlu: 0};
	L.insert(L.end(), hd, hd + 6);
	for (int u = 0; wlu && u < nwlu; ++u) {
	    const WLUNIT& x = wlu[u];
	    const int uh[6] = {x.num, x.nid, x.tlen, x.llmt, x.ulmt, (int) x.scr};
	    L.insert(L.end(), uh, uh + 6);
	    for (int j = 0; j <= x.num; ++j) {		// num HSPs + the slot behind them that seededS_ng overwrites
		const JUXT& t = x.jxt[j];
		const int jr[5] = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		L.insert(L.end(), jr, jr + 5);
	    }
	}
}


int main(int argc, const char** argv)
{
	int	ls = 2, sh = 100, local = 0, ubh = 0, nquant = 0, ori3 = 0, seeded_q = 0, crs = -1;
	long	vmfspace = 0;
const	char*	exg = 0;
	std::vector<int>	udh_list, alg_list;
	int	rng4[4] = {-1, -1, -1, -1};
	std::vector<int>	intron_pos;
	float	spb = 0;
	int	ai = 1;
	for ( ; ai < argc && argv[ai][0] == '-'; ++ai) {
	    switch (argv[ai][1]) {
		case 'l': ls = atoi(argv[++ai]); break;
		case 'w': sh = atoi(argv[++ai]); break;
		case 'L': local = 1; break;
		case 'O': ori3 = 1; break;
		case 'Q': seeded_q = atoi(argv[++ai]) & 3; break;
		case 'B': g_o12_mode = true; break;
		case 'I': {			// query positions that carry a conserved intron (SigII of the query), -J weight
		    const char* p = argv[++ai];
		    while (*p) {
			intron_pos.push_back(atoi(p));
			while (*p && *p != ',') ++p;
			if (*p == ',') ++p;
		    }
		    break;
		}
		case 'J': spb = atof(argv[++ai]); break;
		case 'X': crs = atoi(argv[++ai]); break;	// algmode.crs as -yX sets it (simmtx.cc:704): 0 = same species
		case 'b': bpprm.factor = atof(argv[++ai]); break;	// -yB: weight of the branch-point signal (simmtx.cc:671)
		case 'D': bpprm.maxb3d = atoi(argv[++ai]); break;	// -yD: furthest branch point from its acceptor
		case 'C': local = 3; break;			// -LC: local with LocalC (algmode.lcl & 32)
		case 'A': {
		    const char* p = argv[++ai];
		    while (*p) {
			alg_list.push_back(atoi(p));
			while (*p && *p != ',') ++p;
			if (*p == ',') ++p;
		    }
		    break;
		}
		case 'g': exg = argv[++ai]; break;
		case 'U': ubh = atoi(argv[++ai]); break;
		case 'T': ftable.setpath(argv[++ai], gnm2tab); break;	// species-specific tables, as spaln -T (spaln.cc:484-487)
		case 'V': vmfspace = atol(argv[++ai]); break;
		case 'q': nquant = atoi(argv[++ai]); break;
		case 'r': sscanf(argv[++ai], "%d,%d,%d,%d", rng4, rng4 + 1, rng4 + 2, rng4 + 3); break;
		case 'u': {
		    const char* p = argv[++ai];
		    while (*p) {
			udh_list.push_back(atoi(p));
			while (*p && *p != ',') ++p;
			if (*p == ',') ++p;
		    }
		    break;
		}
		default: fprintf(stderr, "bad option %s\n", argv[ai]); return 1;
	    }
	}
	if (argc - ai < 3) {
	    fprintf(stderr, "usage: ref_dump [opts] genome.fa query.fa out.spdg\n");
	    return 1;
	}
const	char*	files[2] = {argv[ai], argv[ai + 1]};
const	char*	outfn = argv[ai + 2];

	set_default_params();
	optimize(GLOBAL, MAXIMUM);
	algmode.qck = 0;		// -Q0: whole window through lspS_ng
	algmode.blk = 0;
	alprm.ls = ls;
	alprm.sh = sh;
	alprm.ubh = ubh;
	if (local) algmode.lcl |= 16;
	if (local == 3) algmode.lcl |= 32;
	if (crs >= 0) algmode.crs = crs;
	if (vmfspace) MaxVmfSpace = (int) vmfspace;
	if (nquant) IntronPrm.nquant = nquant;
	OutPrm.all_out = 1;

	Seq*	seqs[4];
	initseq(seqs, 4);
	Seq*&	a = seqs[0];
	Seq*&	b = seqs[1];
	SeqServer	svr(2, files, IM_SNGL, 0, UNKNOWN, UNKNOWN);
	if (svr.nextseq(b, 1) == IS_END) { fprintf(stderr, "no genome\n"); return 1; }
	if (svr.nextseq(a, 0) != IS_OK) { fprintf(stderr, "no query\n"); return 1; }
	if (rng4[0] >= 0) { a->left = rng4[0]; a->right = rng4[1]; b->left = rng4[2]; b->right = rng4[3]; }
	if (g_o12_mode) o12_begin();
	g_seeded_q = seeded_q;
	g_alg_list = alg_list;
	if (a->isprotein()) return dump_protein(seqs, exg, udh_list, outfn);
	b->inex.intr = algmode.lsg;
	makeWlprms(prePwd((const Seq**) seqs));
	algmode.alg = 2;		// IntronPenalty builds the quantile table qm only when alg > 1 (codepot.cc:162)
	PwdB*	pwd = new PwdB((const Seq**) seqs);
	makeStdSig53();
	a->inex.intr = 0;
	a->inex.ori = 1;
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
	b->exin = new Exinon(b, pwd, false);
	if (!intron_pos.empty()) {
	    if (spb > 0) alprm2.spb = spb;
	    (void) spb_fact();
	    a->sigII = new SigII(intron_pos.data(), (int) intron_pos.size(), 1);
	}
	if (seeded_q) {
	    // the two-sequence set-up of match_2 (spaln.cc:742-776): the reverse strand beside the forward one, HSPs of
	    // the lowest level that finds any from geneorient()
	    algmode.qck = seeded_q;
	    Seq* const	b0 = b;
	    if (ori3) {			// both strands as genomicseq prepares them for ori = 3 (spaln.cc:1137-1152)
		a->inex.ori = 3;
		delete b->exin;
		b->exin = new Exinon(b, pwd, true);
	    }
	    b->comrev(seqs + 2);
	    if (ori3) seqs[2]->setanti(seqs + 1);
	    seqs[2]->exin = new Exinon(seqs[2], pwd, ori3 != 0);
	    const int np = geneorient(seqs, pwd);
	    if (b != b0) { fprintf(stderr, "ref_dump -Q: the reverse strand won geneorient(); not a fixture\n"); return 3; }
	    if (np == 0) fprintf(stderr, "ref_dump -Q: no HSP at any level\n");
	}

	Writer	w(outfn);
// ---- inputs: everything the engines read from one strand's (query, genome) pair; `pre` = "" for the
//      pair as given, "r_" for the reverse strand (comrev(a) + antiseq(b))
	auto dump_strand = [&](const char* pre) {
	    char	pnb[40];
	    auto pn = [&](const char* nm_) { snprintf(pnb, sizeof pnb, "%s%s", pre, nm_); return (const char*) pnb; };
	w.put(pn("a_codes"), 1, a->at(0), a->len);
	w.put(pn("b_codes"), 1, b->at(0), b->len);
	{
	    std::vector<short>	s5(b->len + 1, 0), s3(b->len + 1, 0);
	    std::vector<signed char> p5(b->len + 1, -2), p3(b->len + 1, -2);
	    std::vector<unsigned char> c5(b->len + 1, 0), c3(b->len + 1, 0);
	    // NB: the acceptor signal of the boundary column (n = b->left when the window starts at 0) comes out of
	    // the pattern scan's reach in front of the sequence (PatMat::calcPatMat from sd->left - 1 on,
	    // codepot.cc:481-486): indeterminate memory in the reference, a few units different from run to run.
	    // The engines READ it (s1_cut_left changes its alignment if it is forced to 0), so it is dumped as this
	    // run saw it: inputs and outputs of a fixture belong to one run; regenerating a fixture may change
	    // this one input element (and nothing else).
	    for (int n = b->left; n <= b->right; ++n) {
		const SGPT2* sg = b->exin->score_n(n);
		s5[n] = sg->sig5; s3[n] = sg->sig3;
		p5[n] = sg->phs5; p3[n] = sg->phs3;
		c5[n] = b->exin->isDonor(n);
		c3[n] = b->exin->isAccpt(n);
	    }
	    w.put(pn("sig5"), 2, s5.data(), s5.size());
	    w.put(pn("sig3"), 2, s3.data(), s3.size());
	    w.put(pn("phs5"), 4, p5.data(), p5.size());
	    w.put(pn("phs3"), 4, p3.data(), p3.size());
	    w.put(pn("cano5"), 1, c5.data(), c5.size());
	    w.put(pn("cano3"), 1, c3.data(), c3.size());
	    // dinucleotide classes exactly as Exinon::intron53_c assigns them (codepot.cc:435-448),
	    // and the junction table behind Exinon::sig53(m, n, IE53) (codepot.cc:411-415):
	    //   sig53(m, n, IE53) = sig3[n] + T53[16 * dinc5[m] + dinc3[n]]
	    std::vector<unsigned char> d5(b->len + 3, 0), d3(b->len + 3, 0);
	    int	nc = 1;
	    for (int i = b->left; i < b->right; ++i) {
		int c = ncredctab[*b->at(i)];
		if (c >= 4) c = 1;
		nc = ((nc << 2) + c) & 0xf;
		if (i - 1 >= 0) d5[i - 1] = nc;
		d3[i + 1] = nc;
	    }
	    w.put(pn("dinc5"), 1, d5.data(), b->len + 1);
	    w.put(pn("dinc3"), 1, d3.data(), b->len + 1);
	    std::vector<int> t53(256, 0), mrep(16, -1), nrep(16, -1);
	    for (int n = b->left; n <= b->right; ++n) {
		if (n >= b->left && n < b->right - 1 && mrep[d5[n]] < 0 && n >= b->left) mrep[d5[n]] = n;
		if (n >= b->left + 2 && nrep[d3[n]] < 0) nrep[d3[n]] = n;
	    }
	    for (int u = 0; u < 16; ++u)
		for (int v = 0; v < 16; ++v)
		    if (mrep[u] >= 0 && nrep[v] >= 0)
			t53[16 * u + v] = b->exin->sig53(mrep[u], nrep[v], IE53) - b->exin->score_n(nrep[v])->sig3;
	    w.put_i32(pn("t53"), t53);
	    int	bad = 0;
	    for (int t = 0; t < 400; ++t) {		// self-check of the restated classes
		int m = b->left + (t * 7919) % std::max(1, b->right - b->left - 2);
		int n = b->left + 2 + (t * 104729) % std::max(1, b->right - b->left - 2);
		int want = b->exin->sig53(m, n, IE53);
		int got = b->exin->score_n(n)->sig3 + t53[16 * d5[m] + d3[n]];
		if (want != got) ++bad;
	    }
	    if (bad) fprintf(stderr, "ref_dump: %d sig53 self-check mismatches\n", bad);
	}
	    std::vector<int> rg = {a->left, a->right, b->left, b->right,
		(int) a->inex.exgl, (int) a->inex.exgr, (int) b->inex.exgl, (int) b->inex.exgr, (int) a->inex.sens};
	    w.put_i32(pn("ranges"), rg);
	};
	if (ori3 && !seeded_q) {
	    // both strands as the caller of alignS_ng prepares them (genomicseq with ori = 3, spaln.cc:1137-1152)
	    delete b->exin;
	    b->exin = new Exinon(b, pwd, true);
	    b->comrev(seqs + 2);
	    seqs[2]->setanti(seqs + 1);
	    seqs[2]->exin = new Exinon(seqs[2], pwd, true);
	}
	dump_strand("");
	{
	    // the splice-signal model behind sig5 / sig3 (Exinon::intron53_n, codepot.cc:479-520): the two position weight
	    // matrices (PatMat, utilseq.h:64-88) as loaded from the parameter tables, the per-dinucleotide terms and the
	    // scale, so that the signal arrays above can be recomputed from b_codes
	    auto dump_pm = [&](const char* tag, const PatMat* pm) {
		char nmb[40];
		if (!pm) return;
		std::vector<int> hd = {pm->rows, pm->cols, pm->offset, pm->order(), pm->nalpha};
		snprintf(nmb, sizeof nmb, "%s_hdr", tag); w.put_i32(nmb, hd);
		std::vector<int> fl(2 + pm->rows * pm->cols);
		memcpy(&fl[0], &pm->tonic, 4); memcpy(&fl[1], &pm->min_elem, 4);
		memcpy(&fl[2], pm->mtx, sizeof(float) * pm->rows * pm->cols);
		snprintf(nmb, sizeof nmb, "%s_f32", tag); w.put_i32(nmb, fl);	// float bit patterns: tonic, min_elem, mtx
	    };
	    dump_pm("pm5", pwd->eijpat->pattern5);
	    dump_pm("pm3", pwd->eijpat->pattern3);
	    const float fs = (float) b->exin->fS * alprm2.sss;		// intron53_n: fs = fS * alprm2.sss (codepot.cc:496)
	    std::vector<int> sm(4);
	    memcpy(&sm[0], &fs, 4);
	    sm[1] = (int) algmode.any; sm[2] = (int) b->many;
	    sm[3] = ori3 ? 1 : 0;					// the both_ori argument Exinon was built with above
	    // the per-dinucleotide terms sig53tab[0 / 1][class] (private to Exinon): what is left of a signal after the
	    // scaled matrix score, which the reference's own PatMat::calcPatMat reproduces (same call, same range as
	    // intron53_n: codepot.cc:481-486)
	    std::vector<int> tab(32, INT_MIN);
	    if (pwd->eijpat->pattern5 && pwd->eijpat->pattern3) {
		--b->left; ++b->right;
		float* p5 = pwd->eijpat->pattern5->calcPatMat(b);
		float* p3 = pwd->eijpat->pattern3->calcPatMat(b);
		++b->left; --b->right;
		std::vector<unsigned char> e5(b->len + 3, 0), e3(b->len + 3, 0);
		int	nc2 = 1;
		for (int i = b->left; i < b->right; ++i) {
		    int c = ncredctab[*b->at(i)];
		    if (c >= 4) c = 1;
		    nc2 = ((nc2 << 2) + c) & 0xf;
		    if (i - 1 >= 0) e5[i - 1] = nc2;
		    e3[i + 1] = nc2;
		}
		int	clash = 0;
		for (int n = b->left + 2; n < b->right - 1; ++n) {
		    const SGPT2* sg = b->exin->score_n(n);
		    const int t5 = sg->sig5 - (STYPE) (fs * p5[n - b->left + 1]);
		    const int t3 = sg->sig3 - (STYPE) (fs * p3[n - b->left + 1]);
		    if (tab[e5[n]] == INT_MIN) tab[e5[n]] = t5; else if (tab[e5[n]] != t5) ++clash;
		    if (tab[16 + e3[n]] == INT_MIN) tab[16 + e3[n]] = t3; else if (tab[16 + e3[n]] != t3) ++clash;
		}
		if (clash) fprintf(stderr, "ref_dump: %d signal-table clashes\n", clash);
		delete[] p5; delete[] p3;
	    }
	    w.put_i32("sig53tab01", tab);
	    w.put_i32("sigmodel", sm);
	}
	if (ori3) {
	    a->comrev();
	    antiseq(seqs + 1);
	    dump_strand("r_");
	    a->comrev();
	    antiseq(seqs + 1);
	    a->inex.sens = 0;			// (comrev toggles it; the pair is back as given)
	}
	{
	    const Simmtx* sm = pwd->simmtx;
	    std::vector<int>	mtx(sm->dim * sm->dim);
	    for (int i = 0; i < sm->dim; ++i)
		for (int j = 0; j < sm->dim; ++j) mtx[i * sm->dim + j] = sm->mtx[i][j];
	    w.put_int("mtx_dim", sm->dim);
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
	    std::vector<int>	ql, qp;
	    for (int j = 0; j < IntronPrm.nquant; ++j) {
		ql.push_back(pwd->IntPen->qm[j].len);
		qp.push_back(pwd->IntPen->qm[j].pen);
	    }
	    w.put_i32("qm_len", ql);
	    w.put_i32("qm_pen", qp);
	    // exact intron-length penalty materialised for every length that can occur
	    std::vector<short>	ip(b->len + 2);
	    for (int l = 0; l < (int) ip.size(); ++l)
		ip[l] = pwd->IntPen->Penalty(l);
	    w.put("intpen", 2, ip.data(), ip.size());
	    if (a->sigII) {
		Cip_score	cs(a);
		std::vector<int>	cv(a->len + 1);
		for (int m = 0; m <= a->len; ++m) cv[m] = (int) cs.cip_score(m);
		w.put_i32("cip", cv);
	    }
	}

// ---- reference results
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

	if (seeded_q) {
	    // alignS_ng with seeding on (fwd2s1.cc:2746 -> globalS_ng :2674 -> seededS_ng :2587 -> interpolateS :2405)
	    std::vector<int> jx;
	    for (int j = 0; b->jxt && j <= b->CdsNo; ++j) {
		const JUXT& t = b->jxt[j];
		const int jr[5] = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		jx.insert(jx.end(), jr, jr + 5);
	    }
	    w.put_i32("seed_jxt", jx);		// CdsNo HSPs + the slot behind them
	    dump_wilip_model(w, pwd);
	    float smn4 = getsmn(4), w2 = alprm2.w, maxsp = alprm.maxsp;
	    int smn4b, w2b, maxspb;
	    memcpy(&smn4b, &smn4, 4); memcpy(&w2b, &w2, 4); memcpy(&maxspb, &maxsp, 4);
	    std::vector<int> sp = {(int) algmode.qck, b->wllvl, b->jxt? b->CdsNo: 0,
		(int) setwlprm(0)->width, (int) setwlprm(1)->width, (int) setwlprm(2)->width, (int) setwlprm(3)->width,
		IntronPrm.elmt, IntronPrm.minl, IntronPrm.tlmt, (int) pwd->Vthr, alprm2.desert, maxspb, (int) algmode.crs,
		smn4b, w2b, (int) b->exin->gc_sig5, (int) algmode.lcl, (int) a->inex.ori, (int) b->exin->at_sig5};
	    w.put_i32("seed_params", sp);
	    // the walk edits its inputs (phs5 / phs3 of the junctions indelfreespjS accepts, :2055-2059; the HSP list's last
	    // slot, :2633, 2667): every run starts from the same state
	    std::vector<SGPT2> sg0(b->right - b->left + 1);
	    for (int n = b->left; n <= b->right; ++n) sg0[n - b->left] = *b->exin->score_n(n);
	    std::vector<JUXT> jx0(b->jxt, b->jxt + (b->jxt? b->CdsNo + 1: 0));
	    std::vector<SGPT2> sgr0;
	    if (ori3)
		for (int n = seqs[2]->left; n <= seqs[2]->right; ++n) sgr0.push_back(*seqs[2]->exin->score_n(n));
	    if (alg_list.empty()) { alg_list.push_back(0); alg_list.push_back(2); }
	    for (size_t k = 0; k < alg_list.size(); ++k) {
const		int	alg = alg_list[k];
		algmode.alg = alg;
		restore();
		for (int n = b->left; n <= b->right; ++n) *b->exin->score_n(n) = sg0[n - b->left];
		if (b->jxt) vcopy(b->jxt, jx0.data(), jx0.size());
		wilip_tap_log.clear();
		wilip_tap_on = true;
		Gsinfo	gsi;
		if (ori3)			// the reverse strand's phase marks start clean as well
		    for (int n = seqs[2]->left; n <= seqs[2]->right; ++n) *seqs[2]->exin->score_n(n) = sgr0[n - seqs[2]->left];
		gsi.skl = alignS_ng(seqs, pwd, &gsi, ori3? 3: 1);
		wilip_tap_on = false;
		if (ori3) {
const		    int	rev = a->inex.sens? 1: 0;
		    snprintf(nm, sizeof nm, "seed_rev_A%d", alg);
		    w.put_int(nm, rev);
		    if (rev) { a->comrev(); antiseq(seqs + 1); }	// back to the strand as given
		}
		snprintf(nm, sizeof nm, "seed_scr_A%d", alg);
		w.put_int(nm, (int) gsi.scr);
		snprintf(nm, sizeof nm, "seed_skl_A%d", alg);
		w.put_i32(nm, skl2vec(gsi.skl));
		snprintf(nm, sizeof nm, "seed_wilip_A%d", alg);
		w.put_i32(nm, wilip_tap_log);
		if (gsi.skl && gsi.skl->n && !ori3) {
		    restore();
		    for (int n = b->left; n <= b->right; ++n) *b->exin->score_n(n) = sg0[n - b->left];
		    VTYPE	rs = skl_rngS_ng((const Seq**) seqs, &gsi, pwd);
		    snprintf(nm, sizeof nm, "seed_rng_scr_A%d", alg);
		    w.put_int(nm, (int) rs);
		}
	    }
	    return 0;
	}

	if (ori3) {
	    // alignS_ng with the default orientation handling (src/fwd2s1.cc:2746-2778 -> infer_orientation :2718)
	    for (int alg = 2; alg >= 0; alg -= 2) {	// -A2 (the _wip engines), then -A0
		algmode.alg = alg;
		restore();
		Gsinfo	gsi;
		gsi.skl = alignS_ng(seqs, pwd, &gsi, 3);
const		int	rev = a->inex.sens? 1: 0;
		snprintf(nm, sizeof nm, "ori3_rev_A%d", alg);
		w.put_int(nm, rev);
		snprintf(nm, sizeof nm, "ori3_scr_A%d", alg);
		w.put_int(nm, (int) gsi.scr);
		snprintf(nm, sizeof nm, "ori3_skl_A%d", alg);
		w.put_i32(nm, skl2vec(gsi.skl));
		if (rev) { a->comrev(); antiseq(seqs + 1); }	// back to the strand as given
	    }
	    return 0;
	}

// (1) engine-level goldens straight from SimdAln2s1, the _wip engines
//     (fwd2s1_wip_simd.h:42,233,476).  tag qn: nquant as configured (what -A2
//     runs), tag q1: nquant = 1 (what -A3 runs, fwd2s1.cc:125).
	for (int pass = 0; pass < 2 && alg_list.empty(); ++pass) {
	    IntronPrm.nquant = pass? 1: nq0;
const	    char*	tag = pass? "q1": "qn";
	    SpJunc	spjcs(b, pwd);
	    WINDOW	wdw;
	    restore();
	    stripe((const Seq**) seqs, &wdw, alprm.sh);
	    if (pass == 0) {
		std::vector<int> wv = {wdw.lw, wdw.up, wdw.width};
		w.put_i32("wdw", wv);
	    }
	    {	// score only (mode 1 as HomScoreS_ng passes for simd > 1, fwd2s1.cc:2709)
		SimdAln2s1 eng((const Seq**) seqs, pwd, wdw, &spjcs, 0, 1);
		VTYPE	s = eng.scoreonlyS1_wip();
		snprintf(nm, sizeof nm, "wip_%s_score", tag);
		w.put_int(nm, (int) s);
	    }
	    {	// forward + bitmap traceback (mode 1, trcbkalignS_ng fwd2s1.cc:1682-1684)
		restore();
		Mfile	mfd(sizeof(SKL));
		SimdAln2s1 eng((const Seq**) seqs, pwd, wdw, &spjcs, 0, 1, 0);
		VTYPE	s = eng.forwardS1_wip(&mfd);
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
	    for (size_t u = 0; u < udh_list.size(); ++u) {	// UDH with explicit n_im
const		int	n_im = udh_list[u];
		restore();
		stripe((const Seq**) seqs, &wdw, alprm.sh);
const		int	mode = ((std::max(abs(wdw.lw), wdw.up) + wdw.width) < SHRT_MAX)? 2: 4;	// fwd2s1.cc:1867
		Dim10*	cpos = new Dim10[n_im + 1];
		for (int i = 0; i <= n_im; ++i) {
		    for (int c = 0; c < 10; ++c) cpos[i][c] = 0;
		    cpos[i][0] = cpos[i][2] = end_of_ulk;
		}
		SimdAln2s1 eng((const Seq**) seqs, pwd, wdw, &spjcs, 0, mode);
		VTYPE	s = eng.hirschbergS1_wip(cpos, n_im);
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

// (2) the Aln2 surface per engine selector -A0..3 (algmode.alg & 3, fwd2s1.cc:112).
//     -A3 goes last: its Aln2s1 ctor sets IntronPrm.nquant = 1 for good.
	for (int alg = 0; alg < 7; ++alg) {
	    if (alg == 4 || alg == 5) continue;	// 6 = -A2 with the recursive switch (algmode.alg & 4)
	    if (!alg_list.empty() && std::find(alg_list.begin(), alg_list.end(), alg) == alg_list.end()) continue;
	    algmode.alg = alg;
	    restore();
	    VTYPE	hs = HomScoreS_ng((const Seq**) seqs, pwd);
	    snprintf(nm, sizeof nm, "hom_scr_A%d", alg);
	    w.put_int(nm, (int) hs);
	    restore();
	    Gsinfo	gsi;
	    gsi.skl = alignS_ng(seqs, pwd, &gsi, 1);
	    snprintf(nm, sizeof nm, "aln_scr_A%d", alg);
	    w.put_int(nm, (int) gsi.scr);
	    snprintf(nm, sizeof nm, "aln_skl_A%d", alg);
	    w.put_i32(nm, skl2vec(gsi.skl));
	    if (gsi.skl && gsi.skl->n) {
		restore();
		VTYPE	rs = skl_rngS_ng((const Seq**) seqs, &gsi, pwd);
		snprintf(nm, sizeof nm, "rng_scr_A%d", alg);
		w.put_int(nm, (int) rs);
		// what the rescoring leaves in Gsinfo: alignment statistics and the per-exon records
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
		// the edit records skl_rngS_ng builds for the Cigar / Vulgar / SAM writers (fwd2s1.cc:469-475, 509-689),
		// one run per format (algmode.nsa selects which record file exists); -A2 and -A0 only
		if (alg == 2 || alg == 0) {
		    const int keep_nsa = algmode.nsa;
		    const int fmts[3] = {CIG_FORM, VLG_FORM, SAM_FORM};
		    const char* tag[3] = {"cigar", "vulgar", "sam"};
		    for (int f = 0; f < 3; ++f) {
			algmode.nsa = fmts[f];
			restore();
			Gsinfo	g2;
			g2.skl = gsi.skl;
			(void) skl_rngS_ng((const Seq**) seqs, &g2, pwd);
			std::vector<int> ops;
			if (f == 0 && g2.cigar)
			    for (int i = 0; i < g2.cigar->size(); ++i) { ops.push_back(g2.cigar->rec[i].ope); ops.push_back(g2.cigar->rec[i].len); }
			if (f == 1 && g2.vlgar)
			    for (int i = 0; i < g2.vlgar->size(); ++i) {
				ops.push_back(g2.vlgar->rec[i].ope); ops.push_back(g2.vlgar->rec[i].alen); ops.push_back(g2.vlgar->rec[i].blen);
			    }
			if (f == 2 && g2.samfm) {
			    for (int i = 0; i < g2.samfm->size(); ++i) { ops.push_back(g2.samfm->rec[i].ope); ops.push_back(g2.samfm->rec[i].len); }
			    std::vector<int> hd = {g2.samfm->flag, g2.samfm->pos, g2.samfm->mapq, g2.samfm->left, g2.samfm->right};
			    snprintf(nm, sizeof nm, "rng_samhdr_A%d", alg);
			    w.put_i32(nm, hd);
			}
			snprintf(nm, sizeof nm, "rng_%s_A%d", tag[f], alg);
			w.put_i32(nm, ops);
			g2.skl = 0;				// (owned by gsi)
		    }
		    algmode.nsa = keep_nsa;
		}
	    }
	}
	if (g_o12_mode) o12_collect(w, seqs[1]);
	return 0;
}
