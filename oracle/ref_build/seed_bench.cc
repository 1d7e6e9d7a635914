// seed_bench -- the seeded path (-Q5 .. -Q7) of a BATCH of (window, query) pairs: the compiled reference on T host
// threads against the library's spdp_align_s_seeded / spdp_align_h_seeded behind the binding of INTEGRATION.md.
// TEST / BENCH INFRASTRUCTURE ONLY (built by oracle/ref_build/Makefile into oracle/_ref/, runs on a GPU box); our own
// driver, no reference code in it.
//
// Per pair the set-up of match_2 (src/spaln.cc:734-790): exg_seq, Exinon on both strands, geneorient() for the HSPs
// (timed as "prep"; both contenders start behind it).  Then
//   reference: alignS_ng(seqs, pwd, gsi, 1) / alignH_ng on T threads, pairs from a shared counter;
//   library:   ONE spdp_align_*_seeded call on the whole batch; the reference's own Wilip answers the recursion
//              levels through SpdpHspSource (called from the walks' threads), as a maintainer's shim would have it.
// Both start from the same state (phase marks and HSP lists restored in between), results are compared pair by pair.
// Prints one line: pairs, compared, identical, prep s, reference s (T threads), library s (second call; the first, cold
// one beside it), marshalling s, and the
// library's counters (device batches, lsp calls, tracebacks, cut ranges, Wilip calls).
//
// usage: seed_bench -Q n [-A alg] [-t threads] [-X crs] list.txt      (list.txt: one "window.fa query.fa" per line)
#include "ref_dump_common.h"
#include "shim_fill.h"		// integration/shim_fill.h: the reference-side binding under test
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

struct Pair {
	Seq*	seqs[4];
	RANGE	ra, rb;
	INEX	ia, ib;
	int	exin_left = 0, exin_right = 0;
	std::vector<SGPT2>	sg2;
	std::vector<SGPT6>	sg6;
	std::vector<JUXT>	jx0;
	bool	skip = false;
	int	ref_scr = 0;
	std::vector<SKL>	ref_skl;
	std::vector<int16_t>	s5, s3;
	SeedCols	c;
	HCols	hc;
	std::vector<SpdpJuxt>	jx;
};

struct BatchSrc { std::vector<Pair*> of_query; const PwdB* pwd; };

static int batch_units(void* user, int32_t query, int32_t level, const int32_t span[8], const int32_t** flat, int32_t* n_flat)
{
	BatchSrc* S = (BatchSrc*) user;
	Pair&	P = *S->of_query[query];
	wilip_flat(P.seqs, S->pwd, level, span, P.c.flat);
	*flat = P.c.flat.data(); *n_flat = (int32_t) P.c.flat.size();
	return 0;
}

static double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F> static void on_threads(int nthr, int n, F f)
{
	std::atomic<int>	next(0);
	std::vector<std::thread> pool;
	for (int t = 0; t < nthr; ++t)
	    pool.emplace_back([&]() { for (;;) { const int k = next.fetch_add(1); if (k >= n) break; f(k); } });
	for (auto& t : pool) t.join();
}

int main(int argc, const char** argv)
{
	int	seeded_q = 3, alg = 2, nthr = 1, crs = -1;
	int	ai = 1;
	for ( ; ai < argc && argv[ai][0] == '-'; ++ai) {
	    const char c = argv[ai][1];
	    if (c == 'Q') seeded_q = atoi(argv[++ai]) & 3;
	    else if (c == 'A') alg = atoi(argv[++ai]);
	    else if (c == 't') nthr = atoi(argv[++ai]);
	    else if (c == 'X') crs = atoi(argv[++ai]);
	    else { fprintf(stderr, "bad option %s\n", argv[ai]); return 2; }
	}
	if (ai >= argc || !seeded_q) { fprintf(stderr, "usage: seed_bench -Q n [-A alg] [-t threads] [-X crs] list.txt\n"); return 2; }
	std::vector<std::pair<std::string, std::string>> files;
	{
	    FILE* f = fopen(argv[ai], "r");
	    if (!f) { perror(argv[ai]); return 2; }
	    char g[1024], q[1024];
	    while (fscanf(f, "%1023s %1023s", g, q) == 2) files.push_back({g, q});
	    fclose(f);
	}
const	int	N = (int) files.size();
	if (!N) return 2;
	SpdpContext* ctx = spdp_create(0);
	if (!ctx) { fprintf(stderr, "seed_bench: no HIP device\n"); return 3; }
	set_default_params();
	optimize(GLOBAL, MAXIMUM);
	algmode.qck = 0;
	algmode.blk = 0;
	alprm.ls = 2;
	if (crs >= 0) algmode.crs = crs;
	OutPrm.all_out = 1;
	PwdB*	pwd = 0;
	bool	protein = false;
	{
	    Seq*	seqs[4];
	    initseq(seqs, 4);
const	    char*	fl[2] = {files[0].first.c_str(), files[0].second.c_str()};
	    SeqServer	svr(2, fl, IM_SNGL, 0, UNKNOWN, UNKNOWN);
	    if (svr.nextseq(seqs[1], 1) == IS_END || svr.nextseq(seqs[0], 0) != IS_OK) return 2;
	    protein = seqs[0]->isprotein();
	    seqs[1]->inex.intr = algmode.lsg;
	    makeWlprms(prePwd((const Seq**) seqs));
	    algmode.alg = 2;		// (the quantile table of the intron penalty exists only when alg > 1 at this point)
	    pwd = new PwdB((const Seq**) seqs);
	    makeStdSig53();
	    algmode.alg = alg;
	    if (alg == 3) IntronPrm.nquant = 1;
	    clearseq(seqs, 4);
	}
	algmode.qck = seeded_q;

	// ---- prep: what match_2 does before the aligner ------------------------------------------------------------------
	std::vector<Pair> pairs(N);
const	double	t_prep0 = now_s();
	on_threads(nthr, N, [&](int k) {
	    Pair&	P = pairs[k];
	    initseq(P.seqs, 4);
	    Seq*&	a = P.seqs[0];
	    Seq*&	b = P.seqs[1];
const	    char*	fl[2] = {files[k].first.c_str(), files[k].second.c_str()};
	    SeqServer	svr(2, fl, IM_SNGL, 0, UNKNOWN, UNKNOWN);
	    if (svr.nextseq(b, 1) == IS_END || svr.nextseq(a, 0) != IS_OK) { P.skip = true; return; }
	    b->inex.intr = algmode.lsg;
	    a->inex.intr = 0;
	    if (!protein) a->inex.ori = 1;
	    if (protein) { b->comrev(P.seqs + 2); b->nuc2tron(); }
	    P.exin_left = b->left; P.exin_right = b->right;
	    if (protein) {
		b->exin = new Exinon(b, pwd, false);
		P.seqs[2]->nuc2tron(); P.seqs[2]->exin = new Exinon(P.seqs[2], pwd, false);
	    }
	    a->exg_seq(algmode.lcl & 4, algmode.lcl & 8);
	    b->exg_seq(algmode.lcl & 1, algmode.lcl & 2);
	    if (!protein) {
		b->exin = new Exinon(b, pwd, false);
		b->comrev(P.seqs + 2);
		P.seqs[2]->exin = new Exinon(P.seqs[2], pwd, false);
	    }
	    P.ra = {a->left, a->right}; P.rb = {b->left, b->right};
	    P.ia = a->inex; P.ib = b->inex;
	    Seq* const	b0 = b;
	    (void) geneorient(P.seqs, pwd);
	    if (b != b0) { P.skip = true; return; }		// the other strand won: not the planted one, left out of both runs
	    if (protein) for (int n = std::max(0, b->left - 1); n <= b->right + 1; ++n) P.sg6.push_back(*b->exin->score_p(n));
	    else for (int n = b->left; n <= b->right; ++n) P.sg2.push_back(*b->exin->score_n(n));
	    if (b->jxt) P.jx0.assign(b->jxt, b->jxt + b->CdsNo + 1);
	});
const	double	t_prep = now_s() - t_prep0;
	std::vector<int> live;
	for (int k = 0; k < N; ++k) if (!pairs[k].skip) live.push_back(k);
const	int	M = (int) live.size();
	if (!M) { fprintf(stderr, "seed_bench: no pair left\n"); return 2; }

	// ---- the reference -------------------------------------------------------------------------------------------------
const	double	t_ref0 = now_s();
	on_threads(nthr, M, [&](int i) {
	    Pair&	P = pairs[live[i]];
	    Gsinfo	gsi;
	    gsi.skl = protein? alignH_ng((const Seq**) P.seqs, pwd, &gsi): alignS_ng(P.seqs, pwd, &gsi, 1);
	    P.ref_scr = (int) gsi.scr;
	    if (gsi.skl) P.ref_skl.assign(gsi.skl, gsi.skl + gsi.skl->n + 1);
	});
const	double	t_ref = now_s() - t_ref0;
	for (int i = 0; i < M; ++i) {			// the walk edits ranges, phase marks and the slot behind the HSP list
	    Pair&	P = pairs[live[i]];
	    Seq*	a = P.seqs[0];
	    Seq*	b = P.seqs[1];
	    a->left = P.ra.left; a->right = P.ra.right; b->left = P.rb.left; b->right = P.rb.right;
	    a->inex = P.ia; b->inex = P.ib;
	    if (protein) for (int n = std::max(0, b->left - 1), j = 0; n <= b->right + 1; ++n, ++j) *b->exin->score_p(n) = P.sg6[j];
	    else for (int n = b->left; n <= b->right; ++n) *b->exin->score_n(n) = P.sg2[n - b->left];
	    if (b->jxt) vcopy(b->jxt, P.jx0.data(), P.jx0.size());
	}

	// ---- the library: marshalling, then one call -----------------------------------------------------------------------
const	double	t_fill0 = now_s();
	SpdpScoring	sc;
	SpdpScoringH	sch;
	std::vector<SpdpProblem> ps(protein? 0: M);
	std::vector<SpdpProblemH> ph(protein? M: 0);
	std::vector<int16_t>	ip;
	int	maxb = 0;
	for (int i = 0; i < M; ++i) maxb = std::max(maxb, pairs[live[i]].seqs[1]->len);
	ip.resize(maxb + 2);
	for (int l = 0; l < (int) ip.size(); ++l) ip[l] = pwd->IntPen->Penalty(l);
	std::vector<uint8_t>	t53_set(256, 0);
	std::vector<int16_t>	t53(256, 0);
	std::mutex	mu;
	on_threads(nthr, M, [&](int i) {
	    Pair&	P = pairs[live[i]];
	    Seq*	a = P.seqs[0];
	    Seq*	b = P.seqs[1];
	    int16_t	loc[256];
	    if (protein) {
		SpdpScoringH	tsc;
		fill_scoring_h(tsc, pwd, b);
		fill_problem_h(ph[i], a, b, P.hc, P.exin_left, P.exin_right);
		ph[i].a_pad = *a->at(a->len);
		fill_exact_h(tsc, ph[i], b, pwd, P.c, false);
		memcpy(loc, tsc.t53, sizeof loc);
		if (i == 0) sch = tsc;
	    } else {
		SpdpScoring	tsc;
		fill_scoring(tsc, pwd, b);
		fill_problem(ps[i], a, b, P.s5, P.s3);
		fill_exact_s(tsc, ps[i], b, pwd, P.c, false);
		memcpy(loc, tsc.t53, sizeof loc);
		if (i == 0) sc = tsc;
	    }
	    {
		std::lock_guard<std::mutex> g(mu);	// the junction table is one per species: a window fills the classes it holds
		for (int u = 0; u < 256; ++u) if (loc[u] && !t53_set[u]) { t53[u] = loc[u]; t53_set[u] = 1; }
	    }
	    for (int j = 0; b->jxt && j <= b->CdsNo; ++j) {
const		JUXT& t = b->jxt[j];
		SpdpJuxt q = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		P.jx.push_back(q);
	    }
	});
	SpdpSeedParams sp;
	fill_seed_params(sp, pwd, pairs[live[0]].seqs[1]);
	if (protein) { memcpy(sch.t53, t53.data(), sizeof sch.t53); sch.intpen = ip.data(); sch.intpen_len = (int) ip.size(); }
	else { memcpy(sc.t53, t53.data(), sizeof sc.t53); sc.intpen = ip.data(); sc.intpen_len = (int) ip.size(); }
	std::vector<const SpdpJuxt*> lists(M);
	std::vector<int32_t> counts(M), lowest(M);
	BatchSrc	bs;
	bs.pwd = pwd;
	for (int i = 0; i < M; ++i) {
	    Pair&	P = pairs[live[i]];
	    lists[i] = P.jx.empty()? 0: P.jx.data();
	    counts[i] = P.seqs[1]->jxt? P.seqs[1]->CdsNo: 0;
	    lowest[i] = P.seqs[1]->wllvl;
	    bs.of_query.push_back(&P);
	}
	SpdpHspSource src = {&bs, batch_units, 0};
	std::vector<SpdpAlignment> al(M);
const	double	t_fill = now_s() - t_fill0;
	// first call: cold (pinned staging, lane contexts, device pools are created in it); second call: what a running service sees
	double	t_cold = 0, t_gpu = 0;
	int	rc = 0;
	for (int pass = 0; pass < 2; ++pass) {
	    if (pass) spdp_free_alignments(al.data(), M);
const	    double	t0 = now_s();
	    rc = protein? spdp_align_h_seeded(ctx, &sch, &sp, ph.data(), M, lists.data(), counts.data(), lowest.data(), &src, al.data())
		: spdp_align_s_seeded(ctx, &sc, &sp, ps.data(), M, lists.data(), counts.data(), lowest.data(), &src, al.data());
	    (pass? t_gpu: t_cold) = now_s() - t0;
	    if (rc < 0) break;
	}
	if (rc < 0) { fprintf(stderr, "seed_bench: %s\n", spdp_last_error(ctx)); return 1; }

	int	compared = 0, same = 0, first_bad = -1;
	for (int i = 0; i < M; ++i) {
	    Pair&	P = pairs[live[i]];
	    if (al[i].n_skl < 0 || (rc == 1 && al[i].score == SPDP_NEVSEL && !al[i].n_skl && !P.ref_skl.empty())) continue;	// not served
	    ++compared;
	    bool	ok = P.ref_scr == al[i].score;
	    if (P.ref_skl.empty()) ok = ok && al[i].n_skl == 0;
	    else {
		ok = ok && al[i].n_skl == (int) P.ref_skl.size() && P.ref_skl[0].m == al[i].skl[0].m && P.ref_skl[0].n == al[i].skl[0].n;
		for (int j = 1; ok && j < al[i].n_skl; ++j)
		    ok = P.ref_skl[j].m == al[i].skl[j].m && P.ref_skl[j].n == al[i].skl[j].n;
	    }
	    if (ok) ++same;
	    else if (first_bad < 0) first_bad = live[i];
	}
	int64_t st[12] = {0};
	spdp_seeded_stats(ctx, st, 12);
	printf("{\"pairs\": %d, \"walked\": %d, \"compared\": %d, \"identical\": %d, \"first_different\": %d, \"threads\": %d, "
	       "\"prep_s\": %.4f, \"reference_s\": %.4f, \"library_s\": %.4f, \"library_cold_s\": %.4f, \"marshal_s\": %.4f, "
	       "\"batches\": %lld, \"lsp\": %lld, \"trcbk\": %lld, \"cut\": %lld, \"wilip\": %lld, "
	       "\"upload_ms\": %.1f, \"walks_ms\": %.1f, \"device_ms\": %.1f, \"hand_ms\": %.1f}\n",
	       N, M, compared, same, first_bad, nthr, t_prep, t_ref, t_gpu, t_cold, t_fill,
	       (long long) st[0], (long long) st[1], (long long) st[2], (long long) st[3], (long long) st[4], st[6] / 1e3, st[7] / 1e3, st[8] / 1e3, st[9] / 1e3);
	spdp_free_alignments(al.data(), M);
	spdp_destroy(ctx);
	return same == compared? 0: 1;
}
