// spaln_gpu_shim -- the reference's own command line program with its aligner calls switched to libspdp_hip.so: the
// reference-side binding of INTEGRATION.md ("the whole program"), i.e. what a maintainer of ogotoh/spaln links instead
// of the bodies of alignS_ng / alignH_ng.  Built by oracle/ref_build/Makefile into oracle/_ref/spaln_gpu where the
// reference's sources are (/root/reference); runs wherever a GPU is.
//
// What is linked: src/spaln.cc and every library object of the reference exactly as in oracle/_ref/spaln, except that
// src/fwd2s1.cc and src/fwd2h1.cc are compiled once more with -DalignS_ng=alignS_ng_ref / -DalignH_ng=alignH_ng_ref (the
// reference's functions under other names, nothing copied); this file supplies alignS_ng and alignH_ng (src/aln.h:351-356,
// src/fwd2s1.cc:2746, src/fwd2h1.cc:3310).  So spalign2 (src/spaln.cc:666-697),
// called by the `-t N` worker threads of match_2 / blkaln, lands here; everything around it -- block search, Exinon,
// skl_rngS_ng, the output writers -- is the reference's.
//
// The worker threads are the producers of a batch: a call parks its pair, and the first thread to find no batch running
// becomes its leader -- it waits a moment for company, runs ONE library call over everything parked (spdp_align_s, or
// spdp_align_s_seeded with the reference's own Wilip behind the HSP callback when algmode.qck != 0), hands the results
// back and wakes the others.  What the library does not take (ori = 2; in this parking form also ori = 3 with seeding: the
// batching boundary below records both orientations) goes to alignS_ng_ref and is counted.  SPALN_GPU_BATCH / SPALN_GPU_WAIT_US size a batch;
// the counts are printed to stderr at exit.
#include "shim_fill.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

SKL* alignS_ng_ref(Seq* seqs[], const PwdB* pwd, Gsinfo* gsi, int ori);	// src/fwd2s1.cc, compiled under this name
SKL* alignH_ng_ref(const Seq* seqs[], const PwdB* pwd, Gsinfo* gsi);		// src/fwd2h1.cc, likewise

bool batch_mode();

namespace {

struct Req {
	Seq**	seqs;
const	PwdB*	pwd;
	int	kind;			// 0 alignS_ng(.., 1), 1 the same with seeding on, 2 HomScoreS_ng, 3 alignH_ng, 4 alignH_ng with seeding on
	SpdpProblem p;
	SpdpProblemH ph;
	HCols hc;
	bool	undefined = false;	// protein: a DP call the reference itself leaves undefined (the caller falls back to it)
	std::vector<int16_t> s5, s3;
	SeedCols c;
	std::vector<SpdpJuxt> jx;
	SKL*	skl = 0;
	VTYPE	scr = 0;
	bool	done = false;
};

std::mutex		g_m;
std::condition_variable	g_cv;
std::vector<Req*>	g_parked;
int			g_leaders = 0;			// batches being gathered / run right now (at most g_nctx)
std::vector<SpdpContext*> g_ctxs;			// one library context per concurrent batch (SPALN_GPU_CONTEXTS, default 4): a seeded
std::vector<int>	g_ctx_free;			// call is a chain of short dependent launches that leaves most of the device idle,
int			g_nctx = 4;			// so several calls side by side multiply the throughput
thread_local SpdpContext* g_ctx = 0;			// the context of the batch this thread leads
thread_local std::vector<int16_t> g_ipen;			// IntronPenalty::Penalty(len), as long as the longest window so far
std::atomic<long>	g_calls[6], g_batches, g_largest, g_us[2], g_seed[11];
int			g_max_batch = 256, g_wait_us = 300;

// SPALN_GPU_DEBUG=1: a backtrace on SIGSEGV (the box has no debugger) and a line per stage
bool g_dbg = false;
void on_segv(int)
{
	void* bt[48];
const	int n = backtrace(bt, 48);
	backtrace_symbols_fd(bt, n, 2);
	_exit(139);
}
struct DbgInit { DbgInit() { if (getenv("SPALN_GPU_DEBUG")) { g_dbg = true; signal(SIGSEGV, on_segv); fprintf(stderr, "[spaln_gpu] loaded\n"); } } } g_dbg_init;

bool own_wilip();
void report()
{
	fprintf(stderr, "[spaln_gpu] alignS_ng on the device: %ld plain, %ld seeded, %ld score-only; alignH_ng: %ld plain, %ld seeded; "
		"left to the reference: %ld; %ld library calls, largest batch %ld\n", g_calls[0].load(), g_calls[1].load(), g_calls[2].load(),
		g_calls[4].load(), g_calls[5].load(), g_calls[3].load(), g_batches.load(), g_largest.load());
	fprintf(stderr, "[spaln_gpu] cDNA batches: %.2f s turning Seq / Exinon into the arrays of include/spdp.h, %.2f s inside the library\n",
		g_us[0].load() * 1e-6, g_us[1].load() * 1e-6);
	if (g_seed[5].load())
	    fprintf(stderr, "[spaln_gpu] seeded calls: upload %.2f s, walks alone %.2f s, device batches %.2f s (%ld batches, %ld lspS_ng, %ld tracebacks), "
		"hand-back %.2f s, in all %.2f s; HSP searches of the recursion levels %ld (%s)\n", g_seed[6].load() * 1e-6, g_seed[7].load() * 1e-6, g_seed[8].load() * 1e-6,
		g_seed[0].load(), g_seed[1].load(), g_seed[2].load(), g_seed[9].load() * 1e-6, g_seed[10].load() * 1e-6, g_seed[4].load(),
		own_wilip()? "the library's own": "the reference's Wilip through the callback");
}

// the HSP searches of the recursion levels: the library's own (spdp_wilip.h, round 5) unless SPALN_GPU_WILIP=ref asks for the
// reference's Wilip behind the callback
bool own_wilip()
{
	static const bool on = [] { const char* e = getenv("SPALN_GPU_WILIP"); return !(e && !strcmp(e, "ref")); }();
	return on;
}
int units_cb(void* user, int32_t q, int32_t level, const int32_t span[8], const int32_t** flat, int32_t* n_flat)
{
	Req* r = ((Req**) user)[q];
	wilip_flat(r->seqs, r->pwd, level, span, r->c.flat);
	*flat = r->c.flat.data(); *n_flat = (int32_t) r->c.flat.size();
	return 0;
}

// the reference's walk writes the phase of every junction it chooses itself into the Exinon (src/fwd2h1.cc:2508-2517) and
// skl_rngH_ng reads it there (:824-825): the library hands those marks out instead of writing into the caller's objects
void apply_phase_marks(Seq* b, int q)
{
const	SpdpPhaseMark* mk = 0;
const	int	n = spdp_seeded_phase_marks(g_ctx, q, &mk);
	for (int k = 0; k < n; ++k) {
	    SGPT6* g = b->exin->score_p(mk[k].n);
	    if (mk[k].side == 5) g->phs5 = mk[k].value; else g->phs3 = mk[k].value;
	}
}

SKL* to_skl(const SpdpAlignment& al, const Seq* a)
{
	if (!al.n_skl) return 0;
	SKL* skl = new SKL[al.n_skl + 1];		// the caller (~Gsinfo) delete[]s it
	memcpy(skl, al.skl, sizeof(SKL) * al.n_skl);
	skl[al.n_skl].m = skl[al.n_skl].n = EOS;
	if (a->inex.sens) skl->m |= A_RevCom;
	return skl;
}

// one library call over the protein requests of one kind (3 plain, 4 seeded)
void run_kind_h(std::vector<Req*>& rq, int kind)
{
const	int n = (int) rq.size();
	SpdpScoringH sc;
	fill_scoring_h(sc, rq[0]->pwd, rq[0]->seqs[1]);
	int	longest = 0;
	for (Req* r : rq) longest = std::max(longest, r->seqs[1]->len);
	if ((int) g_ipen.size() < longest + 2) {
const	    int from = (int) g_ipen.size();
	    g_ipen.resize(longest + 2);
	    for (int l = from; l < longest + 2; ++l) g_ipen[l] = rq[0]->pwd->IntPen->Penalty(l);
	}
	std::vector<SpdpProblemH> probs(n);
	for (int i = 0; i < n; ++i) {
	    Req* r = rq[i];
	    Seq* a = r->seqs[0]; Seq* b = r->seqs[1];
	    fill_problem_h(r->ph, a, b, r->hc, b->left, b->right);	// the Exinon was built for the range b holds at the call (src/spaln.cc:745-752)
	    r->ph.a_pad = *a->at(a->len);			// what exg_seq left behind the query
	    fill_exact_h(sc, r->ph, b, r->pwd, r->c, false);
	    probs[i] = r->ph;
	}
	sc.intpen = g_ipen.data(); sc.intpen_len = (int) g_ipen.size();
	std::vector<SpdpAlignment> al(n);
	int rc;
	if (kind == 3) rc = spdp_align_h(g_ctx, &sc, probs.data(), n, al.data());
	else {
	    SpdpSeedParams sp;
	    fill_seed_params(sp, rq[0]->pwd, rq[0]->seqs[1]);
	    std::vector<const SpdpJuxt*> lists(n);
	    std::vector<int32_t> counts(n), lowest(n);
	    for (int i = 0; i < n; ++i) {
		Seq* b = rq[i]->seqs[1];
		for (int j = 0; b->jxt && j <= b->CdsNo; ++j) {
const		    JUXT& t = b->jxt[j];
		    SpdpJuxt q = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		    rq[i]->jx.push_back(q);
		}
		lists[i] = rq[i]->jx.empty()? 0: rq[i]->jx.data();
		counts[i] = b->jxt? b->CdsNo: 0;
		lowest[i] = b->wllvl;
	    }
	    SpdpHspSource src = {rq.data(), units_cb, 0};
	    static SpdpWilipModel wmodel; static std::once_flag wonce;
	    if (own_wilip()) { std::call_once(wonce, [&] { fill_wilip_model(wmodel, rq[0]->pwd); }); sp.wilip = &wmodel; }
	    rc = spdp_align_h_seeded(g_ctx, &sc, &sp, probs.data(), n, lists.data(), counts.data(), lowest.data(), own_wilip()? 0: &src, al.data());
	}
	if (rc < 0) fatal("spaln_gpu: %s\n", spdp_last_error(g_ctx));
	for (int i = 0; i < n; ++i) {
	    // undefined in the reference itself (spdp_align_h's return value 1: such a query comes back without records and with
	    // NEVSEL; n_skl < 0: its traceback would start outside its bitmap / "Unexpected dir")
	    rq[i]->undefined = al[i].n_skl < 0 || (rc == 1 && al[i].n_skl == 0 && al[i].score == SPDP_NEVSEL);
	    rq[i]->scr = al[i].score;
	    rq[i]->skl = al[i].n_skl > 0? to_skl(al[i], rq[i]->seqs[0]): 0;
	    if (rq[i]->skl) rq[i]->skl->m = 1;			// globalH_ng: skl->m = 1 (no A_RevCom on this path)
	    if (kind == 4 && !rq[i]->undefined) apply_phase_marks(rq[i]->seqs[1], i);
	}
	spdp_free_alignments(al.data(), n);
	++g_batches;
	if (n > g_largest) g_largest = n;
}

// one library call over the requests of one kind
void run_kind(std::vector<Req*>& rq, int kind)
{
	if (rq.empty()) return;
	if (kind >= 3) { run_kind_h(rq, kind); return; }
const	int n = (int) rq.size();
const	auto t0 = std::chrono::steady_clock::now();
	SpdpScoring sc;
	fill_scoring(sc, rq[0]->pwd, rq[0]->seqs[1]);
	int	longest = 0;
	for (Req* r : rq) longest = std::max(longest, r->seqs[1]->len);
	if ((int) g_ipen.size() < longest + 2) {
const	    int from = (int) g_ipen.size();
	    g_ipen.resize(longest + 2);
	    for (int l = from; l < longest + 2; ++l) g_ipen[l] = rq[0]->pwd->IntPen->Penalty(l);
	}
	std::vector<SpdpProblem> probs(n);
	for (int i = 0; i < n; ++i) {
	    Req* r = rq[i];
	    fill_problem(r->p, r->seqs[0], r->seqs[1], r->s5, r->s3);
	    fill_exact_s(sc, r->p, r->seqs[1], r->pwd, r->c, false);
	    probs[i] = r->p;
	}
	sc.intpen = g_ipen.data(); sc.intpen_len = (int) g_ipen.size();
	std::vector<SpdpAlignment> al(n);
	int rc = 0;
const	auto t1 = std::chrono::steady_clock::now();
	if (kind == 2) {
	    std::vector<int32_t> scr(n);
	    rc = spdp_homscore_s(g_ctx, &sc, probs.data(), n, scr.data());
	    for (int i = 0; i < n; ++i) rq[i]->scr = scr[i];
	} else if (kind == 0) {
	    rc = spdp_align_s(g_ctx, &sc, probs.data(), n, al.data());
	    if (rc > 0) rc = 0;
	} else {
	    SpdpSeedParams sp;
	    fill_seed_params(sp, rq[0]->pwd, rq[0]->seqs[1]);
	    std::vector<const SpdpJuxt*> lists(n);
	    std::vector<int32_t> counts(n), lowest(n);
	    for (int i = 0; i < n; ++i) {
		Seq* b = rq[i]->seqs[1];
		for (int j = 0; b->jxt && j <= b->CdsNo; ++j) {		// CdsNo HSPs + the free slot behind them
const		    JUXT& t = b->jxt[j];
		    SpdpJuxt q = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		    rq[i]->jx.push_back(q);
		}
		lists[i] = rq[i]->jx.empty()? 0: rq[i]->jx.data();
		counts[i] = b->jxt? b->CdsNo: 0;
		lowest[i] = b->wllvl;
	    }
	    SpdpHspSource src = {rq.data(), units_cb, 0};
	    static SpdpWilipModel wmodel; static std::once_flag wonce;
	    if (own_wilip()) { std::call_once(wonce, [&] { fill_wilip_model(wmodel, rq[0]->pwd); }); sp.wilip = &wmodel; }
	    rc = spdp_align_s_seeded(g_ctx, &sc, &sp, probs.data(), n, lists.data(), counts.data(), lowest.data(), own_wilip()? 0: &src, al.data());
	    if (rc > 0) rc = 0;
	    int64_t st[11] = {0};
	    spdp_seeded_stats(g_ctx, st, 11);
	    for (int k = 0; k < 11; ++k) g_seed[k] += st[k];
	}
	if (rc < 0) fatal("spaln_gpu: %s\n", spdp_last_error(g_ctx));
const	auto t2 = std::chrono::steady_clock::now();
	g_us[0] += std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
	g_us[1] += std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count();
	if (kind != 2) {
	    for (int i = 0; i < n; ++i) { rq[i]->scr = al[i].score; rq[i]->skl = to_skl(al[i], rq[i]->seqs[0]); }
	    spdp_free_alignments(al.data(), n);
	}
	++g_batches;
	if (n > g_largest) g_largest = n;
}

void submit(Req& r)
{
	std::unique_lock<std::mutex> lk(g_m);
	g_parked.push_back(&r);
	for (;;) {					// wait for my result, or for a leader's seat
	    if (r.done) return;
	    if (g_leaders < g_nctx && !g_parked.empty()) break;
	    g_cv.wait(lk);
	}
	// a seat is free: this thread gathers a batch (and goes on until its own request has run)
	++g_leaders;
	const int slot = g_ctx_free.back(); g_ctx_free.pop_back();
	g_ctx = g_ctxs[slot];
	while (!r.done) {
	    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(g_wait_us);
	    while ((int) g_parked.size() < g_max_batch && std::chrono::steady_clock::now() < until) {
		lk.unlock(); std::this_thread::sleep_for(std::chrono::microseconds(20)); lk.lock();
	    }
	    std::vector<Req*> take;
	    if ((int) g_parked.size() <= g_max_batch) take.swap(g_parked);
	    else { take.assign(g_parked.begin(), g_parked.begin() + g_max_batch); g_parked.erase(g_parked.begin(), g_parked.begin() + g_max_batch); }
	    if (take.empty()) {				// another leader took everything, mine included: wait for it
		g_cv.wait(lk);
		continue;
	    }
	    lk.unlock();
	    for (int kind = 0; kind < 5; ++kind) {
		std::vector<Req*> part;
		for (Req* q : take) if (q->kind == kind) part.push_back(q);
		run_kind(part, kind);
	    }
	    lk.lock();
	    for (Req* q : take) q->done = true;
	    g_cv.notify_all();
	}
	g_ctx_free.push_back(slot);
	--g_leaders;
	g_cv.notify_all();				// a parked thread takes over
}

bool device_up()
{
	static std::once_flag once;
	std::call_once(once, [] {
	    if (batch_mode()) g_nctx = 1;			// (one call over everything: one context)
	    if (const char* e = getenv("SPALN_GPU_CONTEXTS")) g_nctx = std::max(1, std::min(16, atoi(e)));
	    for (int i = 0; i < g_nctx; ++i) {
		SpdpContext* c = spdp_create(0);
		if (!c) fatal("spaln_gpu: no HIP device (there is no CPU path in the library)\n");
		g_ctxs.push_back(c); g_ctx_free.push_back(i);
	    }
	    if (const char* e = getenv("SPALN_GPU_BATCH")) g_max_batch = std::max(1, atoi(e));
	    if (const char* e = getenv("SPALN_GPU_WAIT_US")) g_wait_us = std::max(0, atoi(e));
	    atexit(report);
	});
	return !g_ctxs.empty();
}

VTYPE homscore(Seq* seqs[], const PwdB* pwd)
{
	Req r; r.seqs = seqs; r.pwd = pwd; r.kind = 2;
	submit(r);
	++g_calls[2];
	return r.scr;
}

}	// namespace

// SPALN_GPU_TIME_REF=1: no device at all -- every call goes to the reference's own aligner and the time inside it is added up
// over the worker threads: what share of the program's worker time the call this library replaces is (Amdahl's bound on
// what ANY faster aligner can do for the program; INTEGRATION.md, "where the time of -Q7 goes")
std::atomic<long>	g_ref_us, g_ref_calls;
std::chrono::steady_clock::time_point g_t_start = std::chrono::steady_clock::now();
bool time_ref_mode()
{
	static const bool on = [] {
	    const bool v = getenv("SPALN_GPU_TIME_REF") != 0;
	    if (v) atexit([] {
		const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t_start).count();
		fprintf(stderr, "[spaln_gpu] time-ref mode: %ld aligner calls, %.3f s inside the reference's aligner summed over the worker threads, "
			"program wall %.3f s\n", g_ref_calls.load(), g_ref_us.load() * 1e-6, wall);
	    });
	    return v;
	}();
	return on;
}


// ---- the batching boundary (round 5): map first, align afterwards -----------------------------------------------------
// The reference's worker (src/spaln.cc:1363-1384) takes one query through block search, alignment, rescoring and output before it
// looks at the next: its aligner call sees one pair at a time, and a device wants thousands.  In batch mode (the default;
// SPALN_GPU_MODE=park restores the thread-parking form above) a worker's aligner call only RECORDS its pair -- deep copies
// of the two Seq (the window keeps its Exinon and its HSPs), marshalled into the arrays of include/spdp.h right there, on the
// worker's own thread -- and answers "no alignment", so the reference's blkaln moves on to the next candidate locus and the
// next query at once: the `-t` workers of the reference map the whole input at host speed.  When they have joined
// (MasterWorker calls closeGeneRecord(), src/spaln.cc:1454: spaln_gpu_main.cc points that one call here) every recorded
// pair goes through ONE library call, and what blkaln does with an alignment -- spalign2's rescoring (src/spaln.cc:680-690),
// the Vthr filter, the order by fstat.val, Gsinfo / alnoutput (src/spaln.cc:938-975) -- is done here, per query, on the
// same number of threads.  Needs the statics of src/spaln.cc (outputs, skl2exrng): this file is compiled as part of
// spaln_gpu_main.cc, one translation unit with it.
struct Job {
	Seq*	a = 0;			// the query as the call saw it (its own copy: the HSP callback moves its range about)
	Seq*	b = 0;			// the window, with the Exinon the reference built for it and its HSPs
const	PwdB*	pwd = 0;
	int	kind = 0;		// 0 alignS_ng(.., 1), 1 the same with seeding on, 3 alignH_ng, 4 alignH_ng with seeding on
	long	group = 0;		// calls of one blkaln (one query): consecutive on their worker
	int	order = 0;		// ... in the order blkaln made them
	SpdpProblem p;
	SpdpProblemH ph;		// (kinds 3 / 4: alignH_ng)
	HCols	hc;
	std::vector<int16_t> s5, s3;
	SeedCols c;
	std::vector<SpdpJuxt> jx;
	int16_t	t53[256];		// the junction-pair table as this window shows it (entries of classes it lacks are 0)
	Gsinfo	gsi;
	bool	keep = false;
	// ori = 3 (alignS_ng tries both orientations: src/fwd2s1.cc:2742-2777): the reverse-complemented query and the other strand of
	// the window (genomicseq made it, with an Exinon of its own: src/spaln.cc:1146-1151), marshalled like the pair as given
	int	ori = 1;
	Seq*	a2 = 0;
	Seq*	b2 = 0;
	SpdpProblem p2;
	std::vector<int16_t> s5r, s3r;
	SeedCols cr;
	~Job() { delete a; delete a2; if (b) { delete b->exin; b->exin = 0; delete b; } if (b2) { delete b2->exin; b2->exin = 0; delete b2; } }
};
std::mutex		g_jm;
std::condition_variable	g_jcv;
std::thread		g_warm, g_aligner;
std::vector<Job*>	g_jobs;			// complete groups, ready for the device
struct Pending { std::vector<Job*> jobs; };	// the group a worker is still adding to (one per worker thread, kept past its exit)
std::vector<Pending*>	g_pending;
bool			g_input_done = false;
int			g_chunk = 4096;		// SPALN_GPU_CHUNK: jobs that make a library call worth starting while the workers still map
std::atomic<long>	g_groups;
void aligner_loop();
bool batch_mode()
{
	static const bool on = [] { const char* e = getenv("SPALN_GPU_MODE"); return !(e && !strcmp(e, "park")); }();
	return on;
}
// the worker's side: one aligner call = one job
void record_job(Seq* seqs[], const PwdB* pwd, int kind, int ori = 1)
{
	static thread_local long	t_group = -1;
	static thread_local int	t_order = 0;
	static thread_local const Seq* t_a = 0;
	static thread_local int	t_sid = -1, t_left = -1, t_right = -1;
	static thread_local Pending* t_pend = 0;
	static std::once_flag warm;			// the HIP context comes up (about a second) while the workers map, and a thread
	std::call_once(warm, [] {			// of its own aligns what they have recorded, chunk by chunk, beside them
	    if (const char* e = getenv("SPALN_GPU_CHUNK")) g_chunk = std::max(1, atoi(e));
	    g_warm = std::thread([] { device_up(); });
	    g_aligner = std::thread(aligner_loop);
	});
	Seq*	a = seqs[0];
	Seq*	b = seqs[1];
	if (!t_pend) { t_pend = new Pending; std::lock_guard<std::mutex> lk(g_jm); g_pending.push_back(t_pend); }
	if (t_group < 0 || t_a != a || t_sid != a->sid || t_left != a->left || t_right != a->right) {	// another blkaln: the last one's group is complete
	    if (!t_pend->jobs.empty()) {
		std::lock_guard<std::mutex> lk(g_jm);
		g_jobs.insert(g_jobs.end(), t_pend->jobs.begin(), t_pend->jobs.end());
		t_pend->jobs.clear();
		if ((int) g_jobs.size() >= g_chunk) g_jcv.notify_one();
	    }
	    t_group = g_groups++; t_order = 0; t_a = a; t_sid = a->sid; t_left = a->left; t_right = a->right;
	}
	Job*	j = new Job;
	j->pwd = pwd; j->kind = kind; j->group = t_group; j->order = t_order++;
	j->a = a->copyseq(0, CPY_ALL);
	j->b = b->copyseq(0, CPY_ALL);
	j->b->exin = b->exin; b->exin = 0;		// (Exinon reads its Seq in its constructor only; blkaln deletes it right after the call)
	j->b->CdsNo = b->CdsNo; j->b->wllvl = b->wllvl;
	if (b->jxt) { j->b->jxt = new JUXT[b->CdsNo + 1]; vcopy(j->b->jxt, b->jxt, b->CdsNo + 1); }
	j->a->CdsNo = a->CdsNo;
	if (kind >= 3) {
	    SpdpScoringH sc;
	    fill_scoring_h(sc, pwd, j->b);
	    fill_problem_h(j->ph, j->a, j->b, j->hc, j->b->left, j->b->right);	// the Exinon was built for the range b holds at the call
	    j->ph.a_pad = *j->a->at(j->a->len);			// what exg_seq left behind the query
	    fill_exact_h(sc, j->ph, j->b, pwd, j->c, false);
	    memcpy(j->t53, sc.t53, sizeof j->t53);
	} else {
	    fill_problem(j->p, j->a, j->b, j->s5, j->s3);
	    SpdpScoring sc;				// (per job: only its junction table differs, and that is per window)
	    fill_scoring(sc, pwd, j->b);
	    fill_exact_s(sc, j->p, j->b, pwd, j->c, false);
	    memcpy(j->t53, sc.t53, sizeof j->t53);
	    if (ori == 3) {				// the other orientation: comrev(a) against the anti-strand Seq genomicseq left beside b
		Seq*	anti = *b->getanti();
		j->ori = 3;
		j->a2 = a->copyseq(0, CPY_ALL);
		j->a2->comrev();
		j->b2 = anti->copyseq(0, CPY_ALL);
		j->b2->exin = anti->exin; anti->exin = 0;
		fill_problem(j->p2, j->a2, j->b2, j->s5r, j->s3r);
		SpdpScoring sc2;
		fill_scoring(sc2, pwd, j->b2);
		fill_exact_s(sc2, j->p2, j->b2, pwd, j->cr, false);
		for (int i = 0; i < 256; ++i) if (!j->t53[i]) j->t53[i] = sc2.t53[i];
	    }
	}
	if (kind == 1 || kind == 4)
	    for (int k = 0; j->b->jxt && k <= j->b->CdsNo; ++k) {		// CdsNo HSPs + the free slot behind them
const		JUXT& t = j->b->jxt[k];
		SpdpJuxt q = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		j->jx.push_back(q);
	    }
	t_pend->jobs.push_back(j);			// (my own list: no lock)
}

SKL* alignS_ng(Seq* seqs[], const PwdB* pwd, Gsinfo* gsi, int ori)
{
	if (time_ref_mode()) {
const	    auto t0 = std::chrono::steady_clock::now();
	    SKL* skl = alignS_ng_ref(seqs, pwd, gsi, ori);
	    g_ref_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
	    ++g_ref_calls;
	    return skl;
	}
	if (g_dbg) fprintf(stderr, "[spaln_gpu] alignS_ng ori %d qck %d\n", ori, (int) algmode.qck);
	// recorded now, aligned with everybody else's once the workers have joined: the pair as given, or (ori = 3, the default for a
	// cDNA without a poly-A tail) both orientations -- with seeding on only when the HSP searches are the library's own
	if (batch_mode() && ori == 2) {
	    // -S2: the other orientation alone.  What the reference does to its operands before it aligns (src/fwd2s1.cc:2750-2753: the HSP
	    // list turned around and handed to the anti-strand Seq, the query reverse-complemented, the strands swapped) is done here, at
	    // the call; the pair is then recorded as given -- the library's ori = 1 on the reverse problem
	    Seq*&	b = seqs[1];
	    if (b->jxt) {
		Seq*	c = b;
		if (b->getanti()) {
		    c = seqs[2];
		    if (c->jxt) delete[] c->jxt;
		    c->CdsNo = b->CdsNo;
		    c->jxt = new JUXT[b->CdsNo + 1];
		    vcopy(c->jxt, b->jxt, b->CdsNo + 1);
		}
		c->revjxt();
	    }
	    seqs[0]->comrev();
	    antiseq(seqs + 1);
	    ori = 1;
	}
	if (batch_mode() && algmode.mlt != 1 && (ori == 1 || (ori == 3 && seqs[1]->getanti() && (!algmode.qck || own_wilip())))) {
	    record_job(seqs, pwd, algmode.qck? 1: 0, ori);
	    ++g_calls[algmode.qck? 1: 0];
	    return 0;					// "no alignment": blkaln goes on to the next locus / query (src/spaln.cc:907-912)
	}
	device_up();
	if (g_dbg) fprintf(stderr, "[spaln_gpu] device up\n");
	if (ori == 2 || (ori == 3 && algmode.qck)) { ++g_calls[3]; return alignS_ng_ref(seqs, pwd, gsi, ori); }
	if (ori == 3) {				// infer_orientation (src/fwd2s1.cc:2716-2728), the two scores from the device
	    Seq*& a = seqs[0];
const	    VTYPE scr1 = homscore(seqs, pwd);
	    a->comrev();
	    antiseq(seqs + 1);
const	    VTYPE scr2 = homscore(seqs, pwd);
	    if (!(scr2 > scr1)) { a->comrev(); antiseq(seqs + 1); }
	}
	Req r; r.seqs = seqs; r.pwd = pwd; r.kind = algmode.qck? 1: 0;
	submit(r);
	++g_calls[r.kind];
	gsi->scr = r.scr;
	return r.skl;
}

SKL* alignH_ng(const Seq* seqs[], const PwdB* pwd, Gsinfo* gsi)
{
	if (time_ref_mode()) {
const	    auto t0 = std::chrono::steady_clock::now();
	    SKL* skl = alignH_ng_ref(seqs, pwd, gsi);
	    g_ref_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
	    ++g_ref_calls;
	    return skl;
	}
	if (batch_mode() && algmode.mlt != 1) {
	    record_job((Seq**) seqs, pwd, algmode.qck? 4: 3);
	    ++g_calls[algmode.qck? 5: 4];
	    return 0;
	}
	device_up();
	Req r; r.seqs = (Seq**) seqs; r.pwd = pwd; r.kind = algmode.qck? 4: 3;
	submit(r);
	if (r.undefined) {			// the reference's own result is undefined there (it reads outside its traceback bitmap / the sequences):
	    ++g_calls[3];			// a production shim leaves the case to the host, as INTEGRATION.md says
	    return alignH_ng_ref(seqs, pwd, gsi);
	}
	++g_calls[r.kind + 1];
	gsi->scr = r.scr;
	return r.skl;
}

// ---- after the workers have joined: one library call, then what blkaln does with an alignment ------------------------------
static RANGE* exrng_of(SKL* skl)			// skl2exrng, src/spaln.cc:646-663 (static there; same TU: call it)
{
	return skl2exrng(skl);
}
// one library call over `jobs` (complete groups), then spalign2's second half, blkaln's filter, order and output for them
static std::mutex g_out_m;
void align_jobs(std::vector<Job*>& jobs, int nthr)
{
	if (jobs.empty()) return;
	{
const	    auto t0 = std::chrono::steady_clock::now();
	    if (g_warm.joinable()) g_warm.join();
	    device_up();
	    g_ctx = g_ctxs[0];
	    std::stable_sort(jobs.begin(), jobs.end(), [](const Job* x, const Job* y) { return x->group != y->group? x->group < y->group: x->order < y->order; });
const	    int	n = (int) jobs.size();
	    // ---- one call per kind over everything recorded
	    for (int ko = 0; ko < 4; ++ko) {
const		int kind = ko & 1, ori = ko & 2? 3: 1;
		std::vector<Job*> part;
		for (Job* j : jobs) if (j->kind == kind && j->ori == ori) part.push_back(j);
		if (part.empty()) continue;
const		int m = (int) part.size();
		SpdpScoring sc;
		fill_scoring(sc, part[0]->pwd, part[0]->b);
		{ SeedCols tmp; SpdpProblem pp = part[0]->p; fill_exact_s(sc, pp, part[0]->b, part[0]->pwd, tmp, false); }	// minl, scalar_engines
		memset(sc.t53, 0, sizeof sc.t53);
		int	longest = 0;
		for (Job* j : part) {
		    longest = std::max(longest, j->b->len);
		    for (int i = 0; i < 256; ++i) if (!sc.t53[i]) sc.t53[i] = j->t53[i];
		}
		g_ipen.resize(longest + 2);
		for (int l = 0; l < longest + 2; ++l) g_ipen[l] = part[0]->pwd->IntPen->Penalty(l);
		sc.intpen = g_ipen.data(); sc.intpen_len = (int) g_ipen.size();
		std::vector<SpdpProblem> probs(m), rev(ori == 3? m: 0);
		for (int i = 0; i < m; ++i) { probs[i] = part[i]->p; if (ori == 3) rev[i] = part[i]->p2; }
		std::vector<SpdpAlignment> al(m);
		std::vector<int32_t> orient(m, 0);
		int	rc;
const		auto t1 = std::chrono::steady_clock::now();
		if (kind == 0) rc = ori == 3? spdp_align_s_ori3(g_ctx, &sc, probs.data(), rev.data(), m, al.data(), orient.data())
					    : spdp_align_s(g_ctx, &sc, probs.data(), m, al.data());
		else {
		    SpdpSeedParams sp;
		    fill_seed_params(sp, part[0]->pwd, part[0]->b);
		    std::vector<const SpdpJuxt*> lists(m);
		    std::vector<int32_t> counts(m), lowest(m);
		    std::vector<Req> rq(m);			// the HSP callback's view of a job
		    std::vector<Req*> rqp(m);
		    for (int i = 0; i < m; ++i) {
			Job* j = part[i];
			lists[i] = j->jx.empty()? 0: j->jx.data();
			counts[i] = j->b->jxt? j->b->CdsNo: 0;
			lowest[i] = j->b->wllvl;
			rq[i].seqs = &j->a;			// (a and b sit side by side in the job: seqs[0], seqs[1])
			rq[i].pwd = j->pwd;
			rqp[i] = &rq[i];
		    }
		    SpdpHspSource src = {rqp.data(), units_cb, 0};
		    static SpdpWilipModel wmodel; static std::once_flag wonce;
		    if (own_wilip()) { std::call_once(wonce, [&] { fill_wilip_model(wmodel, part[0]->pwd); }); sp.wilip = &wmodel; }
		    if (ori == 3) { sp.both_ori = 1;		// (Exinon::both_ori of these windows; recorded only when the HSP searches are the library's)
			rc = spdp_align_s_seeded_ori3(g_ctx, &sc, &sp, probs.data(), rev.data(), m, lists.data(), counts.data(), lowest.data(), 0, al.data(), orient.data()); }
		    else rc = spdp_align_s_seeded(g_ctx, &sc, &sp, probs.data(), m, lists.data(), counts.data(), lowest.data(), own_wilip()? 0: &src, al.data());
		    int64_t st[11] = {0};
		    spdp_seeded_stats(g_ctx, st, 11);
		    for (int k = 0; k < 11; ++k) g_seed[k] += st[k];
		}
		if (rc < 0) fatal("spaln_gpu: %s\n", spdp_last_error(g_ctx));
		g_us[1] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count();
		for (int i = 0; i < m; ++i) {
		    Job* j = part[i];
		    if (orient[i]) {			// the reverse orientation stays: what alignS_ng leaves in seqs[0], seqs[1] then (src/fwd2s1.cc:2771-2775)
			if (kind == 1 && j->b->jxt) {	// (seeded path only) reverse_copy_jxt (:2730-2743): the anti-strand Seq carries the HSP list turned around, the
			    Seq* c = j->b2;		// slot behind it as the forward walk left it ({a->len, b->len}, :2693)
			    delete[] c->jxt;
			    c->CdsNo = j->b->CdsNo;
			    c->jxt = new JUXT[c->CdsNo + 1];
			    vcopy(c->jxt, j->b->jxt, c->CdsNo + 1);
			    c->jxt[c->CdsNo].jx = j->a->len; c->jxt[c->CdsNo].jy = j->b->len;
			    c->revjxt();
			}
			std::swap(j->a, j->a2); std::swap(j->b, j->b2);
		    }
		    j->gsi.scr = al[i].score; j->gsi.skl = to_skl(al[i], j->a);
		}
		spdp_free_alignments(al.data(), m);
		++g_batches;
		if (m > g_largest) g_largest = m;
	    }
	    for (int kind = 3; kind < 5; ++kind) {		// alignH_ng, plain and seeded
		std::vector<Job*> part;
		for (Job* j : jobs) if (j->kind == kind) part.push_back(j);
		if (part.empty()) continue;
const		int m = (int) part.size();
		SpdpScoringH sc;
		fill_scoring_h(sc, part[0]->pwd, part[0]->b);
		{ SeedCols tmp; SpdpProblemH pp = part[0]->ph; fill_exact_h(sc, pp, part[0]->b, part[0]->pwd, tmp, false); }	// the scalars of the exact model
		memset(sc.t53, 0, sizeof sc.t53);
		int	longest = 0;
		for (Job* j : part) {
		    longest = std::max(longest, j->b->len);
		    for (int i = 0; i < 256; ++i) if (!sc.t53[i]) sc.t53[i] = j->t53[i];
		}
		g_ipen.resize(longest + 2);
		for (int l = 0; l < longest + 2; ++l) g_ipen[l] = part[0]->pwd->IntPen->Penalty(l);
		sc.intpen = g_ipen.data(); sc.intpen_len = (int) g_ipen.size();
		std::vector<SpdpProblemH> probs(m);
		for (int i = 0; i < m; ++i) probs[i] = part[i]->ph;
		std::vector<SpdpAlignment> al(m);
		int	rc;
const		auto t1 = std::chrono::steady_clock::now();
		if (kind == 3) rc = spdp_align_h(g_ctx, &sc, probs.data(), m, al.data());
		else {
		    SpdpSeedParams sp;
		    fill_seed_params(sp, part[0]->pwd, part[0]->b);
		    std::vector<const SpdpJuxt*> lists(m);
		    std::vector<int32_t> counts(m), lowest(m);
		    std::vector<Req> rq(m);
		    std::vector<Req*> rqp(m);
		    for (int i = 0; i < m; ++i) {
			Job* j = part[i];
			lists[i] = j->jx.empty()? 0: j->jx.data();
			counts[i] = j->b->jxt? j->b->CdsNo: 0;
			lowest[i] = j->b->wllvl;
			rq[i].seqs = &j->a; rq[i].pwd = j->pwd;
			rqp[i] = &rq[i];
		    }
		    SpdpHspSource src = {rqp.data(), units_cb, 0};
		    static SpdpWilipModel wmodel; static std::once_flag wonce;
		    if (own_wilip()) { std::call_once(wonce, [&] { fill_wilip_model(wmodel, part[0]->pwd); }); sp.wilip = &wmodel; }
		    rc = spdp_align_h_seeded(g_ctx, &sc, &sp, probs.data(), m, lists.data(), counts.data(), lowest.data(), own_wilip()? 0: &src, al.data());
		}
		if (rc < 0) fatal("spaln_gpu: %s\n", spdp_last_error(g_ctx));
		g_us[1] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count();
		for (int i = 0; i < m; ++i) {
		    Job* j = part[i];
		    // undefined in the reference itself (it reads outside its traceback bitmap / the sequences): its own aligner decides
		    const bool undefined = al[i].n_skl < 0 || (rc == 1 && al[i].n_skl == 0 && al[i].score == SPDP_NEVSEL);
		    if (undefined) {
			Seq* sqs[2] = {j->a, j->b};
			j->gsi.skl = alignH_ng_ref((const Seq**) sqs, j->pwd, &j->gsi);
			++g_calls[3];
		    } else {
			j->gsi.scr = al[i].score;
			j->gsi.skl = al[i].n_skl > 0? to_skl(al[i], j->a): 0;
			if (j->gsi.skl) j->gsi.skl->m = 1;		// globalH_ng: skl->m = 1 (no A_RevCom on this path)
			if (kind == 4) apply_phase_marks(j->b, i);
		    }
		}
		spdp_free_alignments(al.data(), m);
		++g_batches;
		if (m > g_largest) g_largest = m;
	    }
	    // ---- spalign2's second half and blkaln's filter (src/spaln.cc:680-696, 907-912), the jobs spread over the threads
	    {
		std::atomic<int> next(0);
		auto work = [&] {
		    for (int i; (i = next++) < n; ) {
			Job* j = jobs[i];
			Gsinfo* g = &j->gsi;
			Seq* sqs[2] = {j->a, j->b};
			bool ok = g->skl && g->skl->n != 0;
			if (ok) {
			    if (g->skl->m & AlgnTrb) {
				g->scr = j->kind >= 3? skl_rngH_ng((const Seq**) sqs, g, j->pwd): skl_rngS_ng((const Seq**) sqs, g, j->pwd);
				g->eiscr2rng();
			    }
			    else g->CDSrng = exrng_of(g->skl);
			}
			delete j->b->exin; j->b->exin = 0;		// suppress Boundary output
			j->keep = ok && (OutPrm.all_out || g->scr > j->pwd->Vthr);
			if (!j->keep) g->scr = NEVSEL;
		    }
		};
		std::vector<std::thread> th;
		for (int t = 1; t < nthr; ++t) th.emplace_back(work);
		work();
		for (auto& t : th) t.join();
	    }
	    // ---- per query: the order by fstat.val and the output (src/spaln.cc:938-975)
	    for (int i0 = 0; i0 < n; ) {
		int i9 = i0;
		while (i9 < n && jobs[i9]->group == jobs[i0]->group) ++i9;
const		int np = i9 - i0;
		std::vector<int> odr(np);
		for (int k = 0; k < np; ++k) odr[k] = k;
		for (int k = 1; k < np; ++k) {			// insert sort, as nearly sorted
		    int	l = odr[k];
		    VTYPE v = jobs[i0 + l]->gsi.fstat.val;
		    int	mm = k;
		    while (--mm >= 0 && v > jobs[i0 + odr[mm]]->gsi.fstat.val) odr[mm + 1] = odr[mm];
		    odr[mm + 1] = l;
		}
		INT	n_out = 0;
		for (int k = 0; k < np; ++k) n_out += jobs[i0 + k]->keep;
		if (OutPrm.MaxOut < n_out) n_out = OutPrm.MaxOut;
		for (INT k = 0; k < n_out; ++k) {
		    Job* j = jobs[i0 + odr[k]];
		    Gsinfo* g = &j->gsi;
		    if (!g->skl) continue;
		    Seq* sqs[2] = {j->a, j->b};
		    if (bool(g->skl->m & A_RevCom) ^ bool(j->a->inex.sens)) j->a->comrev();
		    if (algmode.nsa == BED_FORM) g->rscr = selfAlnScr(j->a, j->pwd->simmtx);
		    { std::lock_guard<std::mutex> lk(g_out_m); outputs.alnoutput(sqs, g); }
		}
		i0 = i9;
	    }
	    for (Job* j : jobs) delete j;
	    g_us[0] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
	}
}
// the aligner thread: a library call whenever a chunk of complete groups has gathered, and a last one when the input is done
void aligner_loop()
{
	for (;;) {
	    std::vector<Job*> take;
	    bool done;
	    {
		std::unique_lock<std::mutex> lk(g_jm);
		g_jcv.wait(lk, [] { return g_input_done || (int) g_jobs.size() >= g_chunk; });
		take.swap(g_jobs);
		done = g_input_done;
	    }
	    // (beside the mapping workers a call's host side gets a quarter of the threads; the last call gets them all)
	    align_jobs(take, done? std::max(1, (int) thread_num): std::max(1, (int) thread_num / 4));
	    if (done) return;
	}
}
void spaln_gpu_after_workers()
{
	if (g_aligner.joinable()) {
	    {
		std::lock_guard<std::mutex> lk(g_jm);
		for (Pending* p : g_pending) { g_jobs.insert(g_jobs.end(), p->jobs.begin(), p->jobs.end()); p->jobs.clear(); }
		g_input_done = true;
	    }
	    g_jcv.notify_one();
	    g_aligner.join();
	}
	closeGeneRecord();				// what MasterWorker meant to call
}
