// dumpq.cc -- spaln_dumpq: the reference's own command line program (src/spaln.cc, linked as it is) with ONE of its
// alignH_ng calls recorded as a golden fixture (TEST INFRASTRUCTURE ONLY; build container only).
//
// When a difference between the drop-in and the reference shows up in a run of thousands of queries
// (tools/dropin_demo.py), the pair the aligner saw -- the window blkaln cut, its Exinon, the HSPs the block search left in
// b->jxt -- exists only inside that run.  This program is the reference with its protein aligner wrapped: for the query
// named in DUMPQ_NAME (call number DUMPQ_CALL of that query, default 0) it writes the fixture ref_dump -Q would write
// (ref_dump_h.cc, dump_protein_body: inputs, signal arrays, HSPs, every Wilip reply, score + SKL of the reference's own
// alignH_ng under the program's -A) to DUMPQ_OUT and goes on.  Same objects as oracle/_ref/spaln except src/fwd2h1.cc compiled
// with its entry renamed and wln.o with the constructor tap of ref_dump.cc.
#include <mutex>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <vector>
#include "ref_dump_common.h"
#include "shim_fill.h"		// integration/: what the reference-side binding makes of the same objects (written beside the fixture)

extern "C" void ref_wilip_ctor(Wilip* self, const Seq** seqs, const PwdB* pwd, int level);
bool	g_o12_mode = false;
char	g_o12_prefix[256];
int	g_seeded_q = 0;
std::vector<int>	g_alg_list;
bool		wilip_tap_on = false;
std::vector<int>	wilip_tap_log;
static thread_local bool t_taping = false;	// only the thread that dumps records
Wilip::Wilip(const Seq* seqs[], const PwdB* pwd, const int level)
{
	ref_wilip_ctor(this, seqs, pwd, level);
	if (!wilip_tap_on || !t_taping) return;
	std::vector<int>& L = wilip_tap_log;
	const int hd[6] = {level + (seqs[0]->inex.sens? 16: 0), seqs[0]->left, seqs[0]->right, seqs[1]->left, seqs[1]->right, wlu? nwlu: 0};
	L.insert(L.end(), hd, hd + 6);
	for (int u = 0; wlu && u < nwlu; ++u) {
	    const WLUNIT& x = wlu[u];
	    const int uh[6] = {x.num, x.nid, x.tlen, x.llmt, x.ulmt, (int) x.scr};
	    L.insert(L.end(), uh, uh + 6);
	    for (int j = 0; j <= x.num; ++j) {
		const JUXT& t = x.jxt[j];
		const int jr[5] = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		L.insert(L.end(), jr, jr + 5);
	    }
	}
}

static void on_segv(int) { void* bt[48]; int n = backtrace(bt, 48); backtrace_symbols_fd(bt, n, 2); _exit(139); }
extern SKL* alignH_ng_ref(const Seq* seqs[], const PwdB* pwd, Gsinfo* gsi);
SKL* alignH_ng(const Seq* seqs[], const PwdB* pwd, Gsinfo* gsi)
{
	static const char* want = getenv("DUMPQ_NAME");
	static const char* out = getenv("DUMPQ_OUT");
	static const int call = getenv("DUMPQ_CALL")? atoi(getenv("DUMPQ_CALL")): 0;
	static std::mutex mtx;
	static int seen = 0;
	if (want && out && !strcmp(seqs[0]->sqname(), want)) {
	    std::lock_guard<std::mutex> lk(mtx);
	    if (seen++ == call) {
		t_taping = true;
		signal(SIGSEGV, on_segv);
		g_seeded_q = algmode.qck;
		const int alg0 = algmode.alg;
		const Seq* b = seqs[1];
		fprintf(stderr, "[dumpq] %s call %d: a %d..%d of %d, b %d..%d of %d (%s sens %d), %d HSPs, qck %d alg %d\n", want, call,
		    seqs[0]->left, seqs[0]->right, seqs[0]->len, b->left, b->right, b->len, b->sqname(), (int) b->inex.sens,
		    b->jxt? b->CdsNo: 0, (int) algmode.qck, alg0);
		std::vector<int> none;
		g_alg_list.assign(1, alg0);		// (the PwdB is the program's: its IntPen has no quantiles unless the program runs -A2 / -A3)
		dump_protein_body((Seq**) seqs, (PwdB*) pwd, none, out);
		{   // the binding's view of the pair (integration/shim_fill.h, as spaln_gpu_shim.cc's record_job calls it): <out>.shim
		    SpdpScoringH sc; SpdpProblemH ph; HCols hc; SeedCols c; SpdpSeedParams sp;
		    fill_scoring_h(sc, pwd, b);
		    fill_problem_h(ph, seqs[0], b, hc, b->left, b->right);
		    ph.a_pad = *seqs[0]->at(seqs[0]->len);
		    fill_exact_h(sc, ph, b, pwd, c, true);
		    fill_seed_params(sp, pwd, b);
		    Writer w((std::string(out) + ".shim").c_str());
		    const int N = b->len + 3;
		    w.put("sig5", 2, ph.sig5, N); w.put("sig3", 2, ph.sig3, N); w.put("sigS", 2, ph.sigS, N); w.put("sigT", 2, ph.sigT, N);
		    w.put("sigE", 2, ph.sigE, N); w.put("phs5", 4, ph.phs5, N); w.put("phs3", 4, ph.phs3, N);
		    w.put("dinc", 1, ph.dinc, N); w.put("a", 1, ph.a, ph.a_len + 1); w.put("b", 1, ph.b, ph.b_len + 1);
		    w.put("t53", 2, sc.t53, 256); w.put("intpen", 2, sc.intpen, sc.intpen_len);
		    w.put("mtx", 3, sc.mtx, sizeof sc.mtx / 4);
		    std::vector<int> pr = {ph.a_len, ph.b_len, ph.a_left, ph.a_right, ph.b_left, ph.b_right, ph.a_exgl, ph.a_exgr, ph.b_exgl, ph.b_exgr,
			ph.exin_left, ph.exin_right, ph.a_pad};
		    w.put_i32("problem", pr);
		    SpdpScoringH sc0 = sc; sc0.intpen = 0;
		    w.put("scoring_bytes", 1, &sc0, sizeof sc0);
		    w.put("seed_bytes", 1, &sp, sizeof sp);
		}
		algmode.alg = alg0;
		t_taping = false;
		{   // every Wilip call the reference's walk made, once more through the binding's callback (wilip_flat) on the same pair
		    const std::vector<int> L = wilip_tap_log;
		    std::vector<int32_t> flat;
		    for (size_t at = 0; at + 6 <= L.size(); ) {
			const int32_t span[8] = {L[at + 1], L[at + 2], L[at + 3], L[at + 4], 0, 0, 0, 0};
			const int level = L[at] & 15, nu = L[at + 5];
			size_t e = at + 6;
			for (int u = 0; u < nu; ++u) e += 6 + 5 * (L[e] + 1);
			wilip_flat((Seq**) seqs, pwd, level, span, flat);
			bool same = (int) flat.size() == 1 + (int) (e - at - 6) && flat[0] == nu;
			for (size_t k = 0; same && k < e - at - 6; ++k) same = flat[1 + k] == L[at + 6 + k];
			fprintf(stderr, "[dumpq] Wilip level %d a %d..%d b %d..%d: %d unit(s) logged, callback gives %d -> %s\n", level, span[0], span[1], span[2],
			    span[3], nu, flat.empty()? -1: flat[0], same? "same": "DIFFERENT");
			if (!same) { for (int x : flat) fprintf(stderr, " %d", x); fprintf(stderr, "\n"); }
			at = e;
		    }
		}
	    }
	}
	return alignH_ng_ref(seqs, pwd, gsi);
}
