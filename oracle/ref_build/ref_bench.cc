// ref_bench -- the compiled reference as a CPU baseline without one process start per query.  TEST / BENCH
// INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg); our own driver, no reference code in it.
//
// spaln's CLI aligns every query against ONE genomic entry (or maps through a block index first), so a kernel
// benchmark whose queries each come with their own locus window had to start the binary once per pair -- and paid
// the parameter tables' load and the process start 512 times.  This driver loads the tables once, then T threads
// take (window, query) pairs from a shared counter and run exactly what `spaln -Q0 -A<alg>` runs per pair
// (match_2, src/spaln.cc:734-790: exg_seq, Exinon on the window, alignS_ng with a fixed orientation or alignH_ng,
// then skl_rngS_ng / skl_rngH_ng), as the reference's own -t workers do with whole queries (src/spaln.cc:1389-1468).
// Prints: pairs done, pairs with an alignment, wall seconds of the parallel section.
//
// usage: ref_bench -A alg -t threads list.txt      (list.txt: one "window.fa query.fa" per line)
#include "ref_dump_common.h"
#include <atomic>
#include <chrono>
#include <thread>

int main(int argc, const char** argv)
{
	int	alg = 2, nthr = 1;
	int	ai = 1;
	for ( ; ai < argc && argv[ai][0] == '-'; ++ai) {
	    if (argv[ai][1] == 'A') alg = atoi(argv[++ai]);
	    else if (argv[ai][1] == 't') nthr = atoi(argv[++ai]);
	    else { fprintf(stderr, "bad option %s\n", argv[ai]); return 1; }
	}
	if (ai >= argc) { fprintf(stderr, "usage: ref_bench -A alg -t threads list.txt\n"); return 1; }
	std::vector<std::pair<std::string, std::string>> pairs;
	{
	    FILE* f = fopen(argv[ai], "r");
	    if (!f) { perror(argv[ai]); return 1; }
	    char g[1024], q[1024];
	    while (fscanf(f, "%1023s %1023s", g, q) == 2) pairs.push_back({g, q});
	    fclose(f);
	}
	if (pairs.empty()) return 1;
	set_default_params();
	optimize(GLOBAL, MAXIMUM);
	algmode.qck = 0;		// -Q0
	algmode.blk = 0;
	OutPrm.all_out = 1;
	// the shared, read-only set-up of a run: from the first pair (all pairs are of one kind)
	PwdB*	pwd = 0;
	bool	protein = false;
	{
	    Seq*	seqs[4];
	    initseq(seqs, 4);
const	    char*	files[2] = {pairs[0].first.c_str(), pairs[0].second.c_str()};
	    SeqServer	svr(2, files, IM_SNGL, 0, UNKNOWN, UNKNOWN);
	    if (svr.nextseq(seqs[1], 1) == IS_END || svr.nextseq(seqs[0], 0) != IS_OK) return 1;
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
	std::atomic<int>	next(0), done(0), aligned(0);
	auto worker = [&]() {
	    Seq*	seqs[4];
	    initseq(seqs, 4);
	    for (;;) {
const		int	k = next.fetch_add(1);
		if (k >= (int) pairs.size()) break;
const		char*	files[2] = {pairs[k].first.c_str(), pairs[k].second.c_str()};
		SeqServer	svr(2, files, IM_SNGL, 0, UNKNOWN, UNKNOWN);
		Seq*&	a = seqs[0];
		Seq*&	b = seqs[1];
		if (svr.nextseq(b, 1) == IS_END || svr.nextseq(a, 0) != IS_OK) continue;
		b->inex.intr = algmode.lsg;
		a->inex.intr = 0;
		a->inex.ori = 1;
		if (protein) b->nuc2tron();
		a->exg_seq(algmode.lcl & 4, algmode.lcl & 8);
		b->exg_seq(algmode.lcl & 1, algmode.lcl & 2);
		delete b->exin;
		b->exin = new Exinon(b, pwd, false);
		Gsinfo	gsi;
		gsi.skl = protein? alignH_ng((const Seq**) seqs, pwd, &gsi): alignS_ng(seqs, pwd, &gsi, 1);
		if (gsi.skl && gsi.skl->n) {
		    (void) (protein? skl_rngH_ng((const Seq**) seqs, &gsi, pwd): skl_rngS_ng((const Seq**) seqs, &gsi, pwd));
		    ++aligned;
		}
		++done;
	    }
	    clearseq(seqs, 4);
	};
	const auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> pool;
	for (int t = 0; t < nthr; ++t) pool.emplace_back(worker);
	for (auto& t : pool) t.join();
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("%d %d %.6f\n", done.load(), aligned.load(), dt);
	return 0;
}
