// blk_tap.cc -- TEST INFRASTRUCTURE ONLY: records what the reference's block search (SrchBlk::findblock,
// src/blksrc.cc:2971) sees and decides, for the fixtures of SURVEY 8 row f4 (tests/golden/blk_*.spdg).
//
// The reference keeps the search's parameters in file statics of blksrc.cc (wcp, Ncand, MinSigpr, ExtBlock, ...) and its
// state in private members (read here under gcc's -fno-access-control), so this translation unit IS the reference's blksrc.cc -- included below from where it lies
// under /root/reference/src, not copied -- followed by a recorder.  It is compiled with -finstrument-functions: gcc calls
// __cyg_profile_func_enter at the top of every function of the unit, and the recorder acts when the function is
// SrchBlk::findblock (a new query: the index is written once, then the query's codes), SrchBlk::TestOutput (the state of
// the vote at that moment) or SrchBlk::FindHsp (the candidate block pairs TestOutput has just built).  `this` reaches
// the recorder through the constructor: the Makefile renames the constructor symbols of this object (objcopy, as for the
// Wilip tap of ref_dump.cc) and blk_tap_ctor.cc defines the original names around them.  The object code of every
// reference function is what gcc makes of the reference's text; nothing of it is edited.
//
// Linked with the reference's spaln.cc into oracle/_ref/spaln_blktap: the reference's own CLI, run as
//     SPDP_BLK_LOG=out.spdg spaln_blktap -Q7 -t1 -O4 -dgenome queries.fa
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <pthread.h>
#include <sched.h>
#include <sys/types.h>
#include <unistd.h>
#include <zlib.h>

#include "blksrc.cc"		// (compiled with -fno-access-control: the recorder reads private members)
#include "ref_dump_common.h"	// dump_wilip_model: the HSP search FindHsp calls reads these

#define NOINST __attribute__((no_instrument_function))

SrchBlk*	g_blk_tap_this = 0;		// set by blk_tap_ctor.cc

namespace {
struct NOINST_T {};
std::vector<int>	g_log;			// the query log, one int32 stream
bool	g_index_done = false;
bool	g_hsp_armed = false;
FILE*	g_fd = 0;

NOINST void put(const char* name, unsigned dtype, const void* p, size_t cnt)
{
	static const int esz[5] = {0, 1, 2, 4, 1};
	char	nm[32];
	memset(nm, 0, sizeof(nm));
	strncpy(nm, name, 31);
	fwrite(nm, 1, 32, g_fd);
	unsigned	hd[2] = {dtype, (unsigned) cnt};
	fwrite(hd, 4, 2, g_fd);
	size_t	nb = cnt * esz[dtype];
	if (nb) fwrite(p, 1, nb, g_fd);
	static const char zero[8] = {0};
	if (nb % 8) fwrite(zero, 1, 8 - nb % 8, g_fd);
}
NOINST void put_i32(const char* name, const int* v, size_t n) { put(name, 3, v, n); }

std::vector<int>	g_find;			// second slice (round 5): what TestOutput leaves behind -- the candidate loci FindHsp made
int	g_qno = -1, g_call = 0;
bool	g_find_prm_done = false;

NOINST void at_exit_flush()
{
	if (!g_fd) return;
	put("find_log", 3, g_find.empty()? 0: &g_find[0], g_find.size());
	put("q_log", 3, g_log.empty()? 0: &g_log[0], g_log.size());
	fclose(g_fd);
	g_fd = 0;
}

NOINST void open_log()
{
	if (g_fd) return;
const	char*	fn = getenv("SPDP_BLK_LOG");
	if (!fn) return;
	g_fd = fopen(fn, "wb");
	if (!g_fd) { perror(fn); exit(1); }
	fwrite("SPDG1\0\0\0", 1, 8, g_fd);
	atexit(at_exit_flush);
}

NOINST int f2i(float f) { int i; memcpy(&i, &f, 4); return i; }

NOINST void dump_index(SrchBlk* s)
{
	ContBlk*	wc = s->pbwc;
	int	kk = s->kk;
	// hash geometries the reference's containers end up with (Dhash sizes are functions of the requested size)
	Dhash<INT, int>	hh(2 * wc->MaxBlk, 0);
	Dhash<int, int>	hb(2 * Ncand, -1);
	Dhash<int, int>	ha(2 * Nascr, -1);
	double	pb2c[3] = {s->pb2c->BClw, s->pb2c->BCup, s->pb2c->BCce};
	double	cf = cfact;
	int	prm[64];
	int	n = 0;
	prm[n++] = wcp.Nalpha; prm[n++] = wcp.Ktuple; prm[n++] = wcp.Bitpat2; prm[n++] = wcp.TabSize;	// 0..3
	prm[n++] = wcp.BitPat; prm[n++] = wcp.Nshift; prm[n++] = wcp.blklen; prm[n++] = wcp.MaxGene;	// 4..7
	prm[n++] = wcp.Nbitpat; prm[n++] = wcp.afact;							// 8, 9
	prm[n++] = wc->ConvTS; prm[n++] = (int) wc->WordNo; prm[n++] = (int) wc->ChrNo;		// 10..12
	prm[n++] = wc->AvrScr; prm[n++] = wc->MaxBlk;							// 13, 14
	prm[n++] = kk; prm[n++] = s->DRNA; prm[n++] = (int) s->maxmmc; prm[n++] = s->ptpl;		// 15..18
	prm[n++] = s->nseg; prm[n++] = s->bbt; prm[n++] = s->min_agap;					// 19..21
	prm[n++] = MinSigpr; prm[n++] = Ncand; prm[n++] = Nascr;					// 22..24
	prm[n++] = MaxBlock; prm[n++] = ExtBlock; prm[n++] = ExtBlockL; prm[n++] = shortquery;		// 25..28
	prm[n++] = hh.size(); prm[n++] = hh.size2; prm[n++] = hb.size(); prm[n++] = hb.size2;		// 29..32
	prm[n++] = ha.size(); prm[n++] = ha.size2;							// 33, 34
	prm[n++] = s->rdbt->Phase1T; prm[n++] = f2i(s->rdbt->RbsCoef); prm[n++] = f2i(s->rdbt->RbsCons);	// 35..37
	prm[n++] = s->gnmdb? 1: 0; prm[n++] = OutPrm.MaxOut; prm[n++] = OutPrm.MaxOut2;		// 38..40
	prm[n++] = wc->ChrID? 1: 0;									// 41
	put_i32("blk_prm", prm, n);
	put("blk_pb2c", 1, pb2c, sizeof pb2c);
	put("blk_cfact", 1, &cf, sizeof cf);
	put("blk_rscrtab", 3, s->rdbt->rscrtab, NRTAB);
	put("blk_nblk", 2, wc->Nblk, wcp.TabSize);
	put("blk_wscr", 2, wc->wscr, wcp.TabSize);
	std::vector<int>	off(wcp.TabSize);
	for (INT w = 0; w < wcp.TabSize; ++w) off[w] = wc->blkp[w]? (int) (wc->blkp[w] - wc->blkb) + 1: 0;
	put_i32("blk_blkp", &off[0], off.size());
	put("blk_blkb", 3, wc->blkb, wc->WordNo);
	put("blk_convtab", 1, s->ConvTab, wc->ConvTS);
	std::vector<int>	chr;
	for (size_t c = 0; c <= wc->ChrNo; ++c) {
	    chr.push_back(wc->ChrID? (int) wc->ChrID[c].spos: 0);
	    chr.push_back((int) s->chrblk((int) c));
	}
	put_i32("blk_chr", &chr[0], chr.size());
	std::vector<int>	bp;
	for (int k = 0; k < kk; ++k) {
	    Bitpat*	b = s->bpp[k];
	    bp.push_back(b->weight); bp.push_back(b->width); bp.push_back(b->wshift);
	    for (int i = 0; i < 2 * b->weight; ++i) bp.push_back(b->exam? b->exam[i]: i % b->weight);
	}
	put_i32("blk_bitpat", &bp[0], bp.size());
}

NOINST void snap_vote(SrchBlk* s)
{
	Bhit4*	b = s->bh4;
	g_log.push_back(-2);					// record: state of the vote at a TestOutput call
	for (int d = 0; d < 4; ++d) g_log.push_back(b->sign[d]);
	for (int d = 0; d < 4; ++d) g_log.push_back(b->mmct[d]);
	for (int d = 0; d < 4; ++d) g_log.push_back(b->nhit[d]);
	for (int d = 0; d < 4; ++d) g_log.push_back(b->maxs[d]);
	for (int d = 0; d < 4; ++d) g_log.push_back((int) b->testword[d]);
	for (int d = 0; d < 4; ++d) {
	    int	nb = b->prqueue_b[d]->size();
	    g_log.push_back(nb);
	    for (int i = 0; i < nb; ++i) { g_log.push_back((*b->prqueue_b[d])[i].key); g_log.push_back((*b->prqueue_b[d])[i].bscr); }
	    int	na = b->prqueue_a[d]->size();
	    g_log.push_back(na);
	    for (int i = 0; i < na; ++i) { g_log.push_back((*b->prqueue_a[d])[i].key); g_log.push_back((*b->prqueue_a[d])[i].bscr); }
	    size_t	at = g_log.size();
	    g_log.push_back(0);
	    for (INT x = 0; x < s->nseg; ++x)
		if (b->bscr[d][x]) { g_log.push_back(x); g_log.push_back(b->bscr[d][x]); ++g_log[at]; }
	    at = g_log.size();
	    g_log.push_back(0);
	    for (INT x = 0; x < s->nseg; ++x)
		if (b->ascr[d][x]) { g_log.push_back(x); g_log.push_back(b->ascr[d][x]); ++g_log[at]; }
	}
}

NOINST void snap_pairs(SrchBlk* s)
{
	Bhit4*	b = s->bh4;
	g_log.push_back(-3);					// record: the block pairs TestOutput built (Ncand + 1 slots)
	g_log.push_back(Ncand + 1);
	for (int i = 0; i <= Ncand; ++i) {
	    const BPAIR&	p = b->bpair[i];
	    const int	v[9] = {p.bscr, p.chr, (int) p.lb, (int) p.rb, (int) p.ub, (int) p.db, (int) p.zl, (int) p.zr, (int) p.rvs};
	    g_log.insert(g_log.end(), v, v + 9);
	}
}
}	// namespace

struct TapWriter { NOINST void put_i32(const char* name, const std::vector<int>& v) { put(name, 3, v.empty()? 0: &v[0], v.size()); } };

// the parameters FindHsp / TestOutput's second half read besides the index and the vote (src/blksrc.cc:2346-2545, 2677-2692)
NOINST void dump_find_prm(SrchBlk* s)
{
	int	dr; memcpy(&dr, &drop_rate, 4);
	std::vector<int> v = {(int) s->vthr, dr, (int) NoRetry, gene_rng_max_extend, (int) OutPrm.MaxOut, (int) OutPrm.MaxOut2, s->bbt,
	    s->min_agap, (int) wcp.blklen, (int) ExtBlock, (int) ExtBlockL, s->rdbt->Phase1T, (int) s->pwd->DvsP,
	    (int) s->pwd->BasicGOP, (int) s->pwd->BasicGEP, (int) s->pwd->LongGOP, (int) s->pwd->LongGEP, s->pwd->codonk1,
	    (int) algmode.nsa, (int) algmode.slv, (int) s->query->inex.exgl, (int) s->query->inex.exgr};
	put_i32("find_prm", &v[0], v.size());
	std::vector<short> ip(1 << 19);		// (every length a candidate region can hold: no caller has to extend the table by formula)
	for (size_t n = 0; n < ip.size(); ++n) ip[n] = (short) s->pwd->IntPen->Penalty((int) n);
	put("find_intpen", 2, &ip[0], ip.size());
	// the intron-length limits as THIS run holds them: the program derives maxl from the length model's 99 % quantile and minl from where
	// an intron starts to beat a gap (src/codepot.cc:134-135, 176-212); a harness that sets its own defaults holds other values
	std::vector<int> cp = {IntronPrm.llmt, IntronPrm.minl, IntronPrm.rlmt, IntronPrm.maxl, IntronPrm.nquant, (int) IntronPrm.hard_minl,
	    (int) IntronPrm.hard_maxl, IntronPrm.elmt, IntronPrm.tlmt, IntronPrm.mode};
	put_i32("cli_intron_prm", &cp[0], cp.size());
	TapWriter tw;
	dump_wilip_model(tw, s->pwd);
}

// TestOutput is over: the pairs as FindHsp left them, critjscr, and the candidate loci [gener, curgr) with their HSPs
NOINST void snap_find(SrchBlk* s)
{
	Bhit4*	b = s->bh4;
	g_find.push_back(-4); g_find.push_back(g_qno); g_find.push_back(g_call++);
	g_find.push_back((int) s->critjscr);
	g_find.push_back(Ncand + 1);
	for (int i = 0; i <= Ncand; ++i) {
	    const BPAIR&	p = b->bpair[i];
	    const int	v[10] = {p.bscr, p.chr, (int) p.lb, (int) p.rb, (int) p.ub, (int) p.db, (int) p.zl, (int) p.zr, (int) p.rvs, (int) p.jscr};
	    g_find.insert(g_find.end(), v, v + 10);
	}
	int	n = (int) (s->curgr - s->gener);
	if (n < 0 || n > (int) OutPrm.MaxOut2) n = 0;
	g_find.push_back(n);
	for (int k = 0; k < n; ++k) {
	    Seq*	g = s->gener[k];
	    const int	hd[9] = {g->did, (int) g->inex.sens, g->base_, g->len, g->left, g->right, (int) g->jscr, g->jxt? g->CdsNo: 0, g->wllvl};
	    g_find.insert(g_find.end(), hd, hd + 9);
	    for (int j = 0; g->jxt && j <= g->CdsNo; ++j) {
		const JUXT& t = g->jxt[j];
		const int jr[5] = {t.jx, t.jy, t.jlen, t.nid, (int) t.jscr};
		g_find.insert(g_find.end(), jr, jr + 5);
	    }
	}
}

extern "C" NOINST void __cyg_profile_func_enter(void* fn, void*)
{
	static void* const	f_find = (void*) (&SrchBlk::findblock);
	static void* const	f_test = (void*) (&SrchBlk::TestOutput);
	static void* const	f_hsp = (void*) (&SrchBlk::FindHsp);
	if (fn != f_find && fn != f_test && fn != f_hsp) return;
	SrchBlk*	s = g_blk_tap_this;
	if (!s || !s->bh4) return;
	open_log();
	if (!g_fd) return;
	if (fn == f_find) {
	    if (!g_index_done) { dump_index(s); g_index_done = true; }
	    if (!g_find_prm_done && s->pwd) { dump_find_prm(s); g_find_prm_done = true; }
	    ++g_qno; g_call = 0;
	    Seq*	q = s->query;
	    g_log.push_back(-1);				// record: a query enters findblock
	    g_log.push_back(q->left); g_log.push_back(q->right); g_log.push_back(q->len);
	    for (int i = 0; i < q->len; ++i) g_log.push_back(*q->at(i));
	    g_hsp_armed = false;
	} else if (fn == f_test) {
	    snap_vote(s);
	    g_hsp_armed = true;
	} else if (g_hsp_armed) {
	    snap_pairs(s);
	    g_hsp_armed = false;
	}
}
extern "C" NOINST void __cyg_profile_func_exit(void* fn, void*)
{
	static void* const	f_test = (void*) (&SrchBlk::TestOutput);
	if (fn != f_test || !g_fd) return;
	SrchBlk*	s = g_blk_tap_this;
	if (s && s->bh4) snap_find(s);
}
