// blk_check.cpp -- TEST INFRASTRUCTURE ONLY: the product's block-vote routine (spaln_amd/csrc/spdp_blk_core.h, the text the
// device kernel is compiled from) built with the host compiler, so that the tests without a GPU can hold it against the
// reference's recorded runs (tests/test_blk_core.py).  Nothing in the product links this.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../spaln_amd/csrc/spdp_blk_core.h"
#define SPDP_BLK_DEV_H_          // (interim: the checker still votes with the round-5 routine, which brings its own BlkDev)
#include "../spaln_amd/csrc/spdp_blk_find.h"

extern "C" {
struct BlkIndexC {                       // = BlkIndex of oracle/spdp_oracle_blk.c (oracle/blk.py fills it)
    int32_t nalpha, ktuple, tabsize, nshift, blklen, nbitpat, convts, n_chr, avrscr, maxblk;
    int32_t kk, drna, maxmmc, nseg, minsigpr, ncand, nascr, maxblock, extblock, shortquery;
    int32_t hh_size1, hh_size2, hb_size1, hb_size2, ha_size1, ha_size2, phase1t, gdb, has_chrid, extblockl;
    float rbscoef, rbscons;
    double bclw, bcup, bcce, cfact;
    const uint8_t* convtab; const uint16_t* nblk; const int16_t* wscr; const int32_t* blkp; const uint32_t* blkb;
    const int32_t* rscrtab; const int32_t* chr; const int32_t* bitpat;
};

static void to_dev(const BlkIndexC* c, BlkDev& ix);
int blk_check_vote(const BlkIndexC* c, const uint8_t* q, int q_len, int left, int right, int stop_at, int32_t* out, int cap,
                   int touched_cap)
{
    BlkDev ix;
    to_dev(c, ix);
    std::vector<int32_t> slab(blk_work_ints(ix, touched_cap), 0);
    BlkWork w;
    blk_work_bind(w, ix, slab.data(), touched_cap);
    BlkVote v;
    int calls = 0;
    const int reached = blk_vote_run(ix, w, v, q, q_len, left, right, stop_at, &calls);
    std::vector<BlkPair> bp(ix.ncand + 2);
    std::vector<uint32_t> sw(2 * (2 * ix.ncand + 2) + 2);
    const int n = blk_emit_and_reset(ix, w, v, reached, calls, bp.data(), sw.data(), out, cap);
    for (size_t i = 0; i < 2 * (4 * (size_t) ix.nseg + 2); ++i) if (slab[i]) return -1000000;      // a score slot was left dirty
    return n;
}
static void to_dev(const BlkIndexC* c, BlkDev& ix)
{
    memset(&ix, 0, sizeof ix);
    ix.nalpha = c->nalpha; ix.tabsize = c->tabsize; ix.nshift = c->nshift; ix.nbitpat = c->nbitpat; ix.convts = c->convts;
    ix.n_chr = c->n_chr; ix.kk = c->kk; ix.drna = c->drna; ix.maxmmc = c->maxmmc; ix.nseg = c->nseg; ix.minsigpr = c->minsigpr;
    ix.ncand = c->ncand; ix.nascr = c->nascr; ix.maxblock = c->maxblock; ix.extblock = c->extblock; ix.extblockl = c->extblockl; ix.shortquery = c->shortquery;
    ix.hh_size1 = c->hh_size1; ix.hh_size2 = c->hh_size2; ix.hb_size1 = c->hb_size1; ix.hb_size2 = c->hb_size2;
    ix.ha_size1 = c->ha_size1; ix.ha_size2 = c->ha_size2; ix.gdb = c->gdb;
    ix.rbscoef = c->rbscoef; ix.rbscons = c->rbscons; ix.bclw = c->bclw; ix.bcup = c->bcup; ix.bcce = c->bcce;
    ix.app_c = c->kk > 1 ? pow((double) c->nbitpat, c->cfact) : 1.;
    ix.convtab = c->convtab; ix.nblk = c->nblk; ix.wscr = c->wscr; ix.blkp = c->blkp; ix.blkb = c->blkb;
    ix.rscrtab = c->rscrtab; ix.chr = c->chr; ix.bitpat = c->bitpat;
    for (int k = 0, at = 0; k < c->kk; ++k) { ix.pat_off[k] = at; at += 3 + 2 * c->bitpat[at]; }
    blk_fill_hash_levels(ix);
}

// the block search of one query up to its candidate loci: the product's vote (above) and the product's TestOutput / FindHsp
// (spdp_blk_find.h) with the product's HSP search (spdp_wilip.h), call after call as findblock makes them.  The log has the
// recorder's layout (oracle/ref_build/blk_tap.cc, snap_find): per TestOutput call -4, 0, call, critjscr, n_pairs, pairs x 10,
// n_loci, per locus {chr, sens, base, len, left, right, jscr, CdsNo, 0} + (CdsNo + 1) x 5.  prm: find_prm of the fixture.
int blk_check_find(const BlkIndexC* c, const uint8_t* genome, const int64_t* chr_off, const uint8_t* q, int q_len, int left, int right,
                   const int32_t* prm, const int16_t* intpen, int intpen_len, const SpdpWilipModel* model, int32_t* log, int cap)
{
    BlkDev ix;
    to_dev(c, ix);
    blk_find::Params P;
    P.vthr = prm[0]; memcpy(&P.drop_rate, &prm[1], 4); P.max_out = prm[4]; P.max_out2 = prm[5]; P.bbt = prm[6]; P.min_agap = prm[7];
    P.blklen = prm[8]; P.ext_block = prm[9]; P.ext_block_l = prm[10]; P.phase1t = prm[11]; P.a_exgl = prm[20]; P.a_exgr = prm[21];
    P.dvsp = prm[12]; P.no_retry = prm[2];
    blk_find::Genome G = {genome, chr_off, c->n_chr};
    blk_find::Searcher S;
    S.ix = &ix; S.P = &P; S.G = &G; S.M = model; S.intpen = intpen; S.intpen_len = intpen_len;
    S.gop = prm[13]; S.gep = prm[14]; S.lgop = prm[15]; S.lgep = prm[16]; S.codonk1 = prm[17];
    S.chr_tab = c->chr;
    blk_find::Query Q = {q, q_len, left, right};
    S.q = &Q; S.critjscr = 0;
    const int rcap = 64 + 16 * (4 * c->nseg + 64);
    std::vector<int32_t> rec(rcap);
    int n = 0;
    for (int call = 0; call < 64; ++call) {
        if (blk_check_vote(c, q, q_len, left, right, call, rec.data(), rcap, 1 << 16) < 0) return -2;
        if (!(rec[2] & 1)) break;                        // findblock ended before this call
        int j = 3;
        const int32_t* mmct = rec.data() + j + 4;
        j += 20;
        for (int d = 0; d < 4; ++d) j += 1 + 2 * rec[j];
        const int np = rec[j++];
        std::vector<blk_find::Pair> pairs(np);
        for (int i = 0; i < np; ++i, j += 9)
            pairs[i] = {rec[j], rec[j + 1], 0, (uint32_t) rec[j + 2], (uint32_t) rec[j + 3], (uint32_t) rec[j + 4], (uint32_t) rec[j + 5],
                        (uint32_t) rec[j + 6], (uint32_t) rec[j + 7], rec[j + 8]};
        S.n_runs = rec[j++]; S.runs = rec.data() + j;
        const int res = S.test_output(pairs, mmct, (rec[2] & 8) != 0);
#define PUT(x) do { if (n < cap) log[n] = (x); ++n; } while (0)
        PUT(-4); PUT(0); PUT(call); PUT(S.critjscr); PUT(np);
        for (const blk_find::Pair& b : pairs) {
            PUT(b.bscr); PUT(b.chr); PUT((int) b.lb); PUT((int) b.rb); PUT((int) b.ub); PUT((int) b.db); PUT((int) b.zl); PUT((int) b.zr);
            PUT(b.rvs); PUT(b.jscr);
        }
        const int nl = res > 0 ? res : 0;
        PUT(nl);
        for (int k = 0; k < nl; ++k) {
            const blk_find::Locus& g = S.gener[k];
            const int nh = (int) g.jxt.size() - 1;
            PUT(g.chr); PUT(g.rvs ? 3 : 0); PUT(g.base); PUT(g.len); PUT(g.left); PUT(g.right); PUT(g.jscr); PUT(nh); PUT(0);
            for (const spdp_wl::Juxt& t : g.jxt) { PUT(t.jx); PUT(t.jy); PUT(t.jlen); PUT(t.nid); PUT(t.jscr); }
        }
#undef PUT
        if (res != 0) break;
    }
    return n;
}
}

// the product's Seq::nuc2tron (spdp_blk_find.h) on one sequence, in place: held against the tron codes the reference itself
// made of the protein fixtures' windows (tests/test_blk_find.py)
extern "C" void blk_check_nuc2tron(uint8_t* codes, int len) { blk_find::nuc2tron(codes, len); }
