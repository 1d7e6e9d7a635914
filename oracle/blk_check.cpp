// blk_check.cpp -- TEST INFRASTRUCTURE ONLY: the product's block-vote routine (spaln_amd/csrc/spdp_blk_core.h, the text the
// device kernel is compiled from) built with the host compiler, so that the tests without a GPU can hold it against the
// reference's recorded runs (tests/test_blk_core.py).  Nothing in the product links this.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../spaln_amd/csrc/spdp_blk_core.h"

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

int blk_check_vote(const BlkIndexC* c, const uint8_t* q, int q_len, int left, int right, int stop_at, int32_t* out, int cap,
                   int touched_cap)
{
    BlkDev ix;
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
}
