// spdp_blk_find.h -- what the block search does with a vote: TestOutput's second half and FindHsp (third
// slice; round 5).  Host code of the library; the vote itself is spdp_blk_core.h (device), the HSP search spdp_wilip.h.
//
// Restated (ogotoh/spaln v3.0.7):
//   SrchBlk::TestOutput, second half   src/blksrc.cc:2677-2692   the candidate block pairs, best first, against the random
//                                                                expectation of their mismatch counts; FindHsp on each
//   SrchBlk::FindHsp                   src/blksrc.cc:2346-2545   the region of a pair from the genome (setgnmrng :2004-2013),
//                                                                Wilip at level -1 on it, the units that hold against
//                                                                critjscr / the best one, the pair's ends moved towards
//                                                                what the HSPs leave uncovered, the candidate loci with
//                                                                their overlap / order / pruning rules
// One call of blk_find::test_output = one TestOutput call of one query: in, the vote record of that call (pairs, mismatch
// counts, run scores near the pairs: spdp_blk_vote); out, the candidate loci (region, range, HSPs) or "go on voting".
// critjscr lives across the calls of a query.  Nucleotide queries (PwdB::DvsP = 0) and -- Params::bbt = 3, dvsp = 1 -- protein
// queries against the translated index (-KP): the region is turned into tron codes (Seq::nuc2tron src/seq.cc:774-798 with
// nuc2tron3 src/utilseq.cc:204-225) before the HSP search, and a pair whose ends FindHsp moved is searched again on the
// grown region, NoRetry times at most (:2462-2466).
#ifndef SPDP_BLK_FIND_H_
#define SPDP_BLK_FIND_H_

#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <math.h>
#include "spdp_blk_dev.h"
#include "spdp_wilip.h"
#include "spdp_gencode.h"

namespace blk_find {

// Randbs::randbs (src/blksrc.cc:2064-2069): the score a block reaches by chance after `mmc` rounds
inline int random_expectation(const BlkDev& ix, uint32_t mmc)
{
    if (mmc < 128) return ix.rscrtab[mmc];
    if (ix.rbscoef == 0) return (int) ix.rbscons;
    const double x = (double) (mmc + 1);
    return (int) (ix.rbscoef * (ix.gdb ? log(x) : sqrt(x)) + ix.rbscons);
}


struct Params {                         // statics of src/blksrc.cc and OutPrm
    int vthr;                           // alprm.scale * 2 * alprm.thr (:2210)
    float drop_rate;                    // 1 unless -Xr
    int max_out, max_out2;              // OutPrm.MaxOut, MaxOut2
    int min_agap, bbt;
    int blklen, ext_block, ext_block_l;
    int phase1t;                        // Randbs::Phase1T
    int a_exgl, a_exgr;                 // query->inex.exgl / exgr (Wilip's end bonus)
    int dvsp = 0, no_retry = 0;         // PwdB::DvsP (1: protein query, genomic target), NoRetry (:34)
};
struct Genome {                         // residue codes of the chromosomes, one after the other
    const uint8_t* codes; const int64_t* off; int n_chr;       // chromosome c = codes[off[c] .. off[c + 1])
};
struct Pair { int bscr, chr; int jscr; uint32_t lb, rb, ub, db, zl, zr; int rvs; };
struct Locus {
    int chr, rvs;                       // did, strand
    int base, len;                      // the region cut from the chromosome: [base, base + len) on its forward strand
    int left, right;                    // the range of the region (as the aligner sees it: reverse-complemented when rvs) to align
    int jscr;
    std::vector<spdp_wl::Juxt> jxt;     // CdsNo HSPs + the closing record, coordinates inside the region
    int site(int n) const { return base + (rvs ? len - n : n + 1); }           // Seq::SiteNo
};
struct Query { const uint8_t* codes; int len, left, right; };

inline uint8_t comp_code(uint8_t c)     // complement of a nucleotide code (A 2, C 3, G 5, T 9; src/seq.cc ncredctab / comrev)
{
    switch (c) { case 2: return 9; case 9: return 2; case 3: return 5; case 5: return 3; default: return c; }
}

// Seq::nuc2tron: position p becomes the codon (p - 1, p, p + 1) in the tron alphabet; the sequence's pads stand for the
// residues before the first and behind the last one (first residue ambiguous: the most abundant amino acid of the middle
// one; third residue: ncelements[] of whatever is there)
inline void nuc2tron(uint8_t* s, int len)
{
    static const uint8_t ncredctab[17] = {15, 15, 0, 1, 4, 2, 5, 6, 10, 3, 7, 8, 10, 9, 12, 13, 14};      // src/seq.cc:31
    static const uint8_t ncelements[17] = {0, 0, 0, 1, 2, 2, 0, 2, 0, 3, 3, 3, 1, 1, 2, 3, 0};            // src/seq.cc:33
    static const uint8_t most_abund[4] = {14, 3, 10, 13};                                                   // LYS, ALA, GLY, LEU
    struct Code { uint8_t tron_of[64]; Code() { uint8_t mid[32]; spdp_genetic_code_tables(mid, tron_of); } };
    static const Code code;                                 // (the searchers run on several host threads: initialised once, by the language's rule)
    const uint8_t* tron_of = code.tron_of;
    auto cd = [](uint8_t c) -> int { return c > 16 ? 16 : c; };
    int prev = 0;                                           // the pad
    for (int p = 0; p < len; ++p) {
        const int c0 = prev, c1 = cd(s[p]), c2n = p + 1 < len ? cd(s[p + 1]) : 0;
        prev = c1;
        int aa;
        if (c1 <= 1) aa = 1;                                // UNP: the middle residue is a gap
        else if (ncredctab[c1] >= 4) aa = 2;                // AMB
        else if (ncredctab[c0] >= 4) aa = most_abund[ncredctab[c1]];
        else aa = tron_of[16 * ncredctab[c0] + 4 * ncredctab[c1] + ncelements[c2n]];
        s[p] = (uint8_t) aa;
    }
}

struct Searcher {
    const BlkDev* ix; const Params* P; const Genome* G; const SpdpWilipModel* M;
    const int16_t* intpen; int intpen_len; int gop, gep, lgop, lgep, codonk1;
    const int32_t* chr_tab;             // {spos, first block} x (n_chr + 1), as the index holds them
    // per query
    const Query* q = nullptr;
    int critjscr = 0;
    std::vector<Locus> gener; int curgr = 0;    // gener[0 .. curgr): accepted loci, gener[curgr]: the work slot
    const int32_t* runs = nullptr; int n_runs = 0;      // (block | direction << 28, score) of the vote record
    std::vector<uint8_t> region;

    int chrsize(int c) const { return chr_tab[2 * (c + 1)] - chr_tab[2 * c]; }
    int run_score(int d, uint32_t blk) const
    {
        for (int i = 0; i < n_runs; ++i) if ((uint32_t) runs[2 * i] == (blk | (uint32_t) d << 28)) return runs[2 * i + 1];
        return 0;
    }
    // setgnmrng: the blocks lb .. rb of the pair's chromosome; false: nothing there
    bool cut_region(const Pair& bp, Locus& sd)
    {
        const int64_t clen = G->off[bp.chr + 1] - G->off[bp.chr];
        int64_t x = (int64_t) (bp.zl ? bp.lb - bp.zl : 0) * P->blklen;
        int64_t y = (int64_t) ((bp.zl ? bp.rb - bp.zl : 0) + 1) * P->blklen;
        // getdbseq: "Dbs<c> x+1 y" -- both ends capped at the record's length (src/dbs.cc:839-848)
        if (x + 1 > clen) x = clen - 1;
        if (y > clen) y = clen;
        if (y <= x) return false;
        sd.chr = bp.chr; sd.rvs = bp.rvs; sd.base = (int) x; sd.len = (int) (y - x); sd.left = 0; sd.right = sd.len;
        sd.jscr = 0; sd.jxt.clear();
        region.resize(sd.len + 1);
        const uint8_t* src = G->codes + G->off[bp.chr] + x;
        if (!bp.rvs) memcpy(region.data(), src, sd.len);
        else for (int i = 0; i < sd.len; ++i) region[i] = comp_code(src[sd.len - 1 - i]);
        region[sd.len] = 0;
        return true;
    }
    // 0: nothing that holds, 1: loci made, 2: HSPs but none good enough / no HSP at all
    int find_hsp(Pair& bp)
    {
        const int sr = chrsize(bp.chr);
        const bool rvs = bp.rvs;
        bp.jscr = 0;
        int prv_left = q->right, prv_right = q->left;
        const Pair org = bp;
        const int qlen = q->right - q->left;
        if ((int) gener.size() <= curgr) gener.resize(curgr + 1);
        Locus cursd;
        Pair orgbp = org;
        int retry_no = 0;
        std::vector<spdp_wl::Unit> wl;
        int lcrit = critjscr;
      retry:
        if (!cut_region(bp, cursd)) return 2;
        if (P->bbt == 3) nuc2tron(region.data(), cursd.len);
        const spdp_wl::Pair pr = {q->codes, q->len, q->left, q->right, P->a_exgl, P->a_exgr, region.data(), cursd.len, 0, cursd.len,
                                  P->bbt == 3 ? 3 : 1, nullptr, nullptr, nullptr, intpen, intpen_len, gop, gep, lgop, lgep, codonk1};
        wl.clear();
        spdp_wl::run(M, &pr, -1, wl);
        if (wl.empty()) return 2;
        int n = std::min(P->max_out2, (int) wl.size());
        spdp_wl::Juxt lend = {q->right, 0, 0, 0, 0}, rend = {q->left, 0, 0, 0, 0};
        int nbetter = 0, multi = 0;
        lcrit = critjscr;
        for (int u = 0; u < n; ++u) {
            spdp_wl::Unit& w = wl[u];
            if (w.num > 1) ++multi;
            if (w.tlen > qlen) { const float x = (float) w.scr; w.scr = (int) (x * qlen / w.tlen); }
            if (w.scr >= lcrit) {
                ++nbetter;
                lcrit = P->drop_rate < 1 ? (int) (w.scr * P->drop_rate) : w.scr - P->vthr;
                const spdp_wl::Juxt& f = w.jxt[0];
                if (lend.jx > f.jx) lend = f;
                const spdp_wl::Juxt& l = w.jxt[w.num - 1];
                const int e = l.jx + l.jlen;
                if (rend.jx < e) { rend.jx = e; rend.jy = l.jy + P->bbt * l.jlen; }
            } else {
                for (int v = 0; v < u; ++v) {
                    spdp_wl::Unit& ww = wl[v];
                    if (ww.llmt <= w.ulmt && ww.llmt > w.llmt) ww.llmt = w.llmt;
                    if (ww.ulmt >= w.llmt && ww.ulmt < w.ulmt) ww.ulmt = w.ulmt;
                }
            }
        }
        if (!nbetter) return multi ? 2 : 0;
        const uint32_t EBL = (uint32_t) P->ext_block_l, EB = (uint32_t) P->ext_block;
        if (lend.jx && lend.jx < prv_left) {
            prv_left = lend.jx;
            if (rvs) {
                if (lend.jx <= P->min_agap) bp.rb = bp.db;
                else {
                    bp.rb = std::min(bp.rb + EBL, bp.zr);
                    for ( ; bp.rb > bp.db; --bp.rb) if (run_score(2, bp.rb)) break;
                }
                bp.db = std::min(bp.rb + EB, bp.zr);
            } else {
                if (lend.jx <= P->min_agap) bp.lb = bp.ub;
                else {
                    const uint32_t x = bp.lb > EBL ? bp.lb - EBL : 0;
                    bp.lb = std::max(x, bp.zl);
                    for ( ; bp.lb < bp.ub; ++bp.lb) if (run_score(0, bp.lb)) break;
                }
                const uint32_t x = bp.lb > EB ? bp.lb - EB : 0;
                bp.ub = std::max(x, bp.zl);
            }
        }
        if (q->right > rend.jx && rend.jx > prv_right) {
            prv_right = rend.jx;
            const int dlt = q->right - rend.jx;
            if (rvs) {
                if (dlt == 1 && bp.lb > bp.ub) --bp.lb;
                else if (dlt < P->min_agap) bp.lb = bp.ub;
                else {
                    const uint32_t x = bp.lb > EBL ? bp.lb - EBL : 0;
                    bp.lb = std::max(x, bp.zl);
                    for ( ; bp.lb < bp.ub; ++bp.lb) if (run_score(3, bp.lb)) break;
                }
                const uint32_t x = bp.lb > EB ? bp.lb - EB : 0;
                bp.ub = std::max(x, bp.zl);
            } else {
                if (dlt == 1 && bp.rb < bp.db) ++bp.rb;
                else if (dlt < P->min_agap) bp.rb = bp.db;
                else {
                    bp.rb = std::min(bp.rb + EBL, bp.zr);
                    for ( ; bp.rb > bp.db; --bp.rb) if (run_score(1, bp.rb)) break;
                }
                bp.db = std::min(bp.rb + EB, bp.zr);
            }
        }
        if (P->dvsp == 1 && retry_no++ < P->no_retry && (bp.lb != orgbp.lb || bp.rb != orgbp.rb)) { orgbp = bp; goto retry; }
        int lbias = (int) orgbp.lb - (int) bp.lb, ubias = (int) bp.rb - (int) orgbp.rb;
        if (lbias || ubias) {
            if (!cut_region(bp, cursd)) return 0;
            if (P->bbt == 3) nuc2tron(region.data(), cursd.len);
            if (rvs) std::swap(lbias, ubias);
            if (lbias) {
                const int partial = (rvs && bp.rb == bp.zr) ? (P->blklen - cursd.len % P->blklen) : 0;
                lbias = lbias * P->blklen - partial;
            }
            // Wilip::shift_y(lbias, cursd->len)
            size_t llu = 0;
            for (size_t k = 0; k < wl.size(); ++k) {
                spdp_wl::Unit& w = wl[k];
                if (lbias) {
                    for (int j = 0; j <= w.num; ++j) w.jxt[j].jy += lbias;
                    if (w.llmt) w.llmt += lbias;
                    w.ulmt += lbias;
                }
                if (w.ulmt > wl[llu].ulmt) llu = k;
            }
            wl[llu].ulmt = cursd.len;
            wl[llu].jxt[wl[llu].num].jy = cursd.len;
        }
        const int g_left = 0, g_right = cursd.len;          // cursd->saverange(&grng)
        std::stable_sort(wl.begin(), wl.end(), [](const spdp_wl::Unit& a, const spdp_wl::Unit& b) {
            if (a.scr == b.scr) return b.nid - a.nid < 0;
            return a.scr > b.scr; });
        bp.jscr = wl[0].scr;
        const int lst = P->max_out2;                        // lstgr = gener + MaxOut2
        bool first = true;
        for (size_t u = 0; u < wl.size(); ++u) {
            const spdp_wl::Unit& w = wl[u];
            if (!w.num) break;
            if (w.scr < lcrit) break;
            if (pair_index >= P->max_out && w.scr < critjscr) break;
            if ((int) gener.size() <= curgr) gener.resize(curgr + 1);
            Locus& cg = gener[curgr];
            if (first) { cg = cursd; first = false; }       // (setgnmrng read the region into *curgr)
            else { cg = cursd; cg.left = g_left; cg.right = g_right; }      // aliaseq + restrange
            cg.jscr = w.scr;
            cg.jxt.clear();
            cg.left = w.llmt;
            if (w.ulmt < cg.right) cg.right = w.ulmt;
            const spdp_wl::Juxt& f = w.jxt[0];
            const spdp_wl::Juxt& l = w.jxt[w.num - 1];
            int cl = cg.site(f.jy), cr = cg.site(l.jy + l.jlen);
            if (rvs) std::swap(cl, cr);
            int tl = 0, tr = sr;
            int k = curgr;
            while (--k >= 0) {                              // an accepted locus this one overlaps?
                const Locus& o = gener[k];
                if (o.chr == cg.chr && o.rvs == cg.rvs) {
                    const spdp_wl::Juxt& of = o.jxt[0];
                    const spdp_wl::Juxt& ol = o.jxt[(int) o.jxt.size() - 2];
                    int wlft = o.site(of.jy), wrgt = o.site(ol.jy + ol.jlen);
                    if (rvs) std::swap(wlft, wrgt);
                    if (cr > wlft && cl < wrgt) break;
                    if (cr < wlft && wlft < tr) tr = wlft;
                    if (wrgt < cl && wrgt > tl) tl = wrgt;
                }
            }
            if (k >= 0) continue;
            int at = curgr;                                 // sort on score: the work slot moves up past lower-scoring loci
            while (--at >= 0) {
                if (gener[at + 1].jscr > gener[at].jscr) std::swap(gener[at], gener[at + 1]);
                else break;
            }
            ++at;
            if (at >= lst) break;
            for ( ; curgr > 0; --curgr) {                   // prune low-scoring loci from the end
                if (gener[curgr].jscr >= lcrit) break;
                gener[curgr] = Locus();
            }
            if (curgr >= P->max_out - 1) {
                if ((int) gener.size() > P->max_out - 1) critjscr = gener[P->max_out - 1].jscr - P->vthr;
                if (critjscr < 0) critjscr = 0;
            }
            if (curgr < lst) ++curgr;
            gener[at].jxt.assign(w.jxt.begin(), w.jxt.begin() + w.num + 1);
            if ((int) gener.size() <= curgr) gener.resize(curgr + 1);
            if (curgr == lst) gener[curgr] = Locus();
        }
        return 1;
    }
    int pair_index = 0;                 // wrkbp - bh4->bpair of the FindHsp call in progress

    // TestOutput's second half on the pairs of a vote record.  mmct: Bhit4::mmct of the record; forced: the call is
    // TestOutput(1).  Returns the number of loci (> 0), 0 = go on voting, -1 = the search ends without a locus.
    int test_output(std::vector<Pair>& pairs, const int* mmct, bool forced)
    {
        curgr = 0;
        gener.clear(); gener.resize(1);
        if (pairs.empty()) return forced ? -1 : 0;
        const int force = forced ? 2 : 1;
        int phase1 = 0, nfail = P->max_out2 + 2;
        for (size_t i = 0; i < pairs.size() && nfail; ++i) {
            Pair& bp = pairs[i];
            if (bp.bscr == 0) continue;
            const int d = bp.rvs << 1, e = d + 1;
            if (force != 2 && bp.bscr < random_expectation(*ix, (uint32_t) (mmct[d] + mmct[e])) + P->phase1t) continue;
            pair_index = (int) i;
            switch (find_hsp(bp)) {
                case 1: ++phase1; break;
                case 2: --nfail; break;
                default: break;
            }
        }
        if (phase1) return curgr;
        return force < 2 ? 0 : -1;
    }
};

}   // namespace blk_find
#endif
