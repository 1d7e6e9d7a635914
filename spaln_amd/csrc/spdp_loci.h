// spdp_loci.h -- from a vote of the block search to candidate loci: what the reference's TestOutput does with its list of block
// pairs and what FindHsp does with each of them (ogotoh/spaln v3.0.7 src/blksrc.cc: TestOutput's second half :2677-2692, FindHsp
// :2346-2545, setgnmrng :2294-2344, Wilip::shift_y src/wln.cc:994-1010).  Host code of the library.
//
// The reference works through a query's pairs one after the other, searching each pair's region for HSPs on the spot.  Here the
// search is a DEVICE BATCH over the pairs of thousands of queries (spdp_hsp.hip), so the per-query logic is written as a machine
// that is advanced with answers: it names the searches it will need (all its pairs' first regions are known up front and go into
// one batch), takes the answers in the reference's order, and stops only where a decision depends on an answer it has not been
// given -- a protein query's second look at a region whose ends it has just moved.  Between the batches the machines of all
// queries advance on the host threads.
//
//   Verdict      which of a search's units hold (against the query's current cut-off), how much of the query's ends they leave
//   move_ends    the pair's block range pulled towards what the units leave uncovered (run scores of the vote as sign posts)
//   relocate     the units' coordinates in the region as re-cut
//   admit        the units that hold become loci: overlap with loci already taken, order by score, the list's bounds
#ifndef SPDP_LOCI_H_
#define SPDP_LOCI_H_

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "spdp_hsp_chain.h"

namespace spdp_loci {

using spdp_hsp::Hsp;
using spdp_hsp::Unit;

struct Params {                         // statics of src/blksrc.cc and OutPrm
    int vthr;                           // alprm.scale * 2 * alprm.thr (:2210)
    float drop_rate;                    // 1 unless -Xr
    int max_out, max_out2;              // OutPrm.MaxOut, MaxOut2
    int min_agap, bbt;
    int blklen, ext_block, ext_block_l;
    int phase1t;                        // Randbs::Phase1T
    int a_exgl, a_exgr;                 // query->inex.exgl / exgr (the HSP search's end bonus)
    int dvsp = 0, no_retry = 0;         // PwdB::DvsP (1: protein query, genomic target), NoRetry (:34)
};
struct RandomScore {                    // Randbs::randbs (:2064-2069): what a block reaches by chance after `mmc` rounds
    const int32_t* table; float coef, cons; int gdb;
    int of(uint32_t mmc) const
    {
        if (mmc < 128) return table[mmc];
        if (coef == 0) return (int) cons;
        const double x = (double) (mmc + 1);
        return (int) (coef * (gdb ? log(x) : sqrt(x)) + cons);
    }
};
struct Chromosomes { const int64_t* off; int n; const int32_t* tab; };     // residue offsets; {spos, first block} x (n + 1) of the index
struct Pair { int bscr, chr, jscr; uint32_t lb, rb, ub, db, zl, zr; int rvs; };
struct Region { int chr, rvs, base, len; };             // [base, base + len) of the chromosome's forward strand, read as the other strand when rvs
struct Locus {
    Region at; int left, right, jscr;                   // the range of the region (in the orientation of the search) to align
    std::vector<Hsp> hsp;                               // CdsNo HSPs + the closing record, coordinates inside the region
    int site(int n) const { return at.base + (at.rvs ? at.len - n : n + 1); }       // Seq::SiteNo
};

// the residues of blocks lb .. rb of the pair's chromosome (setgnmrng; getdbseq caps both ends at the record's length, src/dbs.cc:839-848)
inline bool region_of(const Pair& bp, const Chromosomes& G, int blklen, Region& r)
{
    const int64_t clen = G.off[bp.chr + 1] - G.off[bp.chr];
    int64_t x = (int64_t) (bp.zl ? bp.lb - bp.zl : 0) * blklen, y = (int64_t) ((bp.zl ? bp.rb - bp.zl : 0) + 1) * blklen;
    if (x + 1 > clen) x = clen - 1;
    if (y > clen) y = clen;
    if (y <= x) return false;
    r = Region{bp.chr, bp.rvs, (int) x, (int) (y - x)};
    return true;
}

struct Query { int len, left, right; };

// One TestOutput call of one query.
struct Call {
    // ---- what the call is given
    const Params* P; const Chromosomes* G; RandomScore rnd;
    Query q;
    std::vector<Pair> pairs;
    int mmct[4]; bool forced;
    std::vector<std::pair<uint32_t, int>> runs;         // (block | direction << 28, run score), sorted
    int critjscr = 0;                                   // the query's cut-off, carried from call to call
    // ---- where it stands
    std::vector<Locus> loci;                            // taken so far, best first
    size_t at_pair = 0; int n_held = 0, n_fail = 0;
    bool in_pair = false;
    Pair bp, seen; int tries = 0, left_seen = 0, right_seen = 0;
    Region want;                                        // the region whose units the machine waits for
    int result = 0; bool done = false;                  // > 0 loci, 0 go on voting, -1 the search ends without a locus

    int run_score(int d, uint32_t blk) const
    {
        const uint32_t key = blk | (uint32_t) d << 28;
        auto it = std::lower_bound(runs.begin(), runs.end(), std::make_pair(key, INT32_MIN));
        return it != runs.end() && it->first == key ? it->second : 0;
    }
    bool eligible(const Pair& p) const
    {
        if (p.bscr == 0) return false;
        const int d = p.rvs << 1;
        return forced || p.bscr >= rnd.of((uint32_t) (mmct[d] + mmct[d + 1])) + P->phase1t;
    }
    void begin()
    {
        loci.clear(); at_pair = 0; n_held = 0; n_fail = P->max_out2 + 2; in_pair = false; done = false;
        std::sort(runs.begin(), runs.end());
        if (pairs.empty()) { result = forced ? -1 : 0; done = true; }
    }
    // the regions every eligible pair starts with: known before any answer (the batch's first round)
    void first_regions(std::vector<std::pair<int, Region>>& out) const
    {
        for (size_t i = 0; i < pairs.size(); ++i) {
            Region r;
            if (eligible(pairs[i]) && region_of(pairs[i], *G, P->blklen, r)) out.emplace_back((int) i, r);
        }
    }

    // ---- a search's units against the query's cut-off
    struct Verdict { int n_hold = 0, n_chains = 0, cut; Hsp head, tail; };
    Verdict weigh(std::vector<Unit>& wl) const
    {
        Verdict v;
        v.cut = critjscr;
        v.head = Hsp{q.right, 0, 0, 0, 0}; v.tail = Hsp{q.left, 0, 0, 0, 0};
        const int qlen = q.right - q.left, n = std::min(P->max_out2, (int) wl.size());
        for (int u = 0; u < n; ++u) {
            Unit& w = wl[u];
            if (w.num > 1) ++v.n_chains;
            if (w.tlen > qlen) w.scr = (int) ((float) w.scr * qlen / w.tlen);          // more HSP than query: pro rata
            if (w.scr >= v.cut) {
                ++v.n_hold;
                v.cut = P->drop_rate < 1 ? (int) (w.scr * P->drop_rate) : w.scr - P->vthr;
                if (v.head.jx > w.hsp[0].jx) v.head = w.hsp[0];
                const Hsp& l = w.hsp[w.num - 1];
                if (v.tail.jx < l.jx + l.jlen) { v.tail.jx = l.jx + l.jlen; v.tail.jy = l.jy + P->bbt * l.jlen; }
                continue;
            }
            for (int k = 0; k < u; ++k) {               // a unit that fails gives its claim to the better ones it touches
                Unit& b = wl[k];
                if (b.llmt <= w.ulmt && b.llmt > w.llmt) b.llmt = w.llmt;
                if (b.ulmt >= w.llmt && b.ulmt < w.ulmt) b.ulmt = w.ulmt;
            }
        }
        return v;
    }
    // ---- the pair's ends towards what the units leave of the query's ends.  `low`: the end at the smaller block numbers
    void pull(bool low, int d, bool all_the_way)
    {
        const uint32_t far = (uint32_t) P->ext_block_l, margin = (uint32_t) P->ext_block;
        if (low) {
            if (all_the_way) bp.lb = bp.ub;
            else {
                bp.lb = std::max(bp.lb > far ? bp.lb - far : 0u, bp.zl);
                while (bp.lb < bp.ub && !run_score(d, bp.lb)) ++bp.lb;
            }
            bp.ub = std::max(bp.lb > margin ? bp.lb - margin : 0u, bp.zl);
        } else {
            if (all_the_way) bp.rb = bp.db;
            else {
                bp.rb = std::min(bp.rb + far, bp.zr);
                while (bp.rb > bp.db && !run_score(d, bp.rb)) --bp.rb;
            }
            bp.db = std::min(bp.rb + margin, bp.zr);
        }
    }
    void move_ends(const Verdict& v)
    {
        const bool rvs = bp.rvs;
        // the query's head is upstream: the low end of a forward pair, the high end of a reverse one
        if (v.head.jx && v.head.jx < left_seen) {
            left_seen = v.head.jx;
            pull(!rvs, rvs ? 2 : 0, v.head.jx <= P->min_agap);
        }
        if (q.right > v.tail.jx && v.tail.jx > right_seen) {
            right_seen = v.tail.jx;
            const int rest = q.right - v.tail.jx;
            const bool low = rvs;
            if (rest == 1 && (low ? bp.lb > bp.ub : bp.rb < bp.db)) {          // one residue short: one block further
                if (low) { --bp.lb; bp.ub = std::max(bp.lb > (uint32_t) P->ext_block ? bp.lb - (uint32_t) P->ext_block : 0u, bp.zl); }
                else { ++bp.rb; bp.db = std::min(bp.rb + (uint32_t) P->ext_block, bp.zr); }
            } else pull(low, rvs ? 3 : 1, rest < P->min_agap);
        }
    }
    // ---- the units' coordinates after the region was re-cut around the moved pair (Wilip::shift_y)
    static void relocate(std::vector<Unit>& wl, int by, int new_len)
    {
        size_t furthest = 0;
        for (size_t k = 0; k < wl.size(); ++k) {
            Unit& w = wl[k];
            if (by) {
                for (Hsp& h : w.hsp) h.jy += by;
                if (w.llmt) w.llmt += by;
                w.ulmt += by;
            }
            if (w.ulmt > wl[furthest].ulmt) furthest = k;
        }
        wl[furthest].ulmt = new_len;
        wl[furthest].hsp[wl[furthest].num].jy = new_len;
    }
    // ---- the units that hold become loci
    void admit(std::vector<Unit>& wl, const Region& reg, int cut, int pair_index)
    {
        std::stable_sort(wl.begin(), wl.end(), [](const Unit& x, const Unit& y) { return x.scr != y.scr ? x.scr > y.scr : x.nid > y.nid; });
        bp.jscr = wl[0].scr;
        const int room = P->max_out2;
        for (const Unit& w : wl) {
            if (!w.num || w.scr < cut) break;
            if (pair_index >= P->max_out && w.scr < critjscr) break;
            Locus cg;
            cg.at = reg; cg.left = w.llmt; cg.right = std::min(w.ulmt, reg.len); cg.jscr = w.scr;
            const Hsp& f = w.hsp[0]; const Hsp& l = w.hsp[w.num - 1];
            int lo = cg.site(f.jy), hi = cg.site(l.jy + l.jlen);
            if (reg.rvs) std::swap(lo, hi);
            bool taken = false;                         // a locus already taken on this stretch of the strand?
            for (size_t k = loci.size(); k-- > 0 && !taken; ) {
                const Locus& o = loci[k];
                if (o.at.chr != reg.chr || o.at.rvs != reg.rvs) continue;
                const Hsp& of = o.hsp[0]; const Hsp& ol = o.hsp[o.hsp.size() - 2];
                int olo = o.site(of.jy), ohi = o.site(ol.jy + ol.jlen);
                if (reg.rvs) std::swap(olo, ohi);
                taken = hi > olo && lo < ohi;
            }
            if (taken) continue;
            // behind everything that scores at least as much
            size_t at = loci.size();
            while (at > 0 && loci[at - 1].jscr < cg.jscr) --at;
            if ((int) at >= room) break;
            loci.insert(loci.begin() + (long) at, cg);
            while (loci.size() > 1 && loci.back().jscr < cut) loci.pop_back();         // the tail below the cut-off goes (never the best)
            if ((int) loci.size() >= P->max_out) critjscr = std::max(0, loci[P->max_out - 1].jscr - P->vthr);
            loci[at].hsp.assign(w.hsp.begin(), w.hsp.begin() + w.num + 1);
            if ((int) loci.size() > room) loci.pop_back();
        }
    }

    // ---- the machine.  needs(): the region whose units it must be given next (false: the call is over, see `result`).
    // take(units): the answer to that region; units may be empty.
    bool needs(int& pair_index, Region& r)
    {
        while (!done) {
            if (in_pair) { pair_index = (int) at_pair; r = want; return true; }
            for ( ; at_pair < pairs.size() && n_fail && !eligible(pairs[at_pair]); ++at_pair) {}
            if (at_pair >= pairs.size() || !n_fail) { finish(); break; }
            bp = pairs[at_pair]; bp.jscr = 0;
            seen = bp; tries = 0; left_seen = q.right; right_seen = q.left;
            if (!region_of(bp, *G, P->blklen, want)) { close_pair(2); continue; }
            in_pair = true;
        }
        return false;
    }
    void take(std::vector<Unit>& wl)
    {
        if (wl.empty()) { close_pair(2); return; }
        const Verdict v = weigh(wl);
        if (!v.n_hold) { close_pair(v.n_chains ? 2 : 0); return; }
        move_ends(v);
        const bool moved = bp.lb != seen.lb || bp.rb != seen.rb;
        if (P->dvsp == 1 && tries++ < P->no_retry && moved) {         // a protein query looks again at the region as moved
            seen = bp;
            if (!region_of(bp, *G, P->blklen, want)) { close_pair(2); return; }
            return;
        }
        Region reg = want;
        int below = (int) seen.lb - (int) bp.lb, above = (int) bp.rb - (int) seen.rb;
        if (below || above) {
            if (!region_of(bp, *G, P->blklen, reg)) { close_pair(0); return; }
            int ahead = bp.rvs ? above : below;         // blocks added in front of the region as the search reads it
            if (ahead) {
                const int short_last = (bp.rvs && bp.rb == bp.zr) ? P->blklen - reg.len % P->blklen : 0;       // (the chromosome's last block)
                ahead = ahead * P->blklen - short_last;
            }
            relocate(wl, ahead, reg.len);
        }
        admit(wl, reg, v.cut, (int) at_pair);
        pairs[at_pair] = bp;
        close_pair(1);
    }
    void close_pair(int outcome)
    {
        if (in_pair || outcome == 2) { if (at_pair < pairs.size()) pairs[at_pair].jscr = bp.jscr; }
        if (outcome == 1) ++n_held; else if (outcome == 2) --n_fail;
        in_pair = false; ++at_pair;
    }
    void finish()
    {
        done = true;
        result = n_held ? (int) loci.size() : (forced ? -1 : 0);
    }
};

}   // namespace spdp_loci
#endif
