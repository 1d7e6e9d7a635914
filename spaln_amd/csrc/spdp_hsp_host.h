// spdp_hsp_host.h -- the word-seeded HSP search in its host form: one (query range, genomic range) at a time, on the calling
// thread.  The seeded walks ask for the HSPs of the gap they stand in (recursion levels 0 .. 2: thousands of small searches,
// each needed at once by the fiber that asks); the block search's device form (spdp_hsp.hip) hands over the few tasks it cannot
// hold.  Same decisions as the reference's Wilip (ogotoh/spaln v3.0.7 src/wln.cc: Wlp::Wlp :210-232, foldseq / lookup :253-320,
// dmsnno / dmsnno31 / scan_b :554-678, enter / storedh :471-552, eval / reeval :358-469), reached the way the device form
// reaches them: the words both sides share are listed as (diagonal, query position), sorted, and every diagonal's words are then
// scored in one go -- no rolling window of diagonal states, no per-word chains.  The HSPs go through spdp_hsp_chain.h.
#ifndef SPDP_HSP_HOST_H_
#define SPDP_HSP_HOST_H_

#include "spdp_hsp_chain.h"

namespace spdp_hsp {

struct Seqs {                                           // the two sequences as the reference's Seq objects present them to Wilip
    const uint8_t* a; int a_len, a_left, a_right, a_exgl, a_exgr;
    const uint8_t* b; int b_len, b_left, b_right;
    int bbt;                                            // 1: nucleotide query, 3: protein query against tron codes
    const int16_t* sigS; const int16_t* sigE; const int16_t* sigT;     // protein only: the Exinon's start / coding / stop signals by position, or null
};

struct Search {
    const SpdpWilipModel* M; const Seqs* P; int level;
    SpdpWilipLevel L;                                   // the level's parameters as this search uses them
    int width, weight, bbt, mm, precutoff;
    std::vector<int> exam;                              // where the pattern's residues sit
    bool usable = false, count_only = false;            // count_only: distant species at the finer levels -- a diagonal counts its words

    Search(const SpdpWilipModel* m, const Seqs* p, int lv) : M(m), P(p), level(lv)
    {
        L = m->level[std::max(lv, 0)];
        bbt = p->bbt; mm = p->a_right - p->a_left;
        width = L.bitpat_len > 0 ? L.bitpat_len : L.width;
        for (int w = 0; w < width; ++w) if (L.bitpat_len <= 0 || L.bitpat[w]) exam.push_back(w);
        weight = (int) exam.size();
        precutoff = L.cutoff - L.gain * L.tpl;
        count_only = m->crs && lv > 1;
        if (mm <= L.width - 1) return;
        if (lv < 0 && p->a_len < m->shortquery) {       // a short query against a block pair: everything scaled down
            L.cutoff = L.cutoff * p->a_len / m->shortquery;
            L.vthr = L.vthr * p->a_len / m->shortquery;
            precutoff = precutoff * p->a_len / m->shortquery;
        }
        usable = mm - (L.width - 1) > 0 && mm >= L.width && p->b_right - p->b_left >= bbt * L.width;
    }
    int cls(int code) const { return L.convtab[code & 31]; }
    // the word whose first residue is seq[at], residues `step` apart; -1: a residue outside the alphabet
    int64_t word(const uint8_t* seq, int at, int step) const
    {
        int64_t w = 0;
        for (int k = 0; k < weight; ++k) {
            const int c = cls(seq[at + step * exam[k]]);
            if (c >= L.elem) return -1;
            w = w * L.elem + c;
        }
        return w < L.mask ? w : -1;
    }

    struct Seed { int first, diag, last, score; };      // a seed segment: words first .. last of the query on one diagonal
    // ---- the words both sides share, by diagonal; every diagonal scored
    void seeds(std::vector<Seed>& out) const
    {
        const int nk = mm - (L.width - 1);
        const int span_b = bbt == 3 ? 3 * L.width - 1 : L.width - 1;
        const int nn = P->b_right - P->b_left - span_b;
        std::vector<std::pair<int64_t, int>> qw;        // (word, query position)
        for (int m = 0; m < nk; ++m) { const int64_t w = word(P->a, P->a_left + m, 1); if (w >= 0) qw.emplace_back(w, m); }
        std::sort(qw.begin(), qw.end());
        std::vector<uint64_t> shared;                   // (diagonal + bias) << 32 | query position
        const int64_t bias = (int64_t) bbt * nk;
        for (int n = 0; n < nn; ++n) {
            const int64_t w = bbt == 3 ? word(P->b, P->b_left + n + 1, 3) : word(P->b, P->b_left + n, 1);
            if (w < 0) continue;
            for (auto it = std::lower_bound(qw.begin(), qw.end(), std::make_pair(w, -1)); it != qw.end() && it->first == w; ++it)
                shared.push_back((uint64_t) (n - (int64_t) bbt * it->second + bias) << 32 | (uint32_t) it->second);
        }
        std::sort(shared.begin(), shared.end());
        const int tplwt = L.tpl * L.gain;
        for (size_t i = 0; i < shared.size(); ) {
            const uint64_t dkey = shared[i] >> 32;
            const int r = (int) ((int64_t) dkey - bias);
            if (count_only) {                           // the diagonal's words counted; enough of them: the stretch they span is scored
                const int first = (int) (uint32_t) shared[i];
                int n_words = 0, last = first;
                for ( ; i < shared.size() && (shared[i] >> 32) == dkey; ++i) { ++n_words; last = (int) (uint32_t) shared[i]; }
                if (n_words >= M->min_hit) spanned(r, first, last, out);
                continue;
            }
            int score = 0, best = 0, prev = -(L.width + 1), first = 0, best_at = 0;
            for ( ; i < shared.size() && (shared[i] >> 32) == dkey; ++i) {
                const int m = (int) (uint32_t) shared[i];
                const int gap = m - prev - L.width;
                if (gap > 0) {
                    const int floor_ = best - L.cutoff;
                    score -= L.gain * gap;
                    if (floor_ > score || score < 0) {
                        if (floor_ > 0) out.push_back({first, r, best_at, best});
                        score = tplwt + (m < L.width ? L.gain * (L.width - m) : 0);
                        best = score; best_at = first = m;
                    } else score += tplwt;
                } else score += (m - first == 1) ? L.gain1 : L.gain;
                if (score > best) { best = score; best_at = m; }
                prev = m;
            }
            if (best > precutoff) {
                const int over = best_at + 2 * L.width - mm;
                if (over > 0) best += L.gain * over;
                if (best > L.cutoff) out.push_back({first, r, best_at, best});
            }
        }
    }
    // distant species (Wlp::storedh): the best-scoring window of the diagonal around the words first .. last
    void spanned(int r, int first, int last, std::vector<Seed>& out) const
    {
        int lo = first - L.width;
        const int hi = last + 2 * L.width;
        int x = r < 0 ? (bbt - r - 1) / bbt : lo;
        if (x < 0) x = 0;
        int as = P->a_left + x, bs = P->b_left + r + bbt * x + (bbt == 3 ? 1 : 0);
        const int at = std::min(P->a_right, P->a_left + hi), bt = P->b_right;
        int run = 0, best = 0, from = x, w_from = x, w_to = x;
        for (int m = x; as < at && bs < bt; ++as, bs += bbt, ++m) {
            run += M->mtx[P->a[as] * M->mtx_cols + P->b[bs]];
            if (run <= 0) { run = 0; from = m; }
            else if (run > best) { best = run; w_to = m; w_from = from; }
        }
        // (handed on in the seed's form: stretch_and_trim reads first / last back as jx and jlen - width)
        if (best > L.vthr) out.push_back({w_from + 1, r, w_to + 1 - L.width, 0});
    }
    int sig(const int16_t* arr, int i) const { return (i < 0 || i > P->b_len + 2) ? 0 : arr[i]; }
    // ---- stretch and trim (Wlp::eval after Wlp::reeval's shift to sequence coordinates); false: at or below the threshold
    bool stretch_and_trim(const Seed& s, Hsp& h) const
    {
        const bool prot = bbt == 3, exinon = prot && P->sigS;
        int jx = P->a_left + s.first, jy = P->b_left + bbt * s.first + s.diag, jlen = s.last - s.first + L.width;
        int as = jx, bs = jy + (prot ? 1 : 0), bb = jy + 1, scr = 0;
        if (exinon && jx == 0 && P->a[as] == M->met && sig(P->sigS, bb) > 0) scr = L.vthr / 2;
        if (scr <= 0 && P->a_exgl) { const int lend = L.tpl - jx; if (lend > 0) scr += M->end_bonus * std::min(lend, L.tpl); }
        while (--as >= 0 && (bs -= bbt) >= 0) {         // backwards while the classes agree
            if (cls(P->a[as]) != cls(P->b[bs])) break;
            --jx; jy -= bbt; ++jlen; bb -= bbt;
        }
        if (as < 0) bs -= bbt;
        const int a_end = std::min(jx + jlen, P->a_right), b_end = std::min(jy + bbt * jlen, P->b_right);
        const int b_stop = prot ? P->b_len - 1 : P->b_len, from = as;
        int len = 0, nid = 0, best = scr, w_from = 0, w_len = 0, w_nid = 0, restart = 0;
        while (++as < P->a_len && (bs += bbt) < b_stop) {       // forwards: through the seed, then while the classes agree
            const int ca = P->a[as], cb = P->b[bs];
            if ((as >= a_end || bs >= b_end) && cls(ca) != cls(cb)) break;
            ++len;
            scr += M->mtx[ca * M->mtx_cols + cb];
            if (ca == cb || (ca == M->ser && cb == M->ser2)) ++nid;
            if (exinon) { scr += sig(P->sigE, bb); bb += bbt; }
            if (scr < 0) { scr = 0; restart = as - from; len = nid = 0; }
            if (scr > best) { best = scr; w_from = restart; w_len = len; w_nid = nid; }
        }
        jx += w_from; jy += bbt * w_from;
        if (M->crs == 0 && prot) scr -= std::min(w_len - w_nid, 3) * L.vthr;
        if (as == P->a_len && exinon && sig(P->sigT, bb) > 0) scr += L.vthr / 2;
        else {
            const int rend = L.tpl - P->a_right + jx + w_len;
            if (P->a_exgr && rend > 0) scr += M->end_bonus * std::min(rend, L.tpl);
            else if (w_nid == w_len) scr += M->end_bonus * 4;
        }
        h = Hsp{jx, jy, w_len, w_nid, scr};
        return scr > L.vthr;
    }
};

struct GapCosts { const int16_t* intpen; int intpen_len, gop, gep, lgop, lgep, codonk1; };

// Wilip::Wilip(seqs, pwd, level): the units, best first; empty: none
inline void search(const SpdpWilipModel* m, const Seqs& p, const GapCosts& g, int level, std::vector<Unit>& units)
{
    units.clear();
    Search S(m, &p, level);
    if (!S.usable) return;
    std::vector<Search::Seed> seeds;
    S.seeds(seeds);
    std::vector<Hsp> hsps;
    for (const Search::Seed& s : seeds) { Hsp h; if (S.stretch_and_trim(s, h)) hsps.push_back(h); }
    const ChainCost C = {m, g.intpen, g.intpen_len, g.gop, g.gep, g.lgop, g.lgep, g.codonk1, p.bbt, S.L.vthr};
    chain(hsps, C, p.a_left, p.a_right, p.b_left, p.b_right, units);
}

}   // namespace spdp_hsp
#endif
