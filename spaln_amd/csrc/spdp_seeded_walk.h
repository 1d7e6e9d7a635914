// spdp_seeded_walk.h -- the seeded path of alignS_ng: one query's walk over its HSPs, host side.
//
// What it mirrors (ogotoh/spaln v3.0.7, src/fwd2s1.cc):
//   Aln2s1::globalS_ng (algmode.qck != 0)      :2674-2694      SeedWalk::run
//   Aln2s1::seededS_ng                         :2587-2672      SeedWalk::seeded
//   Aln2s1::bestwlu                            :2541-2585      SeedWalk::best_unit
//   Aln2s1::interpolateS                       :2405-2539      SeedWalk::interpolate
//   Aln2s1::indelfreespjS                      :2003-2062      SeedWalk::indel_free_junction
//   Aln2s1::backforth                          :1966-1993      SeedWalk::back_and_forth
//   Aln2s1::creepback / creepfwrd              :2064-2092      SeedWalk::creep_back / creep_fwrd
//   Aln2s1::nearest5ss / nearest3ss            :2096-2162      SeedWalk::nearest_site<5 / 3>
//   Aln2s1::micro_exon                         :2164-2236      SeedWalk::micro_exon
//   Aln2s1::first_exon(_wmm) / last_exon(_wmm) :2238-2403      SeedWalk::end_exon<FIRST / LAST>
//   Aln2s1::shortcutS_ng                       :1899-1930      SeedWalk::shortcut
//   Aln2s1::openendS_ng + the two X-drop end extensions back2ward5endS_ng / for2ward3endS_ng
//                                              :1932-1964, 1384-1627   SeedWalk::open_end, end_extension
//   Aln2s1::diagonalS_ng                       :1629-1665      SeedWalk::diagonal
//   BoyerMoore (nucleotide text and pattern)   src/boyer_moore.cc     ExactFinder
//
// The reference walks the HSPs of one query on one CPU thread and calls its DP engines (lspS_ng, trcbkalignS_ng)
// synchronously wherever no closed-form rule joins two HSPs.  Here the walk is the same sequence of decisions on the
// same mutable state (active ranges, end flags, the record file, the splice-phase marks), but every DP call goes
// through DpBackend: the product's backend (spdp_seeded.cpp) parks the request until the walks of all queries in
// flight have reached a DP call, runs one device batch and resumes them.  Header only: the same source is compiled
// into the product and into the CPU checker the tests use (oracle/walk_check.cpp, callbacks into the oracle).
#ifndef SPDP_SEEDED_WALK_H_
#define SPDP_SEEDED_WALK_H_

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../../include/spdp.h"

namespace spdp_seed {

struct Span { int al, ar, bl, br; uint8_t a_exgl, a_exgr, b_exgl, b_exgr; };     // Seq::left / right + inex.exgl / exgr of both
struct Bound { int la, lb, ua, ub; };                                             // BOUND, src/aln.h:95
struct Hsp { int jx, jy, jlen, nid, jscr; };                                      // JUXT, src/seq.h:174
struct Unit { int num, nid, tlen, llmt, ulmt, scr; std::vector<Hsp> jxt; };      // WLUNIT, src/wln.h:59 (jxt: num + 1)

struct DpBackend {
    virtual ~DpBackend() {}
    // Aln2s1::lspS_ng(wdw) on the span: appends the records it writes, returns its score
    virtual int lsp(const Span& s, const SpdpWindow& w, std::vector<SpdpSkl>& rec) = 0;
    // Aln2s1::trcbkalignS_ng(wdw, spj, mc): cut = {left, right} of the genomic range the sweep jumps over, or null
    virtual int trcbk(const Span& s, const SpdpWindow& w, const int* cut, std::vector<SpdpSkl>& rec) = 0;
    // Wilip(seqs, pwd, level) on the span (src/wln.cc:980): the units in the order Wilip holds them
    virtual bool wilip(int level, const Span& s, std::vector<Unit>& units) = 0;
};

// Exact occurrences of a nucleotide pattern in a text, in the order and with the skips of the reference's search
// (BoyerMoore(b, a, step), src/boyer_moore.cc:36-118, 188-230): two codes match when their base sets intersect
// (code - 1 is a 4-bit set, 0 matches nothing), the shift tables are built with the same relation, a hit is followed
// by the table's own shift.  The occurrence list decides which terminal exon first_exon / last_exon pick, so the
// scan is kept as it is -- including occurrences the relaxed matching makes it skip.
class ExactFinder {
    const uint8_t* text; int tlen; int origin;
    std::vector<uint8_t> pat; int plen;
    std::vector<int> by_code, by_suffix;
    int dir, after_hit, at;
    static bool hit(uint8_t x, uint8_t y) { return x && y && ((x - 1) & (y - 1)); }
public:
    ExactFinder(const uint8_t* b, int bl, int br, const uint8_t* a, int al, int ar, int n_codes, int direction)
        : text(b + bl), tlen(br - bl), origin(bl), pat(a + al, a + ar), plen(ar - al), dir(direction)
    {
        pat.push_back(0);
        if (dir < 0) std::reverse(pat.begin(), pat.begin() + plen);
        by_code.assign(std::max(n_codes, 256), plen);
        for (int j = 0, skip = plen; j < plen; ++j) by_code[pat[j]] = --skip;
        by_suffix.resize(std::max(plen, 1));
        std::vector<int> link(std::max(plen, 1));
        for (int j = 0, v = 2 * plen; j < plen; ++j) by_suffix[j] = --v;
        int j = plen;
        for (int k = plen; --k >= 0; ) {
            link[k] = j;
            pat[plen] = pat[k];                                 // sentinel: the chain below always ends
            while (!hit(pat[j], pat[k])) {
                by_suffix[j] = std::min(by_suffix[j], plen - 1 - k);
                j = link[j];
            }
            --j;
        }
        after_hit = std::max(j + 1, 2) * dir;
        for (int s = j, v = plen, q = 0; q < plen; ++q) {
            by_suffix[q] = std::min(by_suffix[q], s + v--);
            if (q >= s) s = s >= 0 ? link[s] : 0;               // (a pattern that is one repeated base leaves s = -1: the reference
        }                                                       //  reads the word in front of its array there, 0 with glibc)
        if (dir < 0) {
            std::reverse(pat.begin(), pat.begin() + plen);
            std::reverse(by_suffix.begin(), by_suffix.begin() + plen);
        }
        at = dir > 0 ? 0 : tlen;
    }
    bool finished() const { return dir > 0 ? at >= tlen : at <= 0; }
    // next occurrence (position in b of its first base), or -1 when the scan has run out
    int next()
    {
        if (dir > 0) {
            int i = at + plen - 1;
            at = tlen;
            while (i < tlen) {
                int j = plen - 1;
                while (j >= 0 && hit(text[i], pat[j])) { --i; --j; }
                if (j < 0) { at = i + after_hit; return i + 1 + origin; }
                i += std::max(by_code[text[i]], by_suffix[j]);
            }
        } else {
            int i = at - (plen - 1);
            at = 0;
            while (i >= 0) {
                int j = 0;
                while (j < plen && hit(text[i], pat[j])) { ++i; ++j; }
                if (j >= plen) { at = i + after_hit; return i - plen + origin; }
                i -= std::max(by_code[text[i]], by_suffix[j]);
            }
        }
        return -1;
    }
};

// the phase marks of a window as one walk sees them: the caller's array plus the few marks the walk itself has set (a
// private copy of the array per walk cost more than the rest of a walk's set-up)
struct PhaseMarks {
    const int8_t* base = nullptr;
    std::vector<int8_t> own;                    // when the caller gave none: derived from the site flags
    std::vector<std::pair<int, int8_t>> edits;
    int8_t operator[](int n) const
    {
        for (size_t i = edits.size(); i-- > 0; ) if (edits[i].first == n) return edits[i].second;
        return base[n];
    }
    void set(int n, int8_t v)
    {
        for (auto& e : edits) if (e.first == n) { e.second = v; return; }
        edits.push_back({n, v});
    }
    void bind(const int8_t* b) { base = b; own.clear(); edits.clear(); }
    void derive(int N) { own.assign(N, -2); base = own.data(); edits.clear(); }
};

class SeedWalk {
public:
    // inputs (borrowed)
    const uint8_t* a = nullptr; int a_len = 0;
    const uint8_t* b = nullptr; int b_len = 0;
    const int16_t* sig5 = nullptr; const int16_t* sig3 = nullptr;
    const uint8_t* cano5 = nullptr; const uint8_t* cano3 = nullptr; const uint8_t* dinc = nullptr;
    const int32_t* cip = nullptr;               // Cip_score::cip_score(m), m = 0 .. a_len, or null (SpdpProblem::cip)
    PhaseMarks phs5, phs3;                      // SGPT2::phs5 / phs3: the walk marks the junctions it accepts (:2055-2059)
    uint8_t f5[16] = {0}, f3[16] = {0};         // INT53::cano5 / cano3 as levels 0 .. 3 by dinucleotide class (cano5 / cano3 above only say "a site")
    int lvl5(int n) const { return cano5[n] ? (f5[dinc[n] >> 4] ? f5[dinc[n] >> 4] : cano5[n]) : 0; }
    int lvl3(int n) const { return cano3[n] ? (f3[dinc[n] & 15] ? f3[dinc[n] & 15] : cano3[n]) : 0; }
    const SpdpScoring* sc = nullptr;
    const SpdpSeedParams* sp = nullptr;
    DpBackend* dp = nullptr;
    int lowest_level = 0;                       // b->wllvl
    std::vector<Hsp> top_hsps;                  // b->jxt: CdsNo HSPs + one slot (empty: none)
    bool a_sens = false;                        // a->inex.sens (A_RevCom in the header record)

    // state
    Span cur{};
    std::vector<SpdpSkl> rec;                   // the Mfile, dummy record first
    bool is3end = false;
    bool unsupported = false;                   // the walk met a state this restatement does not serve
    // which joins the walk used (tests assert that the fixtures reach every one): see the J_ names below
    enum { J_ABUT, J_DIAGONAL, J_HEAD_CONT, J_HEAD_SHORT, J_HEAD_NOGENOME, J_HEAD_EXTEND, J_HEAD_EXON, J_TAIL_SHORT,
           J_TAIL_NOGENOME, J_TAIL_EXTEND, J_TAIL_EXON, J_JUNCTION, J_MICRO_EXON, J_SHORTCUT, J_BACKFORTH, J_SMALL_DP,
           J_RECURSE, J_DP, J_GIVEUP_LOCALC, J_GIVEUP_HEAD, J_GIVEUP_TAIL, J_GIVEUP_INNER, J_PICK_UNIT, J_COUNT };
    int joins[J_COUNT] = {0};

    int NEV() const { return SPDP_NEVSEL; }
    bool Local() const { return (sp->lcl & 16) != 0; }
    bool LocalC() const { return Local() && (sp->lcl & 32); }
    int sim(int i, int j) const { return sc->mtx[a[i] * sc->mtx_dim + b[j]]; }
    int gap_penalty(int i) const
    {
        if (i == 0) return 0;
        return i > sp->codonk1 ? sc->lgop + i * sc->lgep : sc->gop + i * sc->gep;
    }
    int int_pen(int len) const                  // IntronPenalty::Penalty(n), materialised by the caller
    {
        if (len < 0) return SHRT_MIN;
        if (len >= sc->intpen_len) len = sc->intpen_len - 1;
        return sc->intpen[len];
    }
    // Exinon::sig53(m, n, IE5P3), src/codepot.cc:416-421
    int sig53_5p3(int m, int n) const { return sig5[m] + sig3[n] + sc->t53[16 * (dinc[m] >> 4) + (dinc[n] & 15)]; }
    int is_canon(int d, int ac) const           // Exinon::isCanon, src/codepot.h:108-113
    {
        const int c5 = lvl5(d), c3 = lvl3(ac);
        return ((c5 == 3 && c3 == 3) || (c5 == 2 && c3 == 2) || (c5 == 1 && c3) || (c5 && c3 == 1)) ? c5 + c3 : 0;
    }
    void put(int m, int n) { rec.push_back({m, n}); }
    int end_margin() const { return (int) ((sp->vthr + sc->gop) / sc->gep); }
    int slmt() const { return sp->vthr / 2; }

    // stripe(seqs, &wdw, sh, cmode), src/aln2.cc:156-176
    SpdpWindow stripe(int sh, int cmode = 0) const
    {
        SpdpWindow w;
        if (sh < 0) sh = -sh * std::min(cur.ar - cur.al, cur.br - cur.bl) / 100;
        w.up = cur.br - cur.ar;
        w.lw = cur.bl - cur.al;
        if (cmode == 1) w.lw = w.up;
        else if (cmode == 2) w.up = w.lw;
        else if (w.up < w.lw) std::swap(w.up, w.lw);
        w.up += sh; w.lw -= sh;
        w.up = std::min(w.up, cur.br - cur.al);
        w.lw = std::max(w.lw, cur.bl - cur.ar);
        w.width = w.up - w.lw + 3;
        return w;
    }

    // ---- closed-form joins ------------------------------------------------------------------------------------
    int diagonal()
    {
        const bool LL = Local() && cur.a_exgl && cur.b_exgl, LR = Local() && cur.a_exgr && cur.b_exgr;
        const int dlt = Local() ? 0 : (cur.br - cur.bl) - (cur.ar - cur.al);
        const bool sw = dlt < 0;                        // the reference swaps the two sequences for the walk
        const uint8_t* x = sw ? b : a; const uint8_t* y = sw ? a : b;
        const int xl = sw ? cur.bl : cur.al, xr = sw ? cur.br : cur.ar, yl = sw ? cur.al : cur.bl;
        int mL = xl, mR = xr, scr = 0, best = NEV();
        for (int m = xl; m < xr; ++m) {
            const int p = x[m], q = y[yl + (m - xl)];
            scr += sw ? sc->mtx[q * sc->mtx_dim + p] : sc->mtx[p * sc->mtx_dim + q];
            if (LL && scr < 0) { scr = 0; mL = m + 1; }
            if (LR && scr > best) { best = scr; mR = m + 1; }
        }
        int r = yl - xl;
        if (sw) r -= dlt;
        put(mL, mL + r);
        put(mR, mR + r);
        return LR ? best : scr;
    }

    int creep_back(int ovr, int bscr, const Bound& lub)
    {
        int d = 0;
        while (cur.al > lub.la && cur.bl > lub.lb && (ovr < 0 || std::abs(d) < bscr)) {
            d += sim(cur.al - 1, cur.bl - 1);
            --cur.al; --cur.bl;
            if (++ovr == 0) bscr += d;
        }
        return d;
    }
    int creep_fwrd(int& ovr, int bscr, const Bound& lub)
    {
        int d = 0;
        while (cur.ar < lub.ua && cur.br < lub.ub && (ovr < 0 || std::abs(d) < bscr)) {
            d += sim(cur.ar, cur.br);
            ++cur.ar; ++cur.br;
            if (++ovr == 0) bscr += d;
        }
        return d;
    }

    // two HSPs that overlap on both sequences without room for an intron: where along the overlap to switch
    int back_and_forth(int ovr, const Bound& lub)
    {
        std::vector<int> acc(ovr + 1, 0);
        int scr = 0, i = ovr, m = cur.al, n = cur.bl;
        for (;;) {                                      // donor-side diagonal, backwards from the left ends
            if (--i < 0) break;
            if (--m < lub.la) break;
            if (--n < lub.lb) break;
            acc[i] = scr += sim(m, n);
        }
        int best = scr;
        scr = 0;
        int where = ++i;
        m = cur.ar + i; n = cur.br + i;
        int pa = m, pb = n;
        for (;;) {                                      // acceptor-side diagonal, forwards from the right ends
            if (!(i++ < ovr)) break;
            if (!(m++ < lub.ua)) break;
            if (!(n++ < lub.ub)) break;
            scr += sim(pa++, pb++);
            if ((acc[i] += scr) > best) { best = acc[i]; where = i; }
        }
        SpdpSkl k = {cur.ar + where, cur.br + where};
        rec.push_back(k);
        int dr = (cur.br - cur.ar) - (cur.bl - cur.al);
        if (dr >= 0) k.n -= dr; else k.m -= (dr = -dr);
        rec.push_back(k);
        return best + gap_penalty(dr);
    }

    // two HSPs that abut or overlap on the query and lie an intron apart on the genome: the junction inside the
    // overlap that pays best, the doubly counted match score taken back
    bool indel_free_junction(int agap, int& iscr, bool write)
    {
        const int ilen = cur.br - cur.bl - agap;
        if (ilen < sp->minl) { iscr = gap_penalty(ilen); return true; }
        agap = 1 - agap;
        const int reach = std::min(std::min(cur.al, cur.bl), agap + 16);
        std::vector<int> bw(reach + 2, 0);
        int i = 0, v = 0;
        int pb = cur.bl, pd = cur.bl + ilen, pa = cur.al;   // the three read positions, moved as the reference moves its pointers
        while (i < reach) {
            --pb; --pd;
            if (!(b[pb] == b[pd] || i < agap)) break;
            --pa;
            bw[++i] = v += sim(pa, pb);
        }
        std::reverse(bw.begin(), bw.begin() + i + 1);
        SpdpSkl k = {cur.al - i, cur.bl - i};
        iscr = NEV();
        const int ntry = sp->crs ? 1 : 2;
        for (int retry = 0; retry < ntry && iscr == NEV(); ++retry) {
            int m = k.m, n = k.n;
            int qa = m, qb = pd;
            int t = 0;
            for (v = 0; n <= cur.bl; ++n, ++t, ++m) {
                const int rc = is_canon(n, n + ilen);
                if (retry || rc) {
                    int x = sig53_5p3(n, n + ilen);
                    // use_spb(): an annotated intron position of the query earns its bonus at a canonical junction
                    // (PfqItr::match_score = Cip_score::cip_score of that position, src/gsinfo.h:226-229, gsinfo.cc:65-79)
                    if (cip && m >= 0 && m <= a_len && cip[m] && rc > 3) x += cip[m];
                    const int y = x - v - bw[t];
                    if (y > iscr) { k.n = n; k.m = m; iscr = y; }
                }
                v += sim(qa++, ++qb);
            }
        }
        if (iscr <= NEV()) return false;
        if (write) {
            rec.push_back(k);
            phs5.set(k.n, 0);
            k.n += ilen;
            rec.push_back(k);
            phs3.set(k.n, 0);
            iscr += int_pen(ilen);
        }
        return true;
    }

    // the splice site nearest to the open end of the genomic span: SIDE 5 looks for a donor around b.left,
    // SIDE 3 for an acceptor around b.right
    template <int SIDE>
    int nearest_site(const Bound& bab) const
    {
        const int from = SIDE == 5 ? cur.bl : cur.br;
        const int a0 = SIDE == 5 ? cur.al : cur.ar;
        auto strong = [&](int n, bool retry) {
            return SIDE == 5 ? (sig5[n] > sp->gc_sig5 || (retry && phs5[n] == 0))
                             : (sig3[n] > 0 || (retry && phs3[n] == 0));
        };
        auto sig = [&](int n) { return SIDE == 5 ? sig5[n] : sig3[n]; };
        auto phs = [&](int n) { return SIDE == 5 ? phs5[n] : phs3[n]; };
        for (int retry = 0; ; ) {
            int nu = from, qa = a0, qb = nu;
            const int stop_up = std::max(bab.la, cur.al - 9);
            bool eij = false;
            for ( ; qa > stop_up && nu > bab.lb; --nu) {
                eij = strong(nu, retry != 0);
                if (eij) break;
                if (!sp->crs) { --qa; --qb; if (a[qa] != b[qb]) break; }
            }
            if (nu == from && eij) return nu;
            int nd = from, sd = from;                   // sd: the position whose signals the downward scan looked at last
            qa = a0; qb = nd;
            const int stop_dn = std::min(cur.al + 9, bab.ua);
            eij = false;
            while (qa < stop_dn) {
                if (!(++nd < bab.ub)) break;
                ++sd;
                eij = strong(sd, retry != 0);
                if (eij) break;
                if (!sp->crs) { const bool same = a[qa] == b[qb]; ++qa; ++qb; if (!same) break; }
            }
            if (retry++ == 0 && sig(nu) <= 0 && sig(sd) <= 0) continue;
            if (phs(nu) && phs(sd)) return -1;
            if (phs(nu)) return nd;
            if (phs(sd)) return nu;
            if (from - nu == nd - from) return sig(nu) > sig(sd) ? nu : nd;
            return (from - nu < nd - from) ? nu : nd;
        }
    }

    int micro_exon(const Bound& bab)
    {
        const int l = nearest_site<5>(bab);
        if (l < 0) return NEV();
        const int r = nearest_site<3>(bab);
        if (r < 0) return NEV();
        const int d5 = l - cur.bl, d3 = r - cur.br;
        cur.bl = l; cur.al += d5; cur.br = r; cur.ar += d3;
        const int alen = cur.ar - cur.al, blen = r - l;
        int best = int_pen(blen);
        int f = -1;
        if (alen <= 0) {
            SpdpSkl k = {cur.al, cur.bl};
            if (alen < 0) { k.m = cur.al + alen; rec.push_back(k); k.m = cur.al; }
            rec.push_back(k);
            k.n = cur.br;
            rec.push_back(k);
            return best + sig53_5p3(l, r);
        }
        const int n9 = cur.br - alen - sp->minl;
        for (int n5 = cur.bl + sp->minl; n5 < n9; ++n5) {
            if (phs3[n5] || phs5[n5 + alen]) continue;
            const int n3 = n5 + alen;
            int ms = 0;
            for (int t = 0; t < alen; ++t) ms += sim(cur.al + t, n5 + t);
            const float fs = sp->w2 * ms + sig53_5p3(l, n5) + sig53_5p3(n3, r) + int_pen(n5 - l) + int_pen(r - n3);
            const int scr = (int) fs;
            if (scr > best) { best = scr; f = n5; }
        }
        if (f < 0) {
            cur.bl -= d5; cur.al -= d5; cur.br -= d3; cur.ar -= d3;
            return NEV();
        }
        SpdpSkl k = {cur.al, cur.bl};
        rec.push_back(k);
        if (f != cur.bl) { k.n = f; rec.push_back(k); k.n = f + alen; }
        k.m += alen;
        rec.push_back(k);
        k.n = cur.br;
        rec.push_back(k);
        return best;
    }

    // a terminal exon too short for the HSP search: an exact copy of the query's end somewhere in the genomic span, an
    // intron away from the nearest splice site (FIRST: upstream of the acceptor at b.right; LAST: downstream of the
    // donor at b.left); without an exact copy, the best-scoring ungapped placement at a canonical site
    enum { FIRST = 1, LAST = 2 };
    template <int WHICH>
    int end_exon(const Bound& bab)
    {
        const Span keep = cur;
        auto fail = [&]() { cur.al = keep.al; cur.ar = keep.ar; cur.bl = keep.bl; cur.br = keep.br; return NEV(); };
        const int site = WHICH == FIRST ? nearest_site<3>(bab) : nearest_site<5>(bab);
        if (site < 0) return fail();
        if (WHICH == FIRST) { const int d = site - cur.br; cur.br = site; cur.ar += d; }
        else { const int d = site - cur.bl; cur.bl = site; cur.al += d; }
        if (cur.al >= cur.ar || cur.bl >= cur.br) return fail();
        if (WHICH == FIRST && (cur.ar == 0 || cur.br == 0)) { put(cur.ar, cur.br); return 0; }
        const int alen = cur.ar - cur.al;
        int best = NEV(), pos = -1;
        ExactFinder find(b, cur.bl, cur.br, a, cur.al, cur.ar, sc->mtx_dim, WHICH == FIRST ? -1 : 1);
        while (!find.finished()) {
            const int f = find.next();
            if (f < 0) continue;
            const int don = WHICH == FIRST ? f + cur.ar : site;
            const int acc = WHICH == FIRST ? site : f;
            if (!is_canon(don, acc)) continue;
            const int s = int_pen(acc - don) + sig53_5p3(don, acc);
            if (s > best) { best = s; pos = f; }
        }
        if (pos < 0) {
            // no exact copy: ungapped placements at canonical sites (first_exon_wmm / last_exon_wmm)
            int perfect = 0;
            for (int t = cur.al; t < cur.ar; ++t) perfect += sc->mtx[a[t] * sc->mtx_dim + a[t]];
            perfect = (int) (perfect * sp->w2);
            if (WHICH == FIRST) {
                int n = cur.br - cur.ar - sp->minl;
                for (int nd = n + cur.ar; n >= cur.bl; --n, --nd) {
                    if (!is_canon(nd, site)) continue;
                    int ms = 0;
                    for (int t = cur.al; t < cur.ar; ++t) ms += sim(t, n + (t - cur.al));
                    const int s = (int) ((sig5[nd] + int_pen(site - nd)) + sp->w2 * ms);
                    if (s > best) { pos = n; if (ms == perfect) break; best = s; }
                }
            } else {
                const int rr = cur.br - alen;
                for (int n = cur.bl + sp->minl; n < rr; ++n) {
                    if (!is_canon(site, n)) continue;
                    int ms = 0;
                    for (int t = cur.al; t < cur.ar; ++t) ms += sim(t, n + (t - cur.al));
                    const int s = (int) ((sig3[n] + int_pen(n - site)) + sp->w2 * ms);
                    if (s > best) { pos = n; if (ms == perfect) break; best = s; }
                }
            }
            if (pos < 0) return fail();
        } else {
            for (int t = cur.al; t < cur.ar; ++t) best += sc->mtx[a[t] * sc->mtx_dim + a[t]];
        }
        if (WHICH == FIRST) {
            cur.bl = pos;
            put(cur.al, cur.bl);
            put(cur.ar, cur.bl + cur.ar);
            put(cur.ar, cur.br);
        } else {
            put(cur.al, cur.bl);
            put(cur.al, pos);
            put(cur.ar, pos + alen);
        }
        return best;
    }

    // ---- the intron-less X-drop extensions of an open end (back2ward5endS_ng / for2ward3endS_ng) ------------------
    // Row by row along the query away from the last HSP, columns inside a band that follows the running best cell;
    // a row ends where the score has dropped Vthr below the best end cell seen so far.  Sequential by construction
    // (each row's column range depends on where the previous row peaked and dropped off), a few thousand cells: it
    // runs in the walk.  TOWARDS5 = true is the backward form (5' end), false the forward form (3' end).
    struct Cell { int val, ptr; };
    struct Trail { std::vector<int> m, n, prev;
                   int add(int m_, int n_, int p) { m.push_back(m_); n.push_back(n_); prev.push_back(p); return (int) m.size() - 1; } };
    template <bool TOWARDS5>
    int end_extension(int* last, const SpdpWindow& w, bool lcl, Trail& vmf)
    {
        const int S = TOWARDS5 ? -1 : 1;                // direction of travel along both sequences
        const int NEVv = NEV(), dim = sc->mtx_dim;
        const Cell black = {NEVv, 0};
        const int width = w.width;
        if (width < 3) { unsupported = true; *last = 0; return NEVv; }
        std::vector<Cell> buf(2 * (size_t) width, black);        // H and F by diagonal r = n - m, lw - 1 .. up + 1
        std::vector<uint8_t> dirs(width, 1);
        auto H = [&](int r) -> Cell& { return buf[r - w.lw + 1]; };
        auto F = [&](int r) -> Cell& { return buf[width + r - w.lw + 1]; };
        auto D = [&](int r) -> uint8_t& { return dirs[r - w.lw + 1]; };
        const int m_corner = TOWARDS5 ? cur.ar : cur.al, m_last = TOWARDS5 ? cur.al : cur.ar;
        const int n_corner = TOWARDS5 ? cur.br : cur.bl;
        int best_val = lcl ? 0 : NEVv, best_m = m_corner, best_n = n_corner, best_p = 0;
        vmf.add(0, 0, 0);
        {   // pbinitS_ng / pfinitS_ng: the corner cell and the gap that leaves it along the genome
            int r = n_corner - m_corner;
            H(r).val = 0;
            H(r).ptr = vmf.add(m_corner, n_corner, 0);
            const int rr = TOWARDS5 ? std::min(w.up, cur.br - cur.al) : std::max(w.lw, cur.bl - cur.ar);
            for (int i = 1; TOWARDS5 ? ++r <= rr : --r >= rr; ++i) {
                H(r) = H(r + S);
                if (i == 1) H(r).val += sc->gop;
                H(r).val += sc->gep;
                F(r) = H(r);
            }
        }
        int m = m_corner;
        if (TOWARDS5 ? !cur.a_exgr : !cur.a_exgl) m -= S;       // global end: the corner row is swept as well
        int n1 = m + w.lw, n2 = m + w.up + 1;                   // column limits, carried from row to row
        for (;;) {
            m += S;
            if (TOWARDS5 ? m < cur.al : m > cur.ar) break;
            if (TOWARDS5) { --n1; --n2; }
            lcl = lcl || m == m_last;
            int n = TOWARDS5 ? std::min(n2, cur.br) : std::max(n1, cur.bl);
            const int n_end = TOWARDS5 ? std::max(n1, cur.bl) : std::min(n2, cur.br);
            int r = n - m;
            int nr = n - S;
            bool peak = false;
            Cell e1 = black;
            // the best cell of the current block: a fixed value (an H / F entry that the row does not touch again) or
            // the running horizontal-gap cell itself, which the reference keeps comparing with by address
            bool block_is_e1 = false;
            int block_val = (H(r).val + sp->vthr < best_val) ? NEVv : H(r).val;
            const bool corner_row = m == m_corner;
            const int am = corner_row ? 0 : a[TOWARDS5 ? m : m - 1];
            for (;;) {
                n += S;
                if (TOWARDS5 ? n < n_end : n > n_end) break;
                r += S;
                Cell& h = H(r);
                Cell& f = F(r);
                uint8_t& dir = D(r);
                int which = 0;                          // who holds the cell's best: 0 diagonal (h), 1 horizontal (e1), 2 vertical (f)
                if (!corner_row) {
                    h.val += sc->mtx[am * dim + b[TOWARDS5 ? n : n - 1]];
                    dir = (dir % 8) ? 8 : 0;
                    const Cell& above = H(r + S);       // the same column one row back: not yet passed in this row
                    const int x = above.val + sc->gop;
                    if (x >= F(r + S).val) { f = above; f.val = x; } else f = F(r + S);
                    f.val += sc->gep;
                    if (f.val >= h.val) which = 2;
                }
                {
                    const Cell& beside = H(r - S);      // the cell passed just before on this row
                    const int x = beside.val + sc->gop;
                    if (x >= e1.val) { e1 = beside; e1.val = x; }
                    e1.val += sc->gep;
                    if (e1.val >= (which == 2 ? f.val : h.val)) which = 1;
                }
                Cell& mx = which == 0 ? h : (which == 1 ? e1 : f);
                if (dir & 8) mx.ptr = vmf.add(m - S, n - S, mx.ptr);
                if (lcl && mx.val > best_val) { best_val = mx.val; best_p = mx.ptr; best_m = m; best_n = n; }
                if (mx.val + sp->vthr < best_val) {     // dropped off: the row ends here
                    if (peak) { if (TOWARDS5) n1 = n + 1; else n2 = n - 1; peak = false; }
                    nr = n;
                    break;
                } else if (dir % 8 == 0 && (block_is_e1 || mx.val >= block_val)) {
                    block_is_e1 = which == 1;
                    block_val = mx.val;
                    if (TOWARDS5) { if (nr < n2) n2 = nr; } else { if (nr > n1) n1 = nr; }
                    peak = true;
                }
                if (which != 0) h = mx;
                dir = (uint8_t) which;
            }
            if (peak) { if (TOWARDS5) n1 = n + 1; else n2 = n - 1; }
            if (!TOWARDS5) { ++n1; ++n2; }
        }
        *last = vmf.add(best_m, best_n, best_p);
        if (!TOWARDS5) is3end = true;
        return best_val;
    }

    int open_end(int cmode, bool lcl = true)
    {
        Trail vmf;
        int ptr = 0;
        if (cmode == 3) {
            const int ar_room = a_len - cur.ar;
            if (cur.al > ar_room) { cmode = 2; cur.ar = a_len; }
            else { cmode = 1; cur.al = 0; rec.clear(); }
        }
        const SpdpWindow w = stripe(sc->sh, cmode);
        const int scr = cmode == 1 ? end_extension<true>(&ptr, w, lcl, vmf) : end_extension<false>(&ptr, w, lcl, vmf);
        for (int p = ptr; p; ) {
            put(vmf.m[p], vmf.n[p]);
            p = vmf.prev[p];
        }
        return scr;
    }

    int shortcut(int ovr, const Bound& bab)
    {
        const int margin = sp->minl;
        int scr = 0;
        ovr = (ovr > 0 ? 0 : ovr) - 3;
        const int interval = cur.br - cur.bl - 2 * margin;
        const int cut[2] = {cur.bl + margin, cur.br - margin};
        scr -= creep_back(ovr, slmt(), bab);
        scr -= creep_fwrd(ovr, slmt(), bab);
        const int alen = cur.ar - cur.al;
        int sh = alen / 2;
        if (sc->sh < 0) {
            float f = (float) -sc->sh;
            if (f > 1.f) f /= 100;
            if (f < 0.5f) sh = (int) (alen * f);
        } else if (sc->sh < sh) sh = sc->sh;
        sh = std::max(sh, alen - margin);
        const SpdpWindow w = stripe(sh);
        const uint8_t aexg = cur.a_exgl, bexg = cur.b_exgl;
        cur.a_exgl = cur.b_exgl = 0;
        scr += dp->trcbk(cur, w, interval > 0 ? cut : nullptr, rec);
        cur.a_exgr = aexg;                              // (sic: the reference puts the saved LEFT flags into the RIGHT ones)
        cur.b_exgr = bexg;
        return scr;
    }

    // ---- Aln2s1::interpolateS ------------------------------------------------------------------------------------
    int interpolate(unsigned level, const int cmode, const Hsp* wjxt, const Bound& bab)
    {
        if (is3end) return 0;
        int agap = cur.ar - cur.al, bgap = cur.br - cur.bl;
        int ovr = std::min(agap, bgap);
        const int dgap = bgap - agap;
        const bool cont = agap <= 0;
        ++level;
        const int wlmt = level <= 3 ? sp->wl_width[level] : 0;
        const bool no_rec = ovr < wlmt;
        int iscore = NEV(), scr = 0;
        std::vector<SpdpSkl> saved;
        bool have_saved = false;
        static const bool dbg = getenv("SPDP_WALK_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[walk] interpolate level %u cmode %d a %d..%d b %d..%d agap %d bgap %d no_rec %d wjxt %d,%d bab %d %d %d %d\n",
                         level, cmode, cur.al, cur.ar, cur.bl, cur.br, agap, bgap, (int) no_rec, wjxt ? wjxt->jx : -1, wjxt ? wjxt->jy : -1,
                         bab.la, bab.lb, bab.ua, bab.ub);

        if (dgap == 0) {
            if (agap == 0) { ++joins[J_ABUT]; if (wjxt) put(wjxt->jx, wjxt->jy); return 0; }
            ++joins[J_DIAGONAL];
            iscore = diagonal();
        } else if (cmode == 1 && no_rec && wjxt) {
            if (cont) { ++joins[J_HEAD_CONT]; put(wjxt->jx, wjxt->jy); iscore = 0; }
            else if (agap < sp->elmt) {
                ++joins[J_HEAD_SHORT];
                const int m = wjxt->jx - agap, n = wjxt->jy - agap;
                if (m >= cur.al && n >= cur.bl) put(m, n);
                put(wjxt->jx, wjxt->jy);
                iscore = (int) (agap * sp->smn4);
            } else if (bgap <= 0) {
                ++joins[J_HEAD_NOGENOME];
                cur.al = cur.ar; cur.bl = cur.br;
                put(cur.al, cur.bl);
                iscore = 0;
            } else {
                ++joins[J_HEAD_EXTEND];
                std::vector<SpdpSkl> other = rec;           // first_exon writes into the file, the extension into a copy of it as it was
                const int kscore = end_exon<FIRST>(bab);
                rec.swap(other);
                iscore = open_end(cmode, false);
                if (kscore > iscore) { ++joins[J_HEAD_EXON]; iscore = kscore; rec.swap(other); }
            }
        } else if (cmode == 2 && no_rec) {
            if (agap < sp->elmt) {
                ++joins[J_TAIL_SHORT];
                put(cur.al, cur.bl);
                if (agap < 0) agap = 0;
                iscore = (int) (agap * sp->smn4);
                if (agap) put(cur.al + agap, cur.bl + agap);
            } else if (bgap <= 0) {
                ++joins[J_TAIL_NOGENOME];
                cur.al = cur.br; cur.bl = cur.br;           // (sic)
                put(cur.al, cur.bl);
                iscore = 0;
            } else {
                ++joins[J_TAIL_EXTEND];
                std::vector<SpdpSkl> other = rec;
                const int kscore = end_exon<LAST>(bab);
                rec.swap(other);
                iscore = open_end(cmode, false);
                if (kscore > iscore) { ++joins[J_TAIL_EXON]; iscore = kscore; rec.swap(other); }
            }
        } else if (cmode == 3 && cont && dgap >= sp->minl && indel_free_junction(agap, iscore, true)) {
            ++joins[J_JUNCTION];
            scr += iscore;
            iscore = 0;
        } else if (cmode == 3 && no_rec && dgap >= sp->minl) {
            if (sp->crs == 0) { iscore = micro_exon(bab); if (iscore != NEV()) ++joins[J_MICRO_EXON]; }
            if (iscore == NEV() && agap < sp->elmt) { ++joins[J_SHORTCUT]; iscore = shortcut(ovr, bab); }
        } else if (ovr <= 0 && dgap < sp->minl) {
            ++joins[J_BACKFORTH];
            iscore = back_and_forth(-ovr, bab);
        } else if (std::abs(dgap) < sp->minl) {
            ++joins[J_SMALL_DP];
            scr -= creep_back(ovr, slmt(), bab);
            scr -= creep_fwrd(ovr, slmt(), bab);
            const SpdpWindow w = stripe(std::min(sc->sh, std::abs(dgap) + 3));
            iscore = dp->trcbk(cur, w, nullptr, rec);
        } else if ((int) level < sp->qck) {
            ++joins[J_RECURSE];
            saved = rec; have_saved = true;
            iscore = seeded(level, cmode, bab);
        }
        const float dpspace = std::fabs((float) agap * (float) bgap) / 1048576.f;
        const int max_agap = (sp->desert && (agap > bgap || cmode < 3)) ? sp->desert * (4 - (int) level) : INT_MAX;
        if (iscore == NEV() && (no_rec || (int) level == sp->qck) && dpspace < 32 * sp->maxsp && agap < max_agap &&
            !(LocalC() && sp->qck == 3 && cmode < 3)) {
            const Span before = cur;
            if (cmode & 1) scr -= creep_fwrd(ovr, slmt(), bab);
            if (cmode & 2) scr -= creep_back(ovr, slmt(), bab);
            agap += before.al - cur.al + cur.ar - before.ar;
            bgap += before.bl - cur.bl + cur.br - before.br;
            if (have_saved) rec = saved;
            else { saved = rec; have_saved = true; }
            const SpdpWindow w = stripe(sc->sh);
            ++joins[J_DP];
            iscore = dp->lsp(cur, w, rec);
        }
        if (iscore == NEV()) {
            if (have_saved) rec = saved;
            if (LocalC()) {
                ++joins[J_GIVEUP_LOCALC];
                put(cmode == 1 ? cur.ar : cur.al, cmode == 1 ? cur.br : cur.bl);
                iscore = 0;
            } else if (cmode == 1) {
                ++joins[J_GIVEUP_HEAD];
                if (wjxt) { const int bl = wjxt->jy + end_margin(); if (bl > cur.bl) cur.bl = bl; }
                iscore = open_end(cmode);
            } else if (cmode == 2) {
                ++joins[J_GIVEUP_TAIL];
                if (wjxt) { const int br = bgap - wjxt->jy - end_margin(); if (br > cur.bl && br < cur.br) cur.br = br; }
                iscore = open_end(cmode);
            } else {
                ++joins[J_GIVEUP_INNER];
                iscore = Local() ? open_end(cmode) : shortcut(ovr, bab);
            }
        }
        if (dbg) fprintf(stderr, "[walk]   -> scr %d iscore %d, %zu records, last (%d,%d)\n", scr, iscore, rec.size(),
                         rec.empty() ? -1 : rec.back().m, rec.empty() ? -1 : rec.back().n);
        return scr + iscore;
    }

    // the unit whose first and last HSP join their neighbours best (seededS_ng picks among several Wilip units)
    int best_unit(const std::vector<Unit>& units, int cmode)
    {
        const Span keep = cur;
        int best = NEV(), which = -1;
        for (size_t u = 0; u < units.size(); ++u) {
            const Unit& w = units[u];
            const Hsp* jxt = w.jxt.data();
            cur.al = keep.al; cur.bl = keep.bl;
            cur.ar = jxt->jx; cur.br = jxt->jy;
            int agap = jxt->jx - cur.al;
            if (agap > 0) continue;
            int iscore = NEV(), jscore = 0;
            if (cmode == 1) jscore = w.scr;
            else if (indel_free_junction(agap, iscore, false)) jscore = w.scr + iscore;
            else continue;
            jxt = w.jxt.data() + w.num - 1;
            cur.al = jxt->jx + jxt->jlen; cur.bl = jxt->jy + jxt->jlen;
            cur.ar = jxt[1].jx; cur.br = jxt[1].jy;
            agap = jxt[1].jx - cur.al;
            if (agap > 0) continue;
            if (cmode != 2) {
                if (indel_free_junction(agap, iscore, false)) jscore += iscore;
                else continue;
            }
            if (jscore > best) { best = jscore; which = (int) u; }
        }
        cur.al = keep.al; cur.ar = keep.ar; cur.bl = keep.bl; cur.br = keep.br;
        return best > NEV() ? which : -1;
    }

    // ---- Aln2s1::seededS_ng: eimode 1 = 5' end, 2 = 3' end, 3 = internal -----------------------------------------
    int seeded(unsigned level, int eimode, const Bound& lub)
    {
        const Span at_entry = cur;
        int cmode = eimode, scr = 0;
        std::vector<Unit> units;
        std::vector<Hsp>* list = nullptr;
        int num = 0;
        const int wlmt = level <= 3 ? sp->wl_width[level] : 0;
        Bound bab = lub;
        const bool lowest = (int) level == lowest_level && !top_hsps.empty();
        if (lowest) {
            list = &top_hsps;
            num = (int) top_hsps.size() - 1;
        } else {
            if (!dp->wilip((int) level, cur, units)) { unsupported = true; return NEV(); }
            const int nwlu = (int) units.size();
            int pick = nwlu ? 0 : -1;
            if (nwlu > 1 && cur.br - cur.bl >= sp->minl) { ++joins[J_PICK_UNIT]; pick = best_unit(units, cmode); }
            if (pick >= 0) { list = &units[pick].jxt; num = units[pick].num; }
            else if (nwlu > 1) level = sp->qck - 1;
        }
        const Hsp* wjxt = nullptr;
        if (num) {
            std::vector<Hsp>& jxt = *list;
            jxt[num].jx = cur.ar;
            jxt[num].jy = cur.br;
            cur.a_exgr = 0; cur.b_exgr = 0;
            int k = 0;
            for ( ; k < num; ++k) {
                const Hsp& h = jxt[k];
                wjxt = &h;
                scr += h.jscr;
                cur.ar = h.jx; cur.br = h.jy;
                bab.ua = std::min(h.jx + h.jlen, jxt[k + 1].jx) - wlmt;
                bab.ua = std::max(bab.ua, h.jx + h.jlen / 2);
                bab.ub = h.jy + bab.ua - h.jx;
                if (cmode == 2) cmode = 3;
                const int iscore = interpolate(level, cmode, &h, bab);
                if (iscore != NEV()) {
                    scr += iscore;
                    cmode = 3;
                    cur.al = h.jx + h.jlen; cur.bl = h.jy + h.jlen;
                    cur.a_exgl = 0; cur.b_exgl = 0;
                    bab.la = cur.ar; bab.lb = cur.br;
                }
            }
            wjxt = &jxt[num];                           // the loop leaves the pointer one past the last HSP
            cur.a_exgr = at_entry.a_exgr; cur.b_exgr = at_entry.b_exgr;
            cur.ar = at_entry.ar; cur.br = at_entry.br;
            bab.ua = lub.ua; bab.ub = lub.ub;
            if (eimode == 2 || ((int) level == lowest_level && eimode == 1)) cmode = 2;
        }
        const int iscore = interpolate(level, cmode, wjxt, bab);
        if (iscore > NEV()) scr += iscore; else scr = NEV();
        cur.al = at_entry.al; cur.ar = at_entry.ar; cur.bl = at_entry.bl; cur.br = at_entry.br;
        cur.a_exgl = at_entry.a_exgl; cur.b_exgl = at_entry.b_exgl;
        if ((int) level == lowest_level && wjxt && list == &top_hsps) {
            top_hsps[num].jx = a_len; top_hsps[num].jy = b_len;
        }
        return scr;
    }

    // globalS_ng with seeding on: the record file starts with one dummy record; returns the raw score
    int run(const Span& whole)
    {
        cur = whole;
        rec.clear();
        rec.push_back({0, 0});
        is3end = false;
        const Bound bab = {cur.al, cur.bl, cur.ar, cur.br};
        return seeded((unsigned) lowest_level, 1, bab);
    }
};

// the flat unit record of SpdpHspSource::units (include/spdp.h) -> units
inline bool parse_units(const int32_t* flat, int n, std::vector<Unit>& units)
{
    units.clear();
    if (n < 1 || flat[0] < 0) return false;
    int at = 1;
    for (int u = 0; u < flat[0]; ++u) {
        if (at + 6 > n) return false;
        Unit x;
        x.num = flat[at]; x.nid = flat[at + 1]; x.tlen = flat[at + 2]; x.llmt = flat[at + 3]; x.ulmt = flat[at + 4]; x.scr = flat[at + 5];
        at += 6;
        if (x.num < 0 || at + 5 * (x.num + 1) > n) return false;
        for (int j = 0; j <= x.num; ++j, at += 5) x.jxt.push_back({flat[at], flat[at + 1], flat[at + 2], flat[at + 3], flat[at + 4]});
        units.push_back(std::move(x));
    }
    return true;
}

// points a walk at one query's inputs; phs5 / phs3 are copied (the walk edits them) or derived as
// Exinon::intron53_n derives them from the canonical-site levels (src/codepot.cc:504-518, algmode.any != 2)
inline bool bind_problem(SeedWalk& w, const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpProblem* p,
                         const SpdpJuxt* hsps, int n_hsps, int lowest_level)
{
    if (!sc || !sp || !p || !p->a || !p->b || !p->sig5 || !p->sig3 || !p->cano5 || !p->cano3 || !p->dinc ||
        !sc->intpen || sc->intpen_len <= 0 || sp->qck < 1 || sp->qck > 3) return false;
    w.a = p->a; w.a_len = p->a_len; w.b = p->b; w.b_len = p->b_len;
    w.sig5 = p->sig5; w.sig3 = p->sig3; w.cano5 = p->cano5; w.cano3 = p->cano3; w.dinc = p->dinc; w.cip = p->cip;
    w.sc = sc; w.sp = sp;
    w.lowest_level = lowest_level;
    const int N = p->b_len + 1;
    if (p->phs5 && p->phs3) {
        w.phs5.bind(p->phs5); w.phs3.bind(p->phs3);
    } else {
        w.phs5.derive(N); w.phs3.derive(N);
        std::vector<int8_t>& q5 = w.phs5.own;
        std::vector<int8_t>& q3 = w.phs3.own;
        for (int n = std::max(1, p->b_left); n < std::min(N - 1, p->b_right + 1); ++n) {
            if (q5[n] == -2 && p->cano5[n]) {
                q5[n] = 0;
                if (p->cano5[n] > 1) { q5[n + 1] = 1; q5[n - 1] = q5[n - 1] == 1 ? 2 : -1; }
            }
            if (q3[n] == -2 && p->cano3[n]) {
                q3[n] = 0;
                if (p->cano3[n] > 1) { q3[n + 1] = 1; q3[n - 1] = q3[n - 1] == 1 ? 2 : -1; }
            }
        }
    }
    // canonical-site levels by dinucleotide class, as Exinon::intron53_c assigns them (src/codepot.cc:435-475): classes
    // are 4 * first + second base with A C G T = 0 .. 3; GT-AG = 3, GC-AG / AT-AC = 3 / 2, the rest by algmode.any
    {
        static const uint8_t lac[4] = {0, 2, 3, 1}, lgt[4] = {0, 0, 3, 1};
        const int any = sp->any & 3;
        const uint8_t base = any == 3 ? 1 : 0, gt = lgt[any], ac = lac[any], bo = sp->both_ori ? 1 : 0;
        uint8_t* f5 = w.f5;
        uint8_t* f3 = w.f3;
        for (int c = 0; c < 16; ++c) f5[c] = f3[c] = base;
        enum { AA, AC, AG, AT, CA, CC, CG, CT, GA, GC, GG, GT, TA, TC, TG, TT };
        f3[AA] = ac;
        f3[AC] = 2; if (bo) f5[AC] = 1;
        f3[AG] = 3;
        f5[AT] = 2; f3[AT] = ac;
        f3[CG] = gt;
        f5[CT] = gt; if (bo) f3[CT] = 1;
        f5[GA] = gt;
        f5[GC] = 3;
        f5[GG] = gt; f3[GG] = gt;
        f5[GT] = 3; if (bo) f3[GT] = 1;
        f3[TG] = gt;
        f5[TT] = gt;
    }
    w.top_hsps.clear();
    if (hsps && n_hsps > 0)
        for (int j = 0; j <= n_hsps; ++j) w.top_hsps.push_back({hsps[j].jx, hsps[j].jy, hsps[j].jlen, hsps[j].nid, hsps[j].jscr});
    return true;
}

}   // namespace spdp_seed
#endif
