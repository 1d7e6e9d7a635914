// spdp_walk.h -- the seeded path of alignS_ng / alignH_ng: one query's walk over its HSPs, host side, ONE template.
//
// What it has to reproduce (ogotoh/spaln v3.0.7): Aln2s1::globalS_ng with algmode.qck set -> seededS_ng -> interpolateS and
// the closed-form joins around it (src/fwd2s1.cc:1899-2672), and the protein twin Aln2h1::seededH_ng / interpolateH
// (src/fwd2h1.cc:2232-3290).  The reference walks a query's HSPs on a worker thread and calls its DP engines from deep
// inside; here the walk runs on a fiber and parks every DP call (DpBackend) until a device batch has served it.
//
// How this file is organised -- not as the reference is:
//   * Walk<Path> is the walk: the loop over HSPs (`seeded`), the choice among several units (`pick_unit`) and the gap
//     filler (`fill_gap`), which is a TABLE: rows of (when, how) tried top to bottom, first applicable row fills the gap,
//     then the common fall-backs (full DP, give up into an end extension).  The order of the rows is the order in which
//     the reference tests its conditions -- that order decides ties, so it is data here, visible in one place.
//   * CdnaPath / ProteinPath supply what differs between the two alignments: how many genome positions a query row
//     spans (STEP 1 or 3), what one aligned row scores, the band, the junction search, the terminal-exon searches, the
//     X-drop end extension.  Geometry that both share (diagonal runs, creeping along a diagonal, switching diagonals
//     inside an overlap) is written once over `row_score`.
//   * ExactFinder<Code> is one Boyer-Moore-style scan for both alphabets (frames = |step|).
// Arithmetic quirks of the reference that change results are kept and marked (quirk: ...); they are pinned by the
// reference's own seeded runs (tests/golden q_* / qh_*, tools/seed_fuzz*.py).
#ifndef SPDP_WALK_H_
#define SPDP_WALK_H_

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../../include/spdp.h"
#include "spdp_gencode.h"

namespace spdp_seed {

struct Span { int al, ar, bl, br; uint8_t a_exgl, a_exgr, b_exgl, b_exgr; };     // active ranges + end flags of both sequences
struct Bound { int la, lb, ua, ub; };                                             // how far a join may reach into the neighbours
struct Hsp { int jx, jy, jlen, nid, jscr; };
struct Unit { int num, nid, tlen, llmt, ulmt, scr; std::vector<Hsp> jxt; };      // a chain of HSPs (jxt: num + 1 slots)
typedef std::vector<SpdpSkl> Records;

struct DpBackend {
    virtual ~DpBackend() {}
    virtual int lsp(const Span& s, const SpdpWindow& w, Records& rec) = 0;        // the linear-space ladder on the span
    // one traceback sweep; cut = genomic range the sweep jumps over (or null); introns = false: no splicing (the protein
    // path's small-gap DP; the cDNA engines have no such switch and ignore it)
    virtual int trcbk(const Span& s, const SpdpWindow& w, bool introns, const int* cut, Records& rec) = 0;
    virtual bool wilip(int level, const Span& s, std::vector<Unit>& units) = 0;   // the HSP search of a recursion level
};

// ---- exact occurrences of the query's end in the genomic span ------------------------------------------------------------
// The reference finds terminal exons with its own Boyer-Moore variant (src/boyer_moore.cc): relaxed letter matching, shift
// tables built with a (different) relaxed relation, a fixed jump after a hit -- so it skips some true occurrences, and
// WHICH occurrence is met first decides the exon.  The scan is therefore kept with its tables; Code supplies the two
// relations.  step = +-1 (nucleotides) or +-3 (a protein pattern against tron codes, three frames scanned side by side).
struct NucCode {
    static bool text_eq(uint8_t t, uint8_t p) { return t && p && ((t - 1) & (p - 1)); }     // base sets intersect
    static bool table_eq(uint8_t x, uint8_t y) { return text_eq(x, y); }
    static void alias(std::vector<int>&) {}
};
struct TronCode {
    static bool text_eq(uint8_t t, uint8_t p) { return t == p || (t == 23 && p == 18) || p == 2; }   // SER2 in the text, AMB in the pattern
    static bool table_eq(uint8_t x, uint8_t y) { return x == y || x == 2 || y == 2; }
    static void alias(std::vector<int>& by_code) { by_code[23] = by_code[18]; }
};

template <class Code>
class ExactFinder {
    const uint8_t* text; int tlen, origin;
    std::vector<uint8_t> pat; int plen;
    std::vector<int> by_code, by_suffix, pending;
    int step, frames, after_hit, idx[3];
    int frame_fwd(int i, int k) const { static const char t[3][3] = {{0, 1, 2}, {2, 0, 1}, {1, 2, 0}}; return frames == 1 ? i : i + t[i % 3][k]; }
    int frame_bwd(int i, int k) const { static const char t[3][3] = {{0, 2, 1}, {1, 0, 2}, {2, 1, 0}}; return frames == 1 ? i : i - t[i % 3][k]; }
public:
    ExactFinder(const uint8_t* b, int bl, int br, const uint8_t* a, int al, int ar, int step_, int n_codes = 256)
        : text(b + bl), tlen(br - bl), origin(bl), pat(a + al, a + ar), plen(ar - al), step(step_), frames(std::abs(step_))
    {
        const bool back = step < 0;
        pat.push_back(0);
        if (back) std::reverse(pat.begin(), pat.begin() + plen);
        by_code.assign(std::max(n_codes, 256), plen);
        for (int j = 0; j < plen; ++j) by_code[pat[j]] = plen - 1 - j;
        Code::alias(by_code);
        const int cells = std::max(plen, 1);
        by_suffix.resize(cells);
        std::vector<int> link(cells);
        for (int j = 0; j < plen; ++j) by_suffix[j] = 2 * plen - 1 - j;
        int j = plen;
        for (int k = plen; --k >= 0; ) {
            link[k] = j;
            pat[plen] = pat[k];                                  // sentinel: the chain below always ends
            while (!Code::table_eq(pat[j], pat[k])) {
                by_suffix[j] = std::min(by_suffix[j], plen - 1 - k);
                j = link[j];
            }
            --j;
        }
        after_hit = std::max(j + 1, 2) * step;
        for (int s = j, q = 0; q < plen; ++q) {
            by_suffix[q] = std::min(by_suffix[q], s + plen - q);
            if (q >= s) s = s >= 0 ? link[s] : 0;                // quirk: a one-letter repeat leaves s = -1; the reference reads link[-1] (0 with glibc)
        }
        if (back) {
            std::reverse(pat.begin(), pat.begin() + plen);
            std::reverse(by_suffix.begin(), by_suffix.begin() + plen);
        }
        for (int k = 0; k < frames; ++k) idx[k] = back ? frame_bwd(tlen, k) : frame_fwd(0, k);
    }
    int reach() const
    {
        int v = idx[0];
        for (int k = 1; k < frames; ++k) v = step > 0 ? std::min(v, idx[k]) : std::max(v, idx[k]);
        return v;
    }
    bool finished() const { return step > 0 ? reach() >= tlen : reach() <= 0; }
    bool scanned(int n) const { return pending.empty() && (step > 0 ? reach() >= n - origin : reach() <= n - origin); }
    // the next occurrence (position in b of its first code) or -1; from / upto (positions in b, -1 = where the scan stands /
    // the end of the text) restart and bound the scan as the reference's nexthit3 does
    int next(int from = -1, int upto = -1)
    {
        if (!pending.empty()) { const int v = pending.back(); pending.pop_back(); return v + origin; }
        const int lo = from < 0 ? -1 : std::max(from - origin, 0), hi = upto < 0 ? -1 : std::max(upto - origin, 0);
        for (int k = 0; k < frames; ++k) {
            if (step > 0) {
                int i = (lo >= 0 ? frame_fwd(lo, k) : idx[k]) + step * (plen - 1);
                const int bound = hi >= 0 ? hi : tlen;
                idx[k] = frame_fwd(bound, k);
                while (i < bound) {
                    int j = plen - 1;
                    while (j >= 0 && Code::text_eq(text[i], pat[j])) { i -= step; --j; }
                    if (j < 0) { pending.push_back(i + step); idx[k] = i + after_hit; break; }
                    i += step * std::max(by_code[text[i]], by_suffix[j]);
                }
            } else {
                int i = (hi >= 0 ? frame_bwd(hi, k) : idx[k]) + step * (plen - 1);
                const int bound = lo >= 0 ? lo : 0;
                idx[k] = frame_bwd(bound, k);
                while (i >= bound) {
                    int j = 0;
                    while (j < plen && Code::text_eq(text[i], pat[j])) { i -= step; ++j; }
                    if (j >= plen) { pending.push_back(i + step * plen); idx[k] = i + after_hit; break; }
                    i += step * std::max(by_code[text[i]], by_suffix[j]);
                }
            }
        }
        if (pending.empty()) return -1;
        std::sort(pending.begin(), pending.end());               // nearest first: smallest going right, largest going left
        if (step > 0) std::reverse(pending.begin(), pending.end());
        const int v = pending.back(); pending.pop_back();
        return v + origin;
    }
};

// the phase marks of a window as one walk sees them: the caller's array plus the few marks the walk itself has set
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

// canonical-site levels by dinucleotide class, as the reference's Exinon::intron53_c assigns them (src/codepot.cc:435-475):
// class = 4 * first + second base, A C G T = 0 .. 3; GT-AG = 3, GC-AG / AT-AC = 3 / 2, the rest by algmode.any
inline void site_levels(const SpdpSeedParams* sp, uint8_t f5[16], uint8_t f3[16])
{
    static const uint8_t lac[4] = {0, 2, 3, 1}, lgt[4] = {0, 0, 3, 1};
    const int any = sp->any & 3;
    const uint8_t base = any == 3 ? 1 : 0, gt = lgt[any], ac = lac[any], bo = sp->both_ori ? 1 : 0;
    for (int c = 0; c < 16; ++c) f5[c] = f3[c] = base;
    enum { AA, AC, AG, AT, CA, CC, CG, CT, GA, GC, GG, GT, TA, TC, TG, TT };
    struct { int cls; uint8_t v5, v3; bool set5, set3; } rows[] = {
        {AA, 0, ac, false, true}, {AC, 1, 2, bo != 0, true}, {AG, 0, 3, false, true}, {AT, 2, ac, true, true},
        {CG, 0, gt, false, true}, {CT, gt, 1, true, bo != 0}, {GA, gt, 0, true, false}, {GC, 3, 0, true, false},
        {GG, gt, gt, true, true}, {GT, 3, 1, true, bo != 0}, {TG, 0, gt, false, true}, {TT, gt, 0, true, false}};
    for (const auto& r : rows) { if (r.set5) f5[r.cls] = r.v5; if (r.set3) f3[r.cls] = r.v3; }
}

// ---- what both paths share: the walk's state and the arithmetic on it ------------------------------------------------------
struct WalkState {
    const uint8_t* a = nullptr; int a_len = 0;
    const uint8_t* b = nullptr; int b_len = 0;
    const int16_t* sig5 = nullptr; const int16_t* sig3 = nullptr;
    const uint8_t* dinc = nullptr;
    const int32_t* cip = nullptr;
    PhaseMarks phs5, phs3;                      // the walk marks the junctions it accepts
    uint8_t f5[16] = {0}, f3[16] = {0};
    const SpdpSeedParams* sp = nullptr;
    DpBackend* dp = nullptr;
    int lowest_level = 0;                       // the level whose HSPs came with the query (b->jxt)
    std::vector<Hsp> top_hsps;

    Span cur{};
    Records rec;                                // the record file, dummy record first
    bool is3end = false;
    bool unsupported = false;                   // the walk met a state this restatement does not serve
    int why = 0;
    void mark(int line) { unsupported = true; if (!why) { why = line; if (getenv("SPDP_WALK_DEBUG")) fprintf(stderr, "walk: not served, line %d\n", line); } }

    static int NEV() { return SPDP_NEVSEL; }
    bool Local() const { return (sp->lcl & 16) != 0; }
    bool LocalC() const { return Local() && (sp->lcl & 32); }
    void put(int m, int n) { rec.push_back({m, n}); }
    int slmt() const { return sp->vthr / 2; }
    void restore_ranges(const Span& s) { cur.al = s.al; cur.ar = s.ar; cur.bl = s.bl; cur.br = s.br; }
    static int canon_rank(int c5, int c3) { return ((c5 == 3 && c3 == 3) || (c5 == 2 && c3 == 2) || (c5 == 1 && c3) || (c5 && c3 == 1)) ? c5 + c3 : 0; }
};

struct Trail {                                  // the record chain of an end extension
    std::vector<int> m, n, prev;
    int add(int m_, int n_, int p) { m.push_back(m_); n.push_back(n_); prev.push_back(p); return (int) m.size() - 1; }
};

// ============================================================================================================================
// cDNA x genome
// ============================================================================================================================
struct CdnaPath : WalkState {
    enum { STEP = 1 };
    typedef SpdpScoring Scoring;
    typedef SpdpProblem Problem;
    const SpdpScoring* sc = nullptr;
    const uint8_t* cano5 = nullptr; const uint8_t* cano3 = nullptr;
    bool a_sens = false;
    enum { J_ABUT, J_DIAGONAL, J_HEAD_CONT, J_HEAD_SHORT, J_HEAD_NOGENOME, J_HEAD_EXTEND, J_HEAD_EXON, J_TAIL_SHORT,
           J_TAIL_NOGENOME, J_TAIL_EXTEND, J_TAIL_EXON, J_JUNCTION, J_MICRO_EXON, J_SHORTCUT, J_BACKFORTH, J_SMALL_DP,
           J_RECURSE, J_DP, J_GIVEUP_LOCALC, J_GIVEUP_HEAD, J_GIVEUP_TAIL, J_GIVEUP_INNER, J_PICK_UNIT, J_COUNT };
    int joins[J_COUNT + 1] = {0};

    // policy constants of the gap filler
    static bool joined(int s) { return s != SPDP_NEVSEL; }
    static bool creep_on(int d, int limit) { return std::abs(d) < limit; }
    static bool small_gap(int dgap, int minl) { return std::abs(dgap) < minl; }
    static bool below_rec_limit(int ovr, int wlmt) { return ovr < wlmt; }
    enum { SWITCH_QUERY_STEP = 1, SWITCH_KEEPS_NEGATIVE_N = 1, GIVEUP_COUNTS_SHORTCUT = 0, ABUT_ENDS_THE_JOIN = 1, HEAD_NEEDS_HSP = 1 };
    int rec_limit(unsigned level, int) const { return level <= 3 ? sp->wl_width[level] : 0; }
    bool junction_gap(int agap) const { return agap <= 0; }
    bool dp_affordable(int agap, int bgap, int cmode, unsigned level) const
    {
        const float dpspace = std::fabs((float) agap * (float) bgap) / 1048576.f;
        const int max_agap = (sp->desert && (agap > bgap || cmode < 3)) ? sp->desert * (4 - (int) level) : INT_MAX;
        return dpspace < 32 * sp->maxsp && agap < max_agap;
    }
    int end_margin() const { return (int) ((sp->vthr + sc->gop) / sc->gep); }
    int shortcut_cut_end(int from, int interval) const { return from + interval; }
    int shortcut_shoulder_floor(int alen, int margin) const { return alen - margin; }
    void shortcut_flags(uint8_t aexg, uint8_t bexg) { cur.a_exgr = aexg; cur.b_exgr = bexg; }    // quirk: the saved LEFT flags land in the RIGHT ones
    void prepare_top_hsps() {}

    int sim(int i, int j) const { return sc->mtx[a[i] * sc->mtx_dim + b[j]]; }
    int row_score(int m, int n) const { return sim(m, n); }                 // query row m on the genome position(s) starting at n
    int switch_score(int m, int n) const { return sim(m, n); }
    int gap_penalty(int i) const { return i == 0 ? 0 : (i > sp->codonk1 ? sc->lgop + i * sc->lgep : sc->gop + i * sc->gep); }
    int diagonal_shift_cost(int dr) const { return gap_penalty(dr); }
    int int_pen(int len) const { return len < 0 ? SHRT_MIN : sc->intpen[std::min(len, sc->intpen_len - 1)]; }
    int lvl5(int n) const { return cano5[n] ? (f5[dinc[n] >> 4] ? f5[dinc[n] >> 4] : cano5[n]) : 0; }
    int lvl3(int n) const { return cano3[n] ? (f3[dinc[n] & 15] ? f3[dinc[n] & 15] : cano3[n]) : 0; }
    int is_canon(int d, int ac) const { return canon_rank(lvl5(d), lvl3(ac)); }
    int pair_signal(int m, int n) const { return sig5[m] + sig3[n] + sc->t53[16 * (dinc[m] >> 4) + (dinc[n] & 15)]; }

    SpdpWindow band(int sh, int cmode = 0) const
    {
        SpdpWindow w;
        if (sh < 0) sh = -sh * std::min(cur.ar - cur.al, cur.br - cur.bl) / 100;
        w.up = cur.br - cur.ar;
        w.lw = cur.bl - cur.al;
        if (cmode == 1) w.lw = w.up; else if (cmode == 2) w.up = w.lw; else if (w.up < w.lw) std::swap(w.up, w.lw);
        w.up = std::min(w.up + sh, cur.br - cur.al);
        w.lw = std::max(w.lw - sh, cur.bl - cur.ar);
        w.width = w.up - w.lw + 3;
        return w;
    }
    int scalar_shoulder() const { return sc->sh; }

    // ---- the junction inside an overlap: two HSPs abut / overlap on the query and lie an intron apart -------------------
    // Walk the shared diagonal back from the left ends (as far as the two genomic copies agree, at least over the overlap),
    // then forward again trying every position as the junction: signal of the pair minus the match score counted twice.
    bool junction(int agap, int& iscr, bool write)
    {
        const int ilen = cur.br - cur.bl - agap;
        if (ilen < sp->minl) { iscr = gap_penalty(ilen); return true; }
        const int overlap = 1 - agap;
        const int reach = std::min(std::min(cur.al, cur.bl), overlap + 16);
        std::vector<int> back(reach + 2, 0);
        int depth = 0, probed = 0;                                 // probed: genome positions looked at (one more than depth after a mismatch)
        for (int sum = 0; depth < reach; ) {
            const int gb = cur.bl - ++probed;
            if (!(b[gb] == b[gb + ilen] || depth < overlap)) break;
            ++depth;
            back[depth] = sum += sim(cur.al - depth, gb);
        }
        // back[t] after the flip: score of the diagonal from junction candidate t to the left ends
        std::reverse(back.begin(), back.begin() + depth + 1);
        SpdpSkl best = {cur.al - depth, cur.bl - depth};
        iscr = NEV();
        const int passes = sp->crs ? 1 : 2;                   // second pass: any position, canonical or not
        for (int pass = 0; pass < passes && iscr == NEV(); ++pass) {
            int fwd = 0;
            for (int t = 0, m = cur.al - depth, n = cur.bl - depth; n <= cur.bl; ++t, ++m, ++n) {
                const int rank = is_canon(n, n + ilen);
                if (pass || rank) {
                    int x = pair_signal(n, n + ilen);
                    if (cip && m >= 0 && m <= a_len && cip[m] && rank > 3) x += cip[m];      // an annotated intron position of the query
                    const int y = x - fwd - back[t];
                    if (y > iscr) { best.m = m; best.n = n; iscr = y; }
                }
                fwd += sim(m, cur.bl + ilen - probed + 1 + t);      // quirk: the acceptor-side diagonal starts behind the last position PROBED
            }
        }
        if (iscr <= NEV()) return false;
        if (write) {
            rec.push_back(best);
            phs5.set(best.n, 0);
            best.n += ilen;
            rec.push_back(best);
            phs3.set(best.n, 0);
            iscr += int_pen(ilen);
        }
        return true;
    }

    // ---- the splice site nearest to an open end of the genomic span (SIDE 5: donor around b.left, 3: acceptor around b.right)
    template <int SIDE>
    int nearest_site(const Bound& bab) const
    {
        const int from = SIDE == 5 ? cur.bl : cur.br, a0 = SIDE == 5 ? cur.al : cur.ar;
        auto sig = [&](int n) { return SIDE == 5 ? sig5[n] : sig3[n]; };
        auto phs = [&](int n) { return SIDE == 5 ? phs5[n] : phs3[n]; };
        auto strong = [&](int n, bool lax) { return sig(n) > (SIDE == 5 ? sp->gc_sig5 : 0) || (lax && phs(n) == 0); };
        for (int attempt = 0; ; ++attempt) {
            const bool lax = attempt > 0;
            // upstream: until a strong site, the reach limit, or (same species) a mismatch on the diagonal
            int up = from;
            bool hit = false;
            for (int qa = a0, qb = from, stop = std::max(bab.la, cur.al - 9); qa > stop && up > bab.lb; --up) {
                if ((hit = strong(up, lax))) break;
                if (!sp->crs) { --qa; --qb; if (a[qa] != b[qb]) break; }
            }
            if (up == from && hit) return up;
            // downstream likewise; `seen` is the position whose signal was looked at last
            int down = from, seen = from;
            for (int qa = a0, qb = from, stop = std::min(cur.al + 9, bab.ua); qa < stop; ) {
                if (!(++down < bab.ub)) break;
                ++seen;
                if (strong(seen, lax)) break;
                if (!sp->crs) { const bool same = a[qa] == b[qb]; ++qa; ++qb; if (!same) break; }
            }
            if (!lax && sig(up) <= 0 && sig(seen) <= 0) continue;          // nothing worth a site: once more, laxer
            if (phs(up) && phs(seen)) return -1;
            if (phs(up)) return down;
            if (phs(seen)) return up;
            if (from - up == down - from) return sig(up) > sig(seen) ? up : down;
            return from - up < down - from ? up : down;
        }
    }

    // ---- a short exon of the query lost between two HSPs: both flanking sites fixed, the exon placed where it pays best ---
    int micro_exon(const Bound& bab)
    {
        const int l = nearest_site<5>(bab);
        if (l < 0) return NEV();
        const int r = nearest_site<3>(bab);
        if (r < 0) return NEV();
        const Span before = cur;
        cur.al += l - cur.bl; cur.bl = l;
        cur.ar += r - cur.br; cur.br = r;
        const int alen = cur.ar - cur.al;
        if (alen <= 0) {                                           // nothing left of the query: one intron, perhaps an overlap to take back
            if (alen < 0) put(cur.al + alen, cur.bl);
            put(cur.al, cur.bl);
            put(cur.al, cur.br);
            return int_pen(r - l) + pair_signal(l, r);
        }
        int best = int_pen(r - l), at = -1;
        for (int n5 = cur.bl + sp->minl, last = cur.br - alen - sp->minl; n5 < last; ++n5) {
            if (phs3[n5] || phs5[n5 + alen]) continue;
            int ms = 0;
            for (int t = 0; t < alen; ++t) ms += sim(cur.al + t, n5 + t);
            const int scr = (int) (sp->w2 * ms + pair_signal(l, n5) + pair_signal(n5 + alen, r) + int_pen(n5 - l) + int_pen(r - n5 - alen));
            if (scr > best) { best = scr; at = n5; }
        }
        if (at < 0) { restore_ranges(before); return NEV(); }
        put(cur.al, cur.bl);
        if (at != cur.bl) put(cur.al, at);
        put(cur.al + alen, at != cur.bl ? at + alen : cur.bl);
        put(cur.al + alen, cur.br);
        return best;
    }

    // ---- a terminal exon too short for the HSP search ------------------------------------------------------------------------
    // An exact copy of the query's end somewhere in the genomic span, an intron away from the nearest splice site (HEAD:
    // upstream of the acceptor at b.right; TAIL: downstream of the donor at b.left); failing that the best ungapped
    // placement at a canonical site.
    template <bool HEAD>
    int terminal_exon(const Bound& bab)
    {
        const Span before = cur;
        auto give_up = [&]() { restore_ranges(before); return NEV(); };
        const int site = HEAD ? nearest_site<3>(bab) : nearest_site<5>(bab);
        if (site < 0) return give_up();
        if (HEAD) { cur.ar += site - cur.br; cur.br = site; } else { cur.al += site - cur.bl; cur.bl = site; }
        if (cur.al >= cur.ar || cur.bl >= cur.br) return give_up();
        if (HEAD && (cur.ar == 0 || cur.br == 0)) { put(cur.ar, cur.br); return 0; }
        const int alen = cur.ar - cur.al;
        auto intron = [&](int pos, int& don, int& acc) { don = HEAD ? pos + cur.ar : site; acc = HEAD ? site : pos; };
        int best = NEV(), pos = -1;
        ExactFinder<NucCode> find(b, cur.bl, cur.br, a, cur.al, cur.ar, HEAD ? -1 : 1, sc->mtx_dim);
        while (!find.finished()) {
            const int f = find.next();
            if (f < 0) continue;
            int don, acc;
            intron(f, don, acc);
            if (!is_canon(don, acc)) continue;
            const int s = int_pen(acc - don) + pair_signal(don, acc);
            if (s > best) { best = s; pos = f; }
        }
        int self = 0;                                              // the query's end against itself
        for (int t = cur.al; t < cur.ar; ++t) self += sc->mtx[a[t] * sc->mtx_dim + a[t]];
        if (pos >= 0) best += self;
        else {
            const int perfect = (int) (self * sp->w2);
            const int first = HEAD ? cur.br - cur.ar - sp->minl : cur.bl + sp->minl;
            const int stop = HEAD ? cur.bl - 1 : cur.br - alen;
            for (int n = first; HEAD ? n > stop : n < stop; n += HEAD ? -1 : 1) {
                const int nd = HEAD ? n + cur.ar : site, na = HEAD ? site : n;
                if (!is_canon(nd, na)) continue;
                int ms = 0;
                for (int t = cur.al; t < cur.ar; ++t) ms += sim(t, n + (t - cur.al));
                const int s = (int) (((HEAD ? sig5[nd] : sig3[na]) + int_pen(na - nd)) + sp->w2 * ms);
                if (s > best) { pos = n; if (ms == perfect) break; best = s; }   // quirk: a perfect copy is taken without its score
            }
            if (pos < 0) return give_up();
        }
        if (HEAD) {
            cur.bl = pos;
            put(cur.al, cur.bl); put(cur.ar, cur.bl + cur.ar); put(cur.ar, cur.br);
        } else {
            put(cur.al, cur.bl); put(cur.al, pos); put(cur.ar, pos + alen);
        }
        return best;
    }

    // ---- rows of the gap filler that only this path has: the open end of the query next to the first / last HSP ---------
    template <class W>
    int head_join(W& w, int agap, int bgap, bool cont, const Hsp* wjxt, int cmode, const Bound& bab)
    {
        if (cont) { ++joins[J_HEAD_CONT]; put(wjxt->jx, wjxt->jy); return 0; }
        if (agap < sp->elmt) {
            ++joins[J_HEAD_SHORT];
            if (wjxt->jx - agap >= cur.al && wjxt->jy - agap >= cur.bl) put(wjxt->jx - agap, wjxt->jy - agap);
            put(wjxt->jx, wjxt->jy);
            return (int) (agap * sp->smn4);
        }
        if (bgap <= 0) { ++joins[J_HEAD_NOGENOME]; cur.al = cur.ar; cur.bl = cur.br; put(cur.al, cur.bl); return 0; }
        ++joins[J_HEAD_EXTEND];
        Records with_exon = rec;                                // the exon search writes into the file, the extension into a copy of it as it was
        const int exon = terminal_exon<true>(bab);
        rec.swap(with_exon);
        int s = w.open_end(cmode, false);
        if (exon > s) { ++joins[J_HEAD_EXON]; s = exon; rec.swap(with_exon); }
        return s;
    }
    template <class W>
    int tail_join(W& w, int agap, int bgap, int cmode, Bound& bab)
    {
        if (agap < sp->elmt) {
            ++joins[J_TAIL_SHORT];
            put(cur.al, cur.bl);
            agap = std::max(agap, 0);
            if (agap) put(cur.al + agap, cur.bl + agap);
            return (int) (agap * sp->smn4);
        }
        if (bgap <= 0) { ++joins[J_TAIL_NOGENOME]; cur.al = cur.br; cur.bl = cur.br; put(cur.al, cur.bl); return 0; }   // quirk: a.left takes b.right
        ++joins[J_TAIL_EXTEND];
        Records with_exon = rec;
        const int exon = terminal_exon<false>(bab);
        rec.swap(with_exon);
        int s = w.open_end(cmode, false);
        if (exon > s) { ++joins[J_TAIL_EXON]; s = exon; rec.swap(with_exon); }
        return s;
    }
    // several units: how well the first and last HSP of one join the neighbours
    bool unit_ends(const Unit& u, int cmode, const Span& keep, int& jscore)
    {
        const Hsp& first = u.jxt[0];
        cur.al = keep.al; cur.bl = keep.bl; cur.ar = first.jx; cur.br = first.jy;
        int agap = first.jx - cur.al, s = NEV();
        if (agap > 0) return false;
        jscore = u.scr;
        if (cmode != 1) { if (!junction(agap, s, false)) return false; jscore += s; }
        const Hsp& last = u.jxt[u.num - 1];
        const Hsp& slot = u.jxt[u.num];
        cur.al = last.jx + last.jlen; cur.bl = last.jy + last.jlen; cur.ar = slot.jx; cur.br = slot.jy;
        agap = slot.jx - cur.al;
        if (agap > 0) return false;
        if (cmode != 2) { if (!junction(agap, s, false)) return false; jscore += s; }
        return true;
    }

    // ---- the intron-less X-drop extension of an open end ----------------------------------------------------------------------
    // Row by row along the query away from the last HSP, columns inside a band that follows the running best cell; a row ends
    // where the score has dropped Vthr below the best end cell so far.  Sequential by construction, a few thousand cells.
    struct Cell { int val, ptr; };
    template <bool TOWARDS5>
    int end_extension(int* last, const SpdpWindow& w, bool lcl, Trail& vmf)
    {
        const int S = TOWARDS5 ? -1 : 1;
        const int NEVv = NEV(), dim = sc->mtx_dim, width = w.width;
        const Cell black = {NEVv, 0};
        if (width < 3) { unsupported = true; *last = 0; return NEVv; }
        std::vector<Cell> buf(2 * (size_t) width, black);        // H and F by diagonal r = n - m
        std::vector<uint8_t> dirs(width, 1);
        auto H = [&](int r) -> Cell& { return buf[r - w.lw + 1]; };
        auto F = [&](int r) -> Cell& { return buf[width + r - w.lw + 1]; };
        auto D = [&](int r) -> uint8_t& { return dirs[r - w.lw + 1]; };
        const int m_corner = TOWARDS5 ? cur.ar : cur.al, m_last = TOWARDS5 ? cur.al : cur.ar, n_corner = TOWARDS5 ? cur.br : cur.bl;
        int best_val = lcl ? 0 : NEVv, best_m = m_corner, best_n = n_corner, best_p = 0;
        vmf.add(0, 0, 0);
        {   // the corner cell and the gap that leaves it along the genome
            int r = n_corner - m_corner;
            H(r).val = 0;
            H(r).ptr = vmf.add(m_corner, n_corner, 0);
            const int rr = TOWARDS5 ? std::min(w.up, cur.br - cur.al) : std::max(w.lw, cur.bl - cur.ar);
            for (int i = 1; TOWARDS5 ? ++r <= rr : --r >= rr; ++i) {
                H(r) = H(r + S);
                H(r).val += (i == 1 ? sc->gop : 0) + sc->gep;
                F(r) = H(r);
            }
        }
        int m = m_corner;
        if (TOWARDS5 ? !cur.a_exgr : !cur.a_exgl) m -= S;       // global end: the corner row is swept as well
        int n1 = m + w.lw, n2 = m + w.up + 1;
        for (;;) {
            m += S;
            if (TOWARDS5 ? m < cur.al : m > cur.ar) break;
            if (TOWARDS5) { --n1; --n2; }
            lcl = lcl || m == m_last;
            int n = TOWARDS5 ? std::min(n2, cur.br) : std::max(n1, cur.bl);
            const int n_end = TOWARDS5 ? std::max(n1, cur.bl) : std::min(n2, cur.br);
            int r = n - m, nr = n - S;
            bool peak = false, block_is_e1 = false;
            Cell e1 = black;
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
                int which = 0;                                  // 0 diagonal (h), 1 horizontal (e1), 2 vertical (f)
                if (!corner_row) {
                    h.val += sc->mtx[am * dim + b[TOWARDS5 ? n : n - 1]];
                    dir = (dir % 8) ? 8 : 0;
                    const Cell& above = H(r + S);
                    const int x = above.val + sc->gop;
                    if (x >= F(r + S).val) { f = above; f.val = x; } else f = F(r + S);
                    f.val += sc->gep;
                    if (f.val >= h.val) which = 2;
                }
                {
                    const Cell& beside = H(r - S);
                    const int x = beside.val + sc->gop;
                    if (x >= e1.val) { e1 = beside; e1.val = x; }
                    e1.val += sc->gep;
                    if (e1.val >= (which == 2 ? f.val : h.val)) which = 1;
                }
                Cell& mx = which == 0 ? h : (which == 1 ? e1 : f);
                if (dir & 8) mx.ptr = vmf.add(m - S, n - S, mx.ptr);
                if (lcl && mx.val > best_val) { best_val = mx.val; best_p = mx.ptr; best_m = m; best_n = n; }
                if (mx.val + sp->vthr < best_val) {             // dropped off: the row ends here
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
};

// ============================================================================================================================
// protein x three-frame genome.  A DP column n is a nucleotide position of the tron sequence, a row m an amino acid; the
// codon that ends at column n is b[n - 2].
// ============================================================================================================================
struct ProteinPath : WalkState {
    enum { STEP = 3 };
    typedef SpdpScoringH Scoring;
    typedef SpdpProblemH Problem;
    const SpdpScoringH* sc = nullptr;
    int a_pad = 0;
    const int16_t *sigS = nullptr, *sigT = nullptr, *sigE = nullptr;
    int lv_left = 0, lv_right = 0;              // where dinucleotide classes exist
    uint8_t mid[32], tron_of[64];               // the standard genetic code in the reference's tron alphabet
    int ss[2] = {0, 0};                         // the sites nearest_sites found
    enum { J_DIAGONAL, J_HEAD_NOGENOME, J_HEAD_CDS, J_HEAD_EXON, J_TAIL_NOGENOME, J_TAIL_CDS, J_TAIL_EXON, J_JUNCTION,
           J_MICRO_EXON, J_SHORTCUT, J_BACKFORTH, J_SMALL_DP, J_RECURSE, J_DP, J_GIVEUP_HEAD, J_GIVEUP_TAIL, J_GIVEUP_INNER,
           J_PICK_UNIT, J_EXACT_HEAD, J_EXACT_TAIL, J_COUNT, J_ABUT = J_COUNT, J_GIVEUP_LOCALC = J_COUNT };
    int joins[J_COUNT + 1] = {0};

    static bool joined(int s) { return s > SPDP_NEVSEL; }
    static bool creep_on(int d, int limit) { return std::abs(d) <= limit; }
    static bool small_gap(int dgap, int minl) { return dgap < minl; }
    static bool below_rec_limit(int ovr, int wlmt) { return ovr <= wlmt; }
    enum { SWITCH_QUERY_STEP = 3, SWITCH_KEEPS_NEGATIVE_N = 0, GIVEUP_COUNTS_SHORTCUT = 1, ABUT_ENDS_THE_JOIN = 0, HEAD_NEEDS_HSP = 0 };
    int rec_limit(unsigned level, int cmode) const { const int w = level <= 3 ? sp->wl_width[level] : 0; return (sp->crs == 0 && cmode != 3) ? 3 * w : w; }
    bool junction_gap(int agap) const { return agap <= 1; }
    bool dp_affordable(int agap, int, int cmode, unsigned level) const
    {
        return agap < ((sp->desert && cmode < 3) ? sp->desert * (4 - (int) level) : INT_MAX);
    }
    int end_margin() const { return (int) (sp->vthr / sc->gep); }
    int shortcut_cut_end(int from, int interval) const { return from + (interval > 0 ? interval / 3 * 3 : 0); }
    int shortcut_shoulder_floor(int alen, int margin) const { return alen - margin / 3; }
    void shortcut_flags(uint8_t, uint8_t) { cur.a_exgr = cur.b_exgr = 0; }                     // all four stay cleared: the callers put their own back
    void prepare_top_hsps()                     // the coding potential along every HSP joins its score
    {
        for (size_t k = 0; k + 1 < top_hsps.size(); ++k) {
            int s = 0;
            for (int i = 0, n = top_hsps[k].jy + 1; i < top_hsps[k].jlen; ++i, n += 3) s += sigE[n];
            top_hsps[k].jscr += s;
        }
    }

    enum { DEAD = 0, DIAG = 2, NEWD = 3, VERT = 4, SLA1 = 5, SLA2 = 6, HORI = 8, HOR1 = 9, HOR2 = 10 };      // traceback codes
    static bool is_diag(int d) { d &= 15; return d == DIAG || d == NEWD; }
    static bool is_vert(int d) { d &= 15; return (d >= 4 && d <= 7) || d == 12; }
    int sim(int i, int n) const { return sc->mtx[a[i] * sc->mtx_cols + b[n]]; }
    int simc(int i, int tron) const { return sc->mtx[a[i] * sc->mtx_cols + tron]; }
    int row_score(int m, int n) const { return sim(m, n + 1) + sigE[n + 1]; }
    int switch_score(int m, int n) const { return sim(m, n); }                  // quirk: the forward half of the diagonal switch reads column n, without the potential
    int gap_penalty(int i) const { return i == 0 ? 0 : (i > sc->codonk1 ? sc->lgop + i * sc->lgep : sc->gop + i * sc->gep); }
    int gap_penalty3(int i) const
    {
        if (i == 0) return 0;
        const int x = i % 3 == 1 ? sc->gape1 : (i % 3 == 2 ? sc->gape2 : 0);
        return x + (i > sc->codonk1 ? sc->lgop + i / 3 * sc->lgep : sc->gop + i / 3 * sc->gep);
    }
    int diagonal_shift_cost(int dr) const { return gap_penalty3(dr); }
    int int_pen(int len) const { return len < 0 ? SHRT_MIN : sc->intpen[std::min(len, sc->intpen_len - 1)]; }
    int lvl5(int n) const { return (n >= lv_left - 1 && n < lv_right - 1 && n >= 0 && n <= b_len) ? f5[dinc[n] >> 4] : 0; }
    int lvl3(int n) const { return (n >= lv_left + 1 && n <= lv_right && n >= 0 && n <= b_len) ? f3[dinc[n] & 15] : 0; }
    int is_canon(int d, int ac) const { return canon_rank(lvl5(d), lvl3(ac)); }
    int t53(int m, int n) const { return sc->t53[16 * (dinc[m] >> 4) + (dinc[n] & 15)]; }
    int pair_signal(int m, int n) const { return sig5[m] + sig3[n] + t53(m, n); }
    int junction_score(int n5, int n3) const { return int_pen(n3 - n5) + sig3[n3] + t53(n5, n3); }
    static bool same_residue(int x, int y) { return x == y || (x == 18 && y == 23); }           // SER / SER2
    static int codons_of(int d) { return (d >= 0 ? d + 1 : d - 1) / 3; }

    SpdpWindow band(int sh, int cmode = 0) const
    {
        SpdpWindow w;
        if (sh < 0) sh = -sh * std::min(cur.ar - cur.al, cur.br - cur.bl) / 100;
        w.up = cur.br - 3 * cur.ar;
        w.lw = cur.bl - 3 * cur.al;
        if (cmode == 1) w.lw = w.up; else if (cmode == 2) w.up = w.lw; else if (w.up < w.lw) std::swap(w.up, w.lw);
        w.up = std::min(w.up + 3 * sh, cur.br - 3 * cur.al);
        w.lw = std::max(w.lw - 3 * sh, cur.bl - 3 * cur.ar);
        w.width = w.up - w.lw + 7;
        return w;
    }
    int scalar_shoulder() const { return sc->sh; }

    // the two codons an intron between n5 and n3 can split, as tron codes {phase 1, phase 2}; a codon is defined when its own
    // three bases are (src/codepot.cc:84-106)
    bool split_codon(int n5, int n3, int cs[2])
    {
        cs[0] = cs[1] = 2;
        if (n5 < cur.bl || n3 >= cur.br) return true;
        int w[4];
        const int at[4] = {n5 - 2, n5 - 1, n3, n3 + 1};
        for (int k = 0; k < 4; ++k) {
            if (at[k] < 0 || at[k] > b_len) { mark(__LINE__); return false; }
            w[k] = mid[b[at[k]] & 31];
        }
        if (n3 == 0) w[2] = w[3] = 3;                            // quirk: PHE PHE stands in for position 0
        if (w[1] > 3 || w[2] > 3) return true;
        if (w[0] <= 3) cs[0] = tron_of[16 * w[0] + 4 * w[1] + w[2]];
        if (w[3] <= 3) cs[1] = tron_of[16 * w[1] + 4 * w[2] + w[3]];
        return true;
    }

    // ---- the reading frame runs on beyond an HSP: backwards to the start codon / forwards to the stop -----------------------
    // wmode 0: score only; 1: records when something was gained (and the anchor when the query starts here); 2: the anchor always
    int cds_end5(const SpdpSkl& at, int wmode)
    {
        int best = 0, scr = 0, x = at.m;
        SpdpSkl k = at;
        if (wmode && x == 0) rec.push_back(at);
        for (int y = at.n; y > cur.bl; y -= 3) {
            const int start = sigS[y + 1];
            if (start > 0) scr += start;
            if (scr > best) { best = scr; k.m = x; k.n = y; }
            if (start > 0 || scr + sp->vthr < 0) break;
            scr += sigE[y - 2];
            if (x > 0) { --x; scr += sim(x, y - 2); } else scr += sc->gep;
        }
        if (wmode && best > 0) { rec.push_back(at); rec.push_back(k); }
        else if (wmode == 2) rec.push_back(at);
        return best;
    }
    int cds_end3(const SpdpSkl& at, int best, int wmode)
    {
        int scr = best, x = at.m, rows = 0;
        SpdpSkl k = at;
        if (wmode) rec.push_back(at);
        for (int y = at.n; y < cur.br; y += 3) {
            const int stop = sigT[y + 1];
            scr += stop > 0 ? stop : sigE[y + 1] + sc->gep;
            if (scr > best) { best = scr; k.m = x; k.n = y + 3; }
            if (stop > 0 || scr + sp->vthr < 0) break;
            if (x < a_len) { ++x; scr += sim(at.m + rows, at.n + 1 + 3 * rows); ++rows; }
        }
        if (wmode && k.n != at.n) { rec.push_back(at); rec.push_back(k); }
        else if (wmode == 2) rec.push_back(at);
        return best;
    }

    // ---- the junction inside an overlap, nucleotide by nucleotide: a junction may split a codon -----------------------------
    bool junction(int agap, int& iscr, bool write)
    {
        const int dgap = cur.br - cur.bl - 3 * agap;
        const int first = cur.bl + 3 * agap - 2, lastn = cur.bl + 2, acc0 = cur.br - 2;
        const int passes = (sp->crs || agap) ? 1 : 2;
        const int overlap = 2 - agap;
        const int reach = std::min(std::min(cur.al, cur.bl / 3), overlap + 16);
        std::vector<int> back(std::max(reach, 0) + 2, 0);
        int depth = 0;
        for (int sum = 0; ; ) {
            if (!(++depth < reach)) break;
            const int gb = cur.bl + 1 - 3 * depth, gd = gb + dgap;
            if (gb < 0 || gd < 0 || gd > b_len) { mark(__LINE__); return false; }
            if (!(b[gb] == b[gd] || depth < overlap)) break;
            back[depth] = sum += sim(cur.al - depth, gb) + sigE[gb];
        }
        if (depth > 0) std::reverse(back.begin(), back.begin() + depth);
        SpdpSkl best = {0, 0};
        int best_phase = 0;
        bool residue_ok = true;
        iscr = NEV();
        for (int pass = 0; pass < passes && iscr == NEV(); ++pass) {
            // n runs over nucleotide positions; phs 1, -1, 0 = the junction sits after the first / second / third base of a codon
            int phs = 1, m = cur.ar - 1, qa = m, qb = acc0, t = 0, fwd = 0, walk = acc0;
            for (int n = first; n <= lastn; ++n, ++walk) {
                if (n < 0) continue;
                if (pass || is_canon(n, n + dgap)) {
                    bool ok = true;
                    int y = pair_signal(n, n + dgap) - fwd - back[t] + (cip ? cip[3 * m + phs] : 0);
                    if (phs) {
                        int cs[2];
                        if (!split_codon(n, n + dgap, cs)) return false;
                        const int c = cs[phs == 1 ? 1 : 0];
                        if (qa < 0 || qa >= a_len) { mark(__LINE__); return false; }
                        ok = sp->crs || same_residue(a[qa], c);
                        y = ok ? y + simc(qa, c) : NEV();
                    }
                    if (y > iscr) { best.n = n - phs; best.m = m; iscr = y; best_phase = -phs; residue_ok = ok; }
                }
                if (++phs == 0) ++t;
                else if (phs == 2) { ++m; phs = -1; }
                else { ++qa; qb += 3; fwd += sim(qa, qb) + sigE[walk + 1]; }
            }
        }
        if (!residue_ok || iscr <= NEV()) return false;
        if (write) {
            rec.push_back(best);
            phs5.set(best.n, (int8_t) best_phase);
            best.n += dgap;
            rec.push_back(best);
            phs3.set(best.n, (int8_t) best_phase);
            iscr += int_pen(dgap);
        }
        return true;
    }

    // ---- the (up to two) splice sites nearest to an open end of the genomic span, into ss[]; returns how many ----------------
    template <int SIDE>
    int nearest_sites(const Bound& bab)
    {
        const int from = SIDE == 5 ? cur.bl : cur.br, a0 = SIDE == 5 ? cur.al : cur.ar;
        auto sig = [&](int n) { return SIDE == 5 ? sig5[n] : sig3[n]; };
        auto strong = [&](int n, bool lax) { return SIDE == 5 ? (sig5[n] > sp->gc_sig5 || (lax && phs5[n] == 0)) : (sig3[n] > 0 || (lax && phs3[n] == 0)); };
        int found[2], nss = 0;
        for (int attempt = 0; ; ++attempt) {
            const bool lax = attempt > 0;
            nss = 0;
            // upstream, at most five residues of the query away (a residue = three positions without a site)
            for (int pa = a0, stop = std::max(bab.la, a0 - 5), n = from, plain = 0; pa > stop && n > bab.lb; --n) {
                if (strong(n, lax)) { if (nss < 2) found[nss++] = n; else break; }
                else if (++plain % 3 == 0) --pa;
            }
            const int up_dist = nss ? from - found[nss - 1] : 5;
            for (int pa = a0, stop = std::min(a0 + 5, bab.ua), n = from, plain = 1; pa < stop && ++n < bab.ub; ) {
                if (strong(n, lax)) {
                    if (nss < 2) found[nss++] = n;
                    else if (plain < up_dist || sig(n) > sig(found[nss - 1])) found[nss - 1] = n;
                    if (nss == 2) break;
                } else if (++plain % 3 == 0) ++pa;
            }
            if (nss || lax) break;
        }
        if (nss == 0) return 0;
        if (nss == 2) {
            const int d0 = std::abs(found[0] - from), d1 = std::abs(found[1] - from);
            if (d0 > d1 || (d0 == d1 && sig(found[0]) < sig(found[1]))) std::swap(found[0], found[1]);
            if (sig(found[0]) > sig(found[1])) --nss;
        }
        for (int n = 0; n < nss; ++n) ss[n] = found[n];
        return nss;
    }

    // score of the query rows [s0, ts) laid ungapped on the codons from column `col` on, plus the residues an intron splits
    // at either end (d5 / d3 = -1, 0, 1: where in the codon the junction sits); false: a read outside the sequences
    bool placed_rows(int s0, int ts, int col, int d5, int l5, int n5, int d3, int n3, int r3, int& ms)
    {
        ms = 0;
        int as = s0;
        if (d5) {
            int cs[2];
            if (!split_codon(l5, n5, cs)) return false;
            if (d5 == -1) col += 3;
            ms += simc(as++, cs[d5 == -1 ? 1 : 0]);
        }
        if (d3) {
            int cs[2];
            if (!split_codon(n3, r3, cs)) return false;
            ms += simc(ts, cs[d3 == -1 ? 1 : 0]);
        }
        for ( ; as < ts; col += 3) ms += sim(as++, col);
        return true;
    }

    int micro_exon(const Bound& bab)
    {
        if (!nearest_sites<5>(bab)) return NEV();
        const int l = ss[0];
        if (!nearest_sites<3>(bab)) return NEV();
        const int r = ss[0];
        const Span before = cur;
        int d5 = codons_of(cur.bl - l);
        cur.al -= d5; cur.bl -= 3 * d5;
        d5 = cur.bl - l;
        int d3 = codons_of(cur.br - r);
        cur.ar -= d3; cur.br -= 3 * d3;
        d3 = cur.br - r;
        const int alen = cur.ar - cur.al;
        if (alen <= 0) {
            int scr = 0;
            if (junction(alen, scr, true)) return scr;
            restore_ranges(before);
            return NEV();
        }
        const int s0 = cur.al - (d5 == 1), ts = cur.ar - (d3 == 1);
        const int cds = 3 * alen + d5 - d3;
        int best = NEV(), at = -1;
        for (int n5 = cur.bl + sp->minl, last = cur.br - cds - sp->minl; n5 < last; ++n5) {
            const int n3 = n5 + cds;
            if (phs3[n5] || phs5[n3]) continue;
            int ms;
            if (!placed_rows(s0, ts, n5 + d5 + 1, d5, l, n5, d3, n3, r, ms)) return NEV();
            const int scr = (int) (sp->w2 * ms + pair_signal(l, n5) + pair_signal(n3, r) + int_pen(n5 - l) + int_pen(r - n3));
            if (scr > best) { best = scr; at = n5; }
        }
        if (at < 0) { restore_ranges(before); return NEV(); }
        put(cur.al, cur.bl);
        put(cur.al, at + d5);
        put(cur.al + alen, at + cds + d3);
        put(cur.al + alen, cur.br);
        return best;
    }

    // ---- terminal exons.  Two searches share one frame: for each of the (up to two) candidate sites the query's end is laid
    // either ungapped at every canonical partner site (cross-species, or a single residue) or where an exact copy of it
    // occurs (same species); the frame keeps the best, restores the ranges for the next candidate, writes the records.
    struct EndPlacement { int pos = -1, score = SPDP_NEVSEL; bool perfect = false; };
    // ungapped placements of a head exon upstream of the acceptor `na` (d3: where in the codon the junction sits)
    EndPlacement head_ungapped(int d3, int nss)
    {
        EndPlacement out;
        const int na = cur.br - d3;
        const float second_site = nss > 1 ? sp->w2 - 1 : 0.f;
        int ts = cur.ar, self = 0, best = NEV();
        for (int as = cur.al; as < ts; ++as) self += simc(as, a[as] < sc->mtx_cols ? a[as] : 0);
        if (d3 == -1) self += simc(ts, a[ts]); else if (d3 == 1) --ts;
        for (int n = std::max(cur.bl, cur.br - 3 * cur.ar - sp->minl); n >= cur.bl; --n) {
            const int nd = n + 3 * cur.ar - d3;
            if (sigS[n + 1] <= 0 || !is_canon(nd, na)) continue;
            int ms;
            if (!placed_rows(cur.al, ts, n + 1, 0, 0, 0, d3, nd, na, ms)) { out.pos = -1; return out; }
            const int scr = (int) (sp->w2 * ms + sigS[n + 1] + sig5[nd] + junction_score(nd, na));
            if (scr > best) {
                best = scr; out.pos = n;
                out.score = (int) (best - second_site * ms);
                out.perfect = ms == self;
                if (out.perfect && na - nd > sp->ip_mode) break;
            }
            if (sp->ip_maxl && (na - nd) % sp->ip_maxl == 0 && best > NEV()) break;
        }
        return out;
    }
    EndPlacement tail_ungapped(int d5, int nss)
    {
        EndPlacement out;
        out.pos = INT_MIN / 2;
        const int l = cur.bl - d5, alen = cur.ar - cur.al, ts = cur.ar, s0 = cur.al - (d5 == 1);
        const float second_site = nss > 1 ? sp->w2 - 1 : 0.f;
        int self = 0, best = NEV();
        for (int as = s0; as < ts; ++as) self += simc(as, a[as]);
        for (int n = cur.bl + sp->minl, last = cur.br - 3 * alen - d5 - 1; n < last; ++n) {
            const int stop_at = n + 3 * alen + d5 + 1;
            if (sigT[stop_at] <= 0 || !is_canon(l, n)) continue;
            int ms;
            if (!placed_rows(s0, ts, n + d5 + 1, d5, l, n, 0, 0, 0, ms)) { out.pos = INT_MIN / 2; return out; }
            const int scr = (int) (sp->w2 * ms + sigT[stop_at] + sig5[l] + junction_score(l, n));
            if (scr > best) {
                best = scr; out.pos = n;
                out.score = (int) (best - second_site * ms);
                out.perfect = ms == self;
                if (out.perfect && n - l > sp->ip_mode) break;
            }
            if (sp->ip_maxl && (n - l) % sp->ip_maxl == 0 && best > NEV()) break;
        }
        out.pos += d5;
        return out;
    }

    int head_exon(const Bound& bab)
    {
        const int nss = nearest_sites<3>(bab);
        const auto run_on = [&]() { const SpdpSkl k = {cur.ar, cur.br}; return cds_end5(k, 2); };
        if (nss == 0) return run_on();
        const int sites[2] = {ss[0], ss[1]};
        const Span start = cur;
        Span at_first = cur;
        int best_pos = -1, best = NEV(), winner = 1;
        for (int n = 0; n < nss; ++n) {
            if (n) restore_ranges(start);
            const int r = sites[n];
            int d3 = codons_of(cur.br - r);
            cur.ar -= d3; cur.br -= 3 * d3;
            if (cur.ar == 0 || cur.br < 3) return run_on();
            if (cur.al >= cur.ar || cur.bl >= cur.br) continue;
            d3 = cur.br - r;
            if (sp->crs || cur.ar < 2) {
                if (n == 0) at_first = cur;
                const EndPlacement p = head_ungapped(d3, nss);
                if (unsupported) return NEV();
                if (p.score > best) { best = p.score; best_pos = p.pos; winner = n; if (p.perfect) break; }
                continue;
            }
            // same species: exact occurrences of the terminal stretch, nearest first, in windows of ip_maxl
            const int cds = 3 * cur.ar - d3;
            const int rows_end = cur.ar - (d3 == 1);
            if (rows_end - cur.al < 1) { mark(__LINE__); return NEV(); }
            ExactFinder<TronCode> bm(b, cur.bl, cur.br, a, cur.al, rows_end, -3);
            for (int from = std::max(cur.bl, cur.br - sp->ip_maxl); !bm.finished(); ) {
                int f = bm.next(from, -1) - 1;
                if (f >= 0) {
                    const int nd = f + cds;
                    if (nd < 0 || nd > b_len || f + 1 > b_len + 2) { mark(__LINE__); return NEV(); }
                    if (is_canon(nd, r)) {
                        if (d3) {
                            int cs[2];
                            if (!split_codon(nd, r, cs)) return NEV();
                            if (rows_end >= a_len + 1) { mark(__LINE__); return NEV(); }
                            if (!same_residue(rows_end < a_len ? a[rows_end] : a_pad, cs[d3 == -1 ? 1 : 0])) f = -1;
                        }
                        if (f >= 0) {
                            const int scr = sigS[f + 1] + sig5[nd] + junction_score(nd, r);
                            if (scr > best) { best = scr; best_pos = f; }
                        }
                    }
                }
                if (bm.scanned(from)) { if (best_pos < 0) from = std::max(cur.bl, from - sp->ip_maxl); else break; }
            }
            if (best_pos >= 0) { ++joins[J_EXACT_HEAD]; winner = 1; break; }
        }
        if (best_pos < 0) { restore_ranges(start); return NEV(); }
        if (winner == 0) restore_ranges(at_first);
        cur.bl = best_pos;
        put(cur.al, cur.bl); put(cur.ar, cur.bl + 3 * cur.ar); put(cur.ar, cur.br);
        return best;
    }

    int tail_exon(Bound& bab, const SpdpSkl& at)
    {
        if (cur.ar == a_len) ++bab.ua;
        const int nss = nearest_sites<5>(bab);
        if (nss == 0) return cds_end3(at, 0, 2);
        const int sites[2] = {ss[0], ss[1]};
        const Span start = cur;
        Span at_first = cur;
        int best_pos = -1, best = NEV(), alen = 0, alen_first = 0, winner = 1;
        for (int n = 0; n < nss; ++n) {
            if (n) restore_ranges(start);
            const int l = sites[n];
            int d5 = codons_of(cur.bl - l);
            cur.al -= d5; cur.bl -= 3 * d5;
            d5 = cur.bl - l;
            alen = cur.ar - cur.al;
            if (alen <= 0 && (alen < 0 || d5 <= 0)) {
                // nothing of the query is left behind the site: fine if the codon the site splits can still be the one before
                // it (alen < 0: a tryptophan codon whose last base starts the intron; d5 = -1: a codon with T in the middle)
                static const char ncodon[] = "--NCGAAGAAGATTATTCCCGATGGA";
                bool fits = alen == 0 && d5 == 0;
                if (alen < 0) fits = b[start.bl + 1] == 20;
                else if (d5 == -1) fits = ncodon[b[start.bl] < 26 ? b[start.bl] : 0] == 'T';
                if (!fits) return cds_end3(at, 0, 2);
                rec.push_back(at);
                return 0;
            }
            if (sp->crs || alen < 2) {
                if (n == 0) { at_first = cur; alen_first = alen; }
                const EndPlacement p = tail_ungapped(d5, nss);
                if (unsupported) return NEV();
                if (p.score > best) { best = p.score; best_pos = p.pos; winner = n; if (p.perfect) break; }
                continue;
            }
            int row0 = cur.al;                                   // the row whose codon the junction splits
            if (d5 < 0) { ++cur.al; --alen; d5 += 3; }
            if (cur.ar - cur.al < 1) { mark(__LINE__); return NEV(); }
            ExactFinder<TronCode> bm(b, cur.bl, cur.br, a, cur.al, cur.ar, 3);
            if (d5 == 1) --row0;
            for (int upto = std::min(cur.br, l + sp->ip_maxl); !bm.finished(); ) {
                int f = bm.next(-1, upto) - 1;
                if (f >= 0) {
                    const int na = f - d5;
                    if (is_canon(l, na)) {
                        if (d5) {
                            int cs[2];
                            if (!split_codon(l, na, cs)) return NEV();
                            if (row0 < 0) { mark(__LINE__); return NEV(); }
                            if (!same_residue(a[row0], cs[d5 != 1 ? 1 : 0])) f = -1;
                        }
                        if (f >= 0) {
                            const int t = f + 3 * alen + 1;
                            if (t < 0 || t > b_len + 2) { mark(__LINE__); return NEV(); }
                            if (sigT[t] > 0) { best = sig5[l] + junction_score(l, na); best_pos = f; break; }
                        }
                    }
                }
                if (bm.scanned(upto)) { if (best_pos < 0) upto = std::min(cur.br, upto + sp->ip_maxl); else break; }
            }
            if (d5 == 2) { --cur.al; ++alen; best_pos -= 3; }
            if (best_pos >= 0) { ++joins[J_EXACT_TAIL]; winner = 1; break; }
        }
        if (best_pos < 0) { restore_ranges(start); return NEV(); }
        if (winner == 0) { restore_ranges(at_first); alen = alen_first; }
        put(cur.al, cur.bl);
        put(cur.al, best_pos);
        const SpdpSkl k = {cur.ar, best_pos + 3 * alen};
        return cds_end3(k, best, 2);
    }

    template <class W>
    int head_join(W&, int, int bgap, bool, const Hsp* wjxt, int, const Bound& bab)
    {
        if (bgap < 0) { ++joins[J_HEAD_NOGENOME]; cur.al -= bgap / 3; cur.bl -= bgap; put(cur.al, cur.bl); return 0; }
        int s = NEV();
        if (wjxt && (sp->crs || cur.ar == cur.al)) { ++joins[J_HEAD_CDS]; const SpdpSkl k = {wjxt->jx, wjxt->jy}; s = cds_end5(k, 1); }
        if (s <= 0) {
            Records before = rec;
            const int exon = head_exon(bab);
            if (exon > s) { ++joins[J_HEAD_EXON]; s = exon; } else rec.swap(before);
        }
        return s;
    }
    template <class W>
    int tail_join(W&, int, int bgap, int, Bound& bab)
    {
        if (bgap <= 0) { ++joins[J_TAIL_NOGENOME]; cur.al -= bgap / 3; cur.bl -= bgap; put(cur.al, cur.bl); return 0; }
        const SpdpSkl k = {cur.al, cur.bl};
        int s = NEV();
        if (sp->crs || cur.ar == cur.al) { ++joins[J_TAIL_CDS]; s = cds_end3(k, 0, 1); }
        if (s <= 0) {
            Records before = rec;
            const int exon = tail_exon(bab, k);
            if (exon > s) { ++joins[J_TAIL_EXON]; s = exon; } else rec.swap(before);
        }
        return s;
    }
    bool unit_ends(const Unit& u, int cmode, const Span& keep, int& jscore)
    {
        const Hsp& first = u.jxt[0];
        cur.al = keep.al; cur.bl = keep.bl; cur.ar = first.jx; cur.br = first.jy;
        int agap = first.jx - cur.al, s = NEV();
        if (!sp->crs && agap > (cmode == 1 ? 0 : 1)) return false;
        if (cmode == 1) { const SpdpSkl k = {first.jx, first.jy}; jscore = u.scr + cds_end5(k, 0); }
        else if (junction(agap, s, false)) jscore = u.scr + s;
        else return false;
        const Hsp& last = u.jxt[u.num - 1];
        cur.al = last.jx + last.jlen; cur.bl = last.jy + 3 * last.jlen; cur.ar = keep.ar; cur.br = keep.br;
        agap = u.jxt[u.num].jx - cur.al;
        if (!sp->crs && agap > (cmode == 2 ? 0 : 1)) return false;
        if (cmode == 2) {
            const SpdpSkl k = {cur.al, cur.bl};
            s = cds_end3(k, 0, 0);
            if (s <= 0) return false;
        } else if (!junction(agap, s, false)) return false;
        jscore += s;
        return true;
    }

    // ---- the intron-less X-drop extension of an open end in three frames -------------------------------------------------------
    // A cell takes the codon match, deletions of one / two / three nucleotides, or an insertion from the cells one / two /
    // three columns back (three rotating insertion states); the row ends where all three frames have dropped Vthr below the
    // best score seen, the column range follows the previous row's peaks.
    struct Cell { int val, ptr, dir; };
    template <bool TOWARDS5>
    int end_extension(int* last, const SpdpWindow& w, bool, Trail& vmf)
    {
        const int S = TOWARDS5 ? -1 : 1;
        const Cell black = {NEV(), 0, 0};
        const int width = w.width;
        if (width < 7) { mark(__LINE__); *last = 0; return NEV(); }
        std::vector<Cell> buf(2 * (size_t) width, black);
        auto H = [&](int r) -> Cell& { return buf[r - w.lw + 3]; };
        auto F = [&](int r) -> Cell& { return buf[width + r - w.lw + 3]; };
        auto inbuf = [&](int r) { const long i = (long) r - w.lw + 3; return i >= 0 && i < width; };
        const int m_corner = TOWARDS5 ? cur.ar : cur.al, n_corner = TOWARDS5 ? cur.br : cur.bl;
        int best_val = 0, best_m = m_corner, best_n = n_corner, best_p = 0, maxval = 0;
        vmf.add(0, 0, 0);
        {
            int r = n_corner - 3 * m_corner;
            H(r).val = 0; H(r).dir = DIAG; H(r).ptr = vmf.add(m_corner, n_corner, 0);
            const int rr = TOWARDS5 ? std::min(w.up, cur.br - 3 * cur.al) : std::max(w.lw, cur.bl - 3 * cur.ar);
            for (int i = 1; TOWARDS5 ? ++r <= rr : --r >= rr; ++i) {
                if (!inbuf(r)) { mark(__LINE__); return NEV(); }
                if (i <= 3) {
                    H(r) = H(r + S * i);
                    H(r).val += gap_penalty(i) + (i < 3 ? sc->extragop : 0);
                    H(r).dir = VERT;
                } else {
                    H(r) = H(r + S * 3);
                    H(r).val += i > sc->codonk1 ? sc->lgep : sc->gep;
                }
            }
        }
        int m = m_corner;
        if (TOWARDS5 ? !cur.a_exgr : !cur.a_exgl) m -= S;
        int n1, n2;
        if (TOWARDS5) { n1 = 3 * m + w.lw; n2 = 3 * m + w.up + 1; }
        else { n1 = 3 * m + w.lw - 1; n2 = 3 * m + w.up; best_val = maxval = NEV(); }
        enum { R_NONE, R_H, R_F, R_E };                          // where the best diagonal cell of the current block lives
        for (;;) {
            m += S;
            if (TOWARDS5 ? m < cur.al : m > cur.ar) break;
            n1 += 3 * S; n2 += 3 * S;
            const int n0 = TOWARDS5 ? std::min(n2, cur.br) : std::max(n1, cur.bl);
            const int n9 = TOWARDS5 ? std::max(n1, cur.bl) : std::min(n2, cur.br);
            int n = n0, r = n - 3 * m, count3 = 0;
            Cell e1[3] = {black, black, black};
            if (!inbuf(r)) { mark(__LINE__); return NEV(); }
            if ((TOWARDS5 ? !cur.b_exgr : !cur.b_exgl) && n == n_corner && m == m_corner) { e1[2] = H(r); e1[2].val = sc->gapw3; }
            int nr[3];
            for (int p = 0; p < 3; ++p) nr[((n + S * -p) % 3 + 3) % 3] = n - S * p;
            int ref_kind = (H(r).val + sp->vthr < maxval) ? R_NONE : R_H, ref_idx = r;
            nr[(n % 3 + 3) % 3] = n - 3 * S;
            auto ref_cell = [&]() -> const Cell& { return ref_kind == R_H ? H(ref_idx) : ref_kind == R_F ? F(ref_idx) : ref_kind == R_E ? e1[ref_idx] : black; };
            bool peak = false;
            int q = 0;
            const bool corner_row = m == m_corner;
            const int am = corner_row ? 0 : a[TOWARDS5 ? m : m - 1];
            for (;;) {
                n += S;
                if (TOWARDS5 ? n < n9 : n > n9) break;
                r += S;
                if (!inbuf(r) || !inbuf(r + 3 * S) || !inbuf(r - 3 * S)) { mark(__LINE__); return NEV(); }
                Cell& h = H(r);
                Cell& f = F(r);
                Cell& eq = e1[q];
                const int se = TOWARDS5 ? sigE[n + 1] : sigE[n - 2];
                int mx = 0;                                      // 0: h, 1: f, 2: eq
                if (!corner_row) {
                    if (TOWARDS5 ? n > cur.br - 3 : n < cur.bl + 3) h = black;
                    else {
                        const bool was_diag = is_diag(h.dir);
                        h.val += sc->mtx[am * sc->mtx_cols + b[TOWARDS5 ? n + 1 : n - 2]] + se;
                        h.dir = was_diag ? DIAG : NEWD;
                    }
                    const int y = F(r + 3 * S).val + sc->gep;
                    {   const Cell& fr = H(r + S);               // one nucleotide deleted
                        const int x = fr.val + (is_vert(fr.dir) ? sc->gape1 : sc->gapw1);
                        if (x > y) { f = fr; f.val = x; f.dir = SLA2; } else f.val = y; }
                    {   const Cell& fr = H(r + 2 * S);           // two
                        const int x = fr.val + (is_vert(fr.dir) ? sc->gape2 : sc->gapw2);
                        if (x > f.val) { f = fr; f.val = x; f.dir = SLA1; } }
                    {   const Cell& fr = H(r + 3 * S);           // a codon
                        const int x = fr.val + sc->gapw3;
                        if (x >= f.val) { f = fr; f.val = x; f.dir = VERT; }
                        else if (y >= f.val) { f = F(r + 3 * S); f.val = y; f.dir = VERT; } }
                    if (f.val >= h.val) mx = 1;
                }
                if (TOWARDS5 ? n < n0 - 2 : n > n0 + 2) {        // insertions: three, two, one nucleotide(s) back along the row
                    const Cell& fr = H(r - 3 * S);
                    const bool stop = !TOWARDS5 && m == cur.ar && sigT[n - 2] > 0;      // the stop codon ends the forward form
                    const int x = fr.val + (stop ? sigT[n - 2] : sc->gapw3);
                    const int y = eq.val += sc->gep;
                    if (x > y) {
                        eq = fr; eq.val = x;
                        if (TOWARDS5) { if (eq.dir) eq.dir = HORI; } else eq.dir = stop ? DEAD : HORI;
                    }
                    if (!stop) eq.val += se;
                }
                if (TOWARDS5 ? n < n0 - 1 : n > n0 + 1) {
                    const Cell& fr = H(r - 2 * S);
                    if (fr.val + sc->gapw2 > eq.val) { eq = fr; eq.val += sc->gapw2; eq.dir = HOR2; }
                }
                {
                    const Cell& fr = H(r - S);
                    if (fr.val + sc->gapw1 > eq.val) { eq = fr; eq.val += sc->gapw1; eq.dir = HOR1; }
                }
                if (eq.val >= (mx == 1 ? f.val : h.val)) mx = 2;
                const int qn = q;
                if (++q == 3) q = 0;
                Cell& best = mx == 0 ? h : (mx == 1 ? f : eq);
                if (best.dir == NEWD) best.ptr = vmf.add(m - S, n - 3 * S, best.ptr);
                int x = best.val;
                if (TOWARDS5) {
                    if (x > maxval) maxval = x;
                    if (m == cur.al && sigS[n + 1] > 0) x += sigS[n + 1];
                    if (x > best_val) { best_val = x; best_m = m; best_n = n; best_p = best.ptr; }
                } else {
                    if (x > best_val) { best_val = x; best_m = m; best_n = n; best_p = best.ptr; }
                    if (best.val > maxval) maxval = best.val;
                }
                const int f3 = (n % 3 + 3) % 3;
                if (best.val + sp->vthr < maxval) {
                    if (++count3 == 3 && peak) { if (TOWARDS5) n1 = n + 3; else n2 = n - 3; peak = false; }
                    nr[f3] = n;
                } else {
                    if (is_diag(best.dir) && best.val >= ref_cell().val) {
                        ref_kind = mx == 0 ? R_H : (mx == 1 ? R_F : R_E);
                        ref_idx = mx == 2 ? qn : r;
                        if (TOWARDS5) { if (nr[f3] < n2) n2 = nr[f3]; } else { if (nr[f3] > n1) n1 = nr[f3]; }
                        peak = true;
                    }
                    count3 = 0;
                }
                if (mx != 0) h = best;
            }
            if (TOWARDS5) { if (!ref_cell().dir) break; if (peak) n1 = n + 3; }
            else { if (peak) n2 = n - 3; if (!ref_cell().dir) break; }
        }
        *last = vmf.add(best_m, best_n, best_p);
        if (!TOWARDS5) is3end = true;
        return best_val;
    }
};

// ============================================================================================================================
// the walk
// ============================================================================================================================
template <class Path>
class Walk : public Path {
    typedef Path P;
    enum { K = P::STEP };
public:
    using P::cur; using P::rec; using P::sp; using P::dp; using P::joins; using P::put; using P::NEV;
    // ---- geometry shared by both paths, over P::row_score -----------------------------------------------------------------
    // the gap is a straight diagonal: sum it up (local ends may trim it)
    int diagonal()
    {
        const bool trim_l = P::Local() && cur.a_exgl && cur.b_exgl, trim_r = P::Local() && cur.a_exgr && cur.b_exgr;
        int scr = 0, best = NEV(), from = cur.al, to = cur.ar;
        for (int m = cur.al; m < cur.ar; ++m) {
            scr += P::row_score(m, cur.bl + K * (m - cur.al));
            if (trim_l && scr < 0) { scr = 0; from = m + 1; }
            if (trim_r && scr > best) { best = scr; to = m + 1; }
        }
        put(from, cur.bl + K * (from - cur.al));
        put(to, cur.bl + K * (to - cur.al));
        return trim_r ? best : scr;
    }
    // extend the left / right HSP along its diagonal into the gap while that does not cost more than `limit`
    int creep(bool back, int& ovr, int limit, const Bound& lub)
    {
        int d = 0;
        for (;;) {
            const bool room = back ? (cur.al > lub.la && cur.bl > lub.lb) : (cur.ar < lub.ua && cur.br < lub.ub);
            if (!room || !(ovr < 0 || P::creep_on(d, limit))) break;
            if (back) { d += P::row_score(cur.al - 1, cur.bl - K); --cur.al; cur.bl -= K; }
            else { d += P::row_score(cur.ar, cur.br); ++cur.ar; cur.br += K; }
            if ((ovr += K) == 0) limit += d;
        }
        return d;
    }
    // both ways; the forward creep advances the caller's overlap count, the backward one works on a copy of it
    int creep_both(int& ovr, const Bound& lub)
    {
        int back_ovr = ovr;
        const int d = creep(true, back_ovr, P::slmt(), lub);
        return d + creep(false, ovr, P::slmt(), lub);
    }
    // two HSPs overlap on both sequences without room for an intron: where along the overlap to change diagonals
    int switch_diagonals(int ovr, const Bound& lub)
    {
        const int cells = ovr / K;
        std::vector<int> acc(cells + 1, 0);
        int i = cells, scr = 0;
        for (int j = 1; ; ++j) {                                  // left HSP's diagonal, backwards from the left ends
            if (--i < 0) break;
            if (cur.al - P::SWITCH_QUERY_STEP * j < lub.la) break;     // quirk (protein): the query bound moves by three per residue
            if (cur.bl - K * j < lub.lb) break;
            acc[i] = scr += P::row_score(cur.al - j, cur.bl - K * j);
        }
        int best = scr, where = ++i;
        scr = 0;
        for (int t = where; ; ++t) {                              // right HSP's diagonal, forwards from the right ends
            if (!(i++ < cells)) break;
            if (!(cur.ar + t < lub.ua)) break;
            if (!(cur.br + K * t < lub.ub)) break;
            scr += P::switch_score(cur.ar + t, cur.br + K * t);
            if ((acc[i] += scr) > best) { best = acc[i]; where = i; }
        }
        SpdpSkl k = {cur.ar + where, cur.br + K * where};
        rec.push_back(k);
        int shift = (cur.br - K * cur.ar) - (cur.bl - K * cur.al);
        if (shift >= 0) k.n -= shift; else k.m -= (shift = -shift) / K;
        if (P::SWITCH_KEEPS_NEGATIVE_N || k.n >= 0) rec.push_back(k);
        return best + P::diagonal_shift_cost(shift);
    }

    // the X-drop extension of an open end, its records appended start to end
    int open_end(int cmode, bool lcl = true)
    {
        Trail vmf;
        int ptr = 0;
        if (cmode == 3) {
            if (cur.al > P::a_len - cur.ar) { cmode = 2; cur.ar = P::a_len; }
            else { cmode = 1; cur.al = 0; rec.clear(); }
        }
        const SpdpWindow w = P::band(P::scalar_shoulder(), cmode);
        const int scr = cmode == 1 ? P::template end_extension<true>(&ptr, w, lcl, vmf) : P::template end_extension<false>(&ptr, w, lcl, vmf);
        for (int p = ptr; p; p = vmf.prev[p]) put(vmf.m[p], vmf.n[p]);
        return scr;
    }

    // one traceback sweep over both flanks of a gap the HSPs leave open, the genomic middle jumped over as one insertion
    int shortcut(int ovr, const Bound& bab)
    {
        const int margin = sp->minl;
        const int interval = cur.br - cur.bl - 2 * margin;
        const int cut[2] = {cur.bl + margin, P::shortcut_cut_end(cur.bl + margin, interval)};
        const bool jump = K == 1 ? interval > 0 : cut[1] > cut[0];
        ovr = (ovr > 0 ? 0 : ovr) - 3;
        int scr = -creep_both(ovr, bab);
        const int alen = cur.ar - cur.al;
        int sh = alen / 2;
        const int given = P::scalar_shoulder();
        if (given < 0) {
            float f = (float) -given;
            if (f > 1.f) f /= 100;
            if (f < 0.5f) sh = (int) (alen * f);
        } else if (given < sh) sh = given;
        sh = std::max(sh, P::shortcut_shoulder_floor(alen, margin));
        const SpdpWindow w = P::band(sh);
        const uint8_t aexg = cur.a_exgl, bexg = cur.b_exgl;
        cur.a_exgl = cur.b_exgl = 0;
        if (K == 3) cur.a_exgr = cur.b_exgr = 0;
        scr += dp->trcbk(cur, w, true, jump ? cut : nullptr, rec);
        P::shortcut_flags(aexg, bexg);
        return scr;
    }

    // ---- the gap filler ---------------------------------------------------------------------------------------------------------
    // One gap between two HSPs (or an HSP and an end of the query).  cmode 1 / 2 / 3: the gap is the 5' end, the 3' end,
    // internal.  The rows of `fill_table` are tried in order; `when` sees the gap's geometry, `how` fills it and returns the
    // score (NEV: could not).  After the table: the full DP where it is affordable, then giving up into an end extension.
    struct Gap {
        unsigned level; int cmode; const Hsp* wjxt; Bound bab;
        int agap, bgap, ovr, dgap; bool cont, no_rec;
        int scr = 0;                                            // what the filler spent besides the join's own score
        Records saved; bool have_saved = false;
    };
    struct Row { bool (Walk::*when)(const Gap&); int (Walk::*how)(Gap&); };

    bool when_straight(const Gap& g) { return g.dgap == 0 && (P::ABUT_ENDS_THE_JOIN || g.agap); }
    int how_straight(Gap& g)
    {
        if (g.agap == 0) { ++joins[P::J_ABUT]; if (g.wjxt) put(g.wjxt->jx, g.wjxt->jy); return 0; }
        ++joins[P::J_DIAGONAL];
        return diagonal();
    }
    bool when_head(const Gap& g) { return g.cmode == 1 && g.no_rec && (!P::HEAD_NEEDS_HSP || g.wjxt); }
    int how_head(Gap& g) { return P::head_join(*this, g.agap, g.bgap, g.cont, g.wjxt, g.cmode, g.bab); }
    bool when_tail(const Gap& g) { return g.cmode == 2 && g.no_rec; }
    int how_tail(Gap& g) { return P::tail_join(*this, g.agap, g.bgap, g.cmode, g.bab); }
    // (a row whose `when` has a side effect: the junction search IS the test, and it writes the junction when it succeeds)
    int junction_score_ = 0;
    bool when_junction(const Gap& g)
    {
        return g.cmode == 3 && P::junction_gap(g.agap) && g.dgap >= sp->minl && P::junction(g.agap, junction_score_, true);
    }
    int how_junction(Gap&) { ++joins[P::J_JUNCTION]; return junction_score_; }
    bool when_lost_exon(const Gap& g) { return g.cmode == 3 && g.no_rec && g.dgap >= sp->minl; }
    int how_lost_exon(Gap& g)
    {
        int s = NEV();
        if (sp->crs == 0) { s = P::micro_exon(g.bab); if (s != NEV()) ++joins[P::J_MICRO_EXON]; }
        if (s == NEV() && g.agap < sp->elmt) { ++joins[P::J_SHORTCUT]; s = shortcut(g.ovr, g.bab); }
        return s;
    }
    bool when_overlap(const Gap& g) { return g.ovr <= 0 && g.dgap < sp->minl; }
    int how_overlap(Gap& g) { ++joins[P::J_BACKFORTH]; return switch_diagonals(-g.ovr, g.bab); }
    bool when_small(const Gap& g) { return P::small_gap(g.dgap, sp->minl); }
    int how_small(Gap& g)
    {
        ++joins[P::J_SMALL_DP];
        g.scr -= creep_both(g.ovr, g.bab);
        const bool crossed = K == 3 && cur.bl > cur.br;         // (the protein path's creeps can cross the genomic ends)
        if (crossed) std::swap(cur.bl, cur.br);
        const SpdpWindow w = P::band(std::min(P::scalar_shoulder(), std::abs(g.dgap) + 3));
        const int s = dp->trcbk(cur, w, K == 1, nullptr, rec);
        if (crossed) std::swap(cur.bl, cur.br);
        return s;
    }
    bool when_deeper(const Gap& g) { return (int) g.level < sp->qck; }
    int how_deeper(Gap& g)
    {
        ++joins[P::J_RECURSE];
        g.saved = rec; g.have_saved = true;
        return seeded(g.level, g.cmode, g.bab);
    }
    static const Row* fill_table(int& n)
    {
        static const Row rows[] = {
            {&Walk::when_straight, &Walk::how_straight}, {&Walk::when_head, &Walk::how_head}, {&Walk::when_tail, &Walk::how_tail},
            {&Walk::when_junction, &Walk::how_junction}, {&Walk::when_lost_exon, &Walk::how_lost_exon},
            {&Walk::when_overlap, &Walk::how_overlap}, {&Walk::when_small, &Walk::how_small}, {&Walk::when_deeper, &Walk::how_deeper}};
        n = (int) (sizeof rows / sizeof rows[0]);
        return rows;
    }

    int fill_gap(unsigned level, const int cmode, const Hsp* wjxt, const Bound& bab_in)
    {
        if (P::is3end) return 0;
        Gap g;
        g.cmode = cmode; g.wjxt = wjxt; g.bab = bab_in;
        g.agap = cur.ar - cur.al; g.bgap = cur.br - cur.bl;
        g.ovr = std::min(K * g.agap, g.bgap);
        g.dgap = g.bgap - K * g.agap;
        g.cont = g.agap <= 0;
        const int wlmt = K == 1 ? P::rec_limit(level + 1, cmode) : P::rec_limit(level, cmode);     // quirk: the cDNA path reads the limit of the level it is about to enter
        g.level = ++level;
        g.no_rec = P::below_rec_limit(g.ovr, wlmt);
        int iscore = NEV();
        int n_rows;
        const Row* rows = fill_table(n_rows);
        for (int i = 0; i < n_rows; ++i)
            if ((this->*rows[i].when)(g)) {
                iscore = (this->*rows[i].how)(g);
                if (i == 0 && P::ABUT_ENDS_THE_JOIN && g.agap == 0) return 0;
                if (i == 0 && !P::ABUT_ENDS_THE_JOIN) { g.scr += iscore; iscore = 0; }
                break;
            }
        if (P::unsupported) return NEV();
        if (iscore == NEV() && (g.no_rec || (int) level == sp->qck) && P::dp_affordable(g.agap, g.bgap, cmode, level) &&
            !(P::LocalC() && sp->qck == 3 && cmode < 3)) {
            const Span before = cur;
            if (cmode & 1) g.scr -= creep(false, g.ovr, P::slmt(), g.bab);
            if (cmode & 2) { int o2 = g.ovr; g.scr -= creep(true, o2, P::slmt(), g.bab); }
            g.agap += before.al - cur.al + cur.ar - before.ar;
            g.bgap += before.bl - cur.bl + cur.br - before.br;
            if (g.have_saved) rec = g.saved; else { g.saved = rec; g.have_saved = true; }
            ++joins[P::J_DP];
            iscore = dp->lsp(cur, P::band(P::scalar_shoulder()), rec);
        }
        if (iscore == NEV()) {
            if (g.have_saved) rec = g.saved;
            if (K == 1 && P::LocalC()) {
                ++joins[P::J_GIVEUP_LOCALC];
                put(cmode == 1 ? cur.ar : cur.al, cmode == 1 ? cur.br : cur.bl);
                iscore = 0;
            } else if (cmode == 1) {
                ++joins[P::J_GIVEUP_HEAD];
                if (wjxt) cur.bl = std::max(cur.bl, wjxt->jy + P::end_margin());
                iscore = open_end(cmode);
            } else if (cmode == 2) {
                ++joins[P::J_GIVEUP_TAIL];
                if (wjxt) { const int br = g.bgap - wjxt->jy - P::end_margin(); if (br > cur.bl && br < cur.br) cur.br = br; }
                iscore = open_end(cmode);
            } else {
                ++joins[P::J_GIVEUP_INNER];
                if (P::Local()) iscore = open_end(cmode);
                else { if (P::GIVEUP_COUNTS_SHORTCUT) ++joins[P::J_SHORTCUT]; iscore = shortcut(g.ovr, g.bab); }
            }
        }
        return g.scr + iscore;
    }

    // among several units of the HSP search: the one whose ends join their neighbours best
    int pick_unit(const std::vector<Unit>& units, int cmode)
    {
        const Span keep = cur;
        int best = NEV(), which = -1;
        for (size_t u = 0; u < units.size(); ++u) {
            int jscore = 0;
            if (P::unit_ends(units[u], cmode, keep, jscore) && jscore > best) { best = jscore; which = (int) u; }
            if (P::unsupported) break;
        }
        P::restore_ranges(keep);
        return best > NEV() ? which : -1;
    }

    // the HSPs of one level, left to right: every gap filled, the HSP scores added; eimode = where this stretch sits in
    // the query (1: reaches the 5' end, 2: the 3' end, 3: internal)
    int seeded(unsigned level, int eimode, const Bound& lub)
    {
        const Span at_entry = cur;
        int cmode = eimode, scr = 0, num = 0;
        std::vector<Unit> units;
        std::vector<Hsp>* list = nullptr;
        const int wlmt = level <= 3 ? sp->wl_width[level] : 0;
        Bound bab = lub;
        const bool given = (int) level == P::lowest_level && !P::top_hsps.empty();
        if (given) {
            list = &this->top_hsps;
            num = (int) list->size() - 1;
            P::prepare_top_hsps();
        } else {
            if (!dp->wilip((int) level, cur, units)) { P::mark(__LINE__); return NEV(); }
            int pick = units.empty() ? -1 : 0;
            if (units.size() > 1 && cur.br - cur.bl >= sp->minl) { ++joins[P::J_PICK_UNIT]; pick = pick_unit(units, cmode); }
            if (K == 3 && P::unsupported) return NEV();
            if (pick >= 0) { list = &units[pick].jxt; num = units[pick].num; }
            else if (units.size() > 1) level = sp->qck - 1;
        }
        const Hsp* wjxt = nullptr;
        if (num) {
            std::vector<Hsp>& jxt = *list;
            jxt[num].jx = cur.ar; jxt[num].jy = cur.br;          // the slot behind the last HSP: the right end of this stretch
            cur.a_exgr = cur.b_exgr = 0;
            for (int k = 0; k < num; ++k) {
                const Hsp& h = jxt[k];
                scr += h.jscr;
                cur.ar = h.jx; cur.br = h.jy;
                bab.ua = std::max(std::min(h.jx + h.jlen, jxt[k + 1].jx) - wlmt, h.jx + h.jlen / 2);
                bab.ub = h.jy + K * (bab.ua - h.jx);
                if (cmode == 2) cmode = 3;
                const int s = fill_gap(level, cmode, &h, bab);
                if (K == 3 && P::unsupported) return NEV();
                if (P::joined(s)) {
                    scr += s;
                    cmode = 3;
                    cur.al = h.jx + h.jlen; cur.bl = h.jy + K * h.jlen;
                    cur.a_exgl = cur.b_exgl = 0;
                    bab.la = cur.ar; bab.lb = cur.br;
                }
            }
            wjxt = &jxt[num];
            cur.a_exgr = at_entry.a_exgr; cur.b_exgr = at_entry.b_exgr;
            cur.ar = at_entry.ar; cur.br = at_entry.br;
            bab.ua = lub.ua; bab.ub = lub.ub;
            if (eimode == 2 || ((int) level == P::lowest_level && eimode == 1)) cmode = 2;
        }
        const int s = fill_gap(level, cmode, wjxt, bab);
        if (K == 3 && P::unsupported) return NEV();
        scr = s > NEV() ? scr + s : NEV();
        P::restore_ranges(at_entry);
        cur.a_exgl = at_entry.a_exgl; cur.b_exgl = at_entry.b_exgl;
        if ((int) level == P::lowest_level && wjxt && list == &this->top_hsps) { (*list)[num].jx = P::a_len; (*list)[num].jy = P::b_len; }
        return scr;
    }

    // the whole query: the record file starts with one dummy record; returns the raw score
    int run(const Span& whole)
    {
        cur = whole;
        rec.clear();
        rec.push_back({0, 0});
        P::is3end = false;
        const Bound bab = {cur.al, cur.bl, cur.ar, cur.br};
        return seeded((unsigned) P::lowest_level, 1, bab);
    }
};

typedef Walk<CdnaPath> SeedWalk;
typedef Walk<ProteinPath> SeedWalkH;

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

template <class W>
inline void bind_common(W& w, const SpdpSeedParams* sp, const SpdpJuxt* hsps, int n_hsps, int lowest_level)
{
    w.sp = sp; w.lowest_level = lowest_level;
    site_levels(sp, w.f5, w.f3);
    w.top_hsps.clear();
    if (hsps && n_hsps > 0)
        for (int j = 0; j <= n_hsps; ++j) w.top_hsps.push_back({hsps[j].jx, hsps[j].jy, hsps[j].jlen, hsps[j].nid, hsps[j].jscr});
}

// points a walk at one query's inputs; without phs5 / phs3 the marks are derived from the canonical-site levels as the
// reference's Exinon::intron53_n derives them (src/codepot.cc:504-518, algmode.any != 2)
inline bool bind_problem(SeedWalk& w, const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpProblem* p,
                         const SpdpJuxt* hsps, int n_hsps, int lowest_level)
{
    if (!sc || !sp || !p || !p->a || !p->b || !p->sig5 || !p->sig3 || !p->cano5 || !p->cano3 || !p->dinc ||
        !sc->intpen || sc->intpen_len <= 0 || sp->qck < 1 || sp->qck > 3) return false;
    w.a = p->a; w.a_len = p->a_len; w.b = p->b; w.b_len = p->b_len;
    w.sig5 = p->sig5; w.sig3 = p->sig3; w.cano5 = p->cano5; w.cano3 = p->cano3; w.dinc = p->dinc; w.cip = p->cip;
    w.sc = sc;
    const int N = p->b_len + 1;
    if (p->phs5 && p->phs3) { w.phs5.bind(p->phs5); w.phs3.bind(p->phs3); }
    else {
        w.phs5.derive(N); w.phs3.derive(N);
        for (int side = 0; side < 2; ++side) {
            std::vector<int8_t>& q = side ? w.phs3.own : w.phs5.own;
            const uint8_t* cano = side ? p->cano3 : p->cano5;
            const int n_end = std::min(N - 1, p->b_right + 1);
            for (int n = std::max(1, p->b_left); n < n_end; ++n) {
                if (n + 8 <= n_end) {                   // (most positions are no site: eight at a glance, then straight to the next one)
                    uint64_t eight;
                    memcpy(&eight, cano + n, 8);
                    if (!eight) { n += 7; continue; }
                    n += __builtin_ctzll(eight) >> 3;   // (little-endian: the lowest non-zero byte is the first site)
                }
                if (q[n] == -2 && cano[n]) {
                    q[n] = 0;
                    if (cano[n] > 1) { q[n + 1] = 1; q[n - 1] = q[n - 1] == 1 ? 2 : -1; }
                }
            }
        }
    }
    bind_common(w, sp, hsps, n_hsps, lowest_level);
    return true;
}

inline bool bind_problem_h(SeedWalkH& w, const SpdpScoringH* sc, const SpdpSeedParams* sp, const SpdpProblemH* p,
                           const SpdpJuxt* hsps, int n_hsps, int lowest_level)
{
    if (!sc || !sp || !p || !p->a || !p->b || !p->sig5 || !p->sig3 || !p->sigS || !p->sigT || !p->sigE || !p->phs5 || !p->phs3 ||
        !p->dinc || !sc->intpen || sc->intpen_len <= 0 || sp->qck < 1 || sp->qck > 3) return false;
    w.a = p->a; w.a_len = p->a_len; w.a_pad = p->a_pad; w.b = p->b; w.b_len = p->b_len;
    w.sig5 = p->sig5; w.sig3 = p->sig3; w.sigS = p->sigS; w.sigT = p->sigT; w.sigE = p->sigE; w.dinc = p->dinc; w.cip = p->cip;
    w.sc = sc;
    w.phs5.bind(p->phs5); w.phs3.bind(p->phs3);
    spdp_genetic_code_tables(w.mid, w.tron_of);
    w.lv_left = p->exin_left; w.lv_right = p->exin_right;
    bind_common(w, sp, hsps, n_hsps, lowest_level);
    return true;
}

}   // namespace spdp_seed
#endif
