// spdp_seeded_walk_h.h -- the seeded path of alignH_ng (protein query x three-frame genome): one query's walk, host side.
//
// What it mirrors (ogotoh/spaln v3.0.7, src/fwd2h1.cc):
//   Aln2h1::globalH_ng (algmode.qck != 0)      :3267-3286      SeedWalkH::run
//   Aln2h1::seededH_ng, addsigEjxt             :3180-3265, 2397  SeedWalkH::seeded
//   Aln2h1::bestwlu                            :3134-3178      SeedWalkH::best_unit
//   Aln2h1::interpolateH                       :3022-3132      SeedWalkH::interpolate
//   Aln2h1::indelfreespjH                      :2447-2520      SeedWalkH::indel_free_junction
//   Aln2h1::backforth                          :2293-2329      SeedWalkH::back_and_forth
//   Aln2h1::creepback / creepfwrd              :2411-2441      SeedWalkH::creep_back / creep_fwrd
//   Aln2h1::cds5end / cds3end                  :2331-2395      SeedWalkH::cds_end5 / cds_end3
//   Aln2h1::nearest5ss / nearest3ss            :2524-2616      SeedWalkH::nearest_sites<5 / 3>
//   Aln2h1::micro_exon                         :2620-2705      SeedWalkH::micro_exon
//   Aln2h1::first_exon(_wmm) / last_exon(_wmm) :2709-3020      SeedWalkH::first_exon / last_exon: ungapped placements
//                                                              (cross-species) or exact occurrences in three frames
//   BoyerMoore(b, a, +-3) + nexthit3           src/boyer_moore.cc      ExactFinder3
//   Aln2h1::openendH_ng + back2ward5endH_ng / for2ward3endH_ng  :2262-2291, 1522-1960   SeedWalkH::open_end, end_extension
//   Aln2h1::diagonalH_ng                       :1963-1995      SeedWalkH::diagonal
//   SpJunc::spjseq / spjscr                    src/codepot.cc:74-107   SeedWalkH::split_codon / junction_score
//   Aln2h1::shortcutH_ng                       :2232-2260      SeedWalkH::shortcut (forwardH_ng with a cut range behind
//                                                              DpBackendH::trcbk)
// Not served (the walk marks itself and the query comes back without an alignment): reads the reference would make
// outside its sequences.
//
// Same design as spdp_seeded_walk.h: the reference's decisions on the same mutable state, every DP call (lspH_ng,
// trcbkalignH_ng) through DpBackendH, header only, compiled into the product (device behind the calls) and into the CPU
// checker (oracle behind them).  A DP column n is a nucleotide position of the tron sequence, a row m an amino acid;
// the codon that ends at column n is b[n - 2].
#ifndef SPDP_SEEDED_WALK_H_H_
#define SPDP_SEEDED_WALK_H_H_

#include "spdp_seeded_walk.h"
#include "spdp_gencode.h"

namespace spdp_seed {

struct DpBackendH {
    virtual ~DpBackendH() {}
    virtual int lsp(const Span& s, const SpdpWindow& w, std::vector<SpdpSkl>& rec) = 0;         // Aln2h1::lspH_ng(wdw)
    // trcbkalignH_ng(wdw, spj, mc): cut = {left, right} of the genomic range the sweep jumps over, or null; spj false =
    // no introns -- only the scalar engine (-A0, fewer than 8 rows, or a cut) listens to it (:2004-2018)
    virtual int trcbk(const Span& s, const SpdpWindow& w, bool spj, const int* cut, std::vector<SpdpSkl>& rec) = 0;
    virtual bool wilip(int level, const Span& s, std::vector<Unit>& units) = 0;
};

// Exact occurrences of an amino-acid pattern in a tron text, all three frames, in the order and with the skips of the
// reference's search (BoyerMoore(b, a, +-3) + nexthit3, src/boyer_moore.cc:36-76, 114-153, 174-215): a text code
// matches a pattern residue when equal, when the text holds the AGY serine and the pattern SER, or when the pattern
// is AMB; the shift tables are built with "equal or either AMB".  Each call runs the three frames up to the bound
// and hands the hits out nearest first (the reference's priority queue of three).
class ExactFinder3 {
    const uint8_t* text; int tlen, origin;
    std::vector<uint8_t> pat; int plen;
    std::vector<int> by_code, by_suffix;
    int step, after_hit, idx[3];
    std::vector<int> queue;                                      // hits not handed out yet, next one last
    static bool src_eq(uint8_t t, uint8_t p) { return t == p || (t == 23 && p == 18) || p == 2; }
    static bool tab_eq(uint8_t x, uint8_t y) { return x == y || x == 2 || y == 2; }
    static int plus_k(int i, int k) { static const char t[3][3] = {{0, 1, 2}, {2, 0, 1}, {1, 2, 0}}; return i + t[i % 3][k]; }
    static int minus_k(int i, int k) { static const char t[3][3] = {{0, 2, 1}, {1, 0, 2}, {2, 1, 0}}; return i - t[i % 3][k]; }
public:
    ExactFinder3(const uint8_t* b, int bl, int br, const uint8_t* a, int al, int ar, int direction)
        : text(b + bl), tlen(br - bl), origin(bl), pat(a + al, a + ar), plen(ar - al), step(3 * direction)
    {
        pat.push_back(0);
        if (step < 0) std::reverse(pat.begin(), pat.begin() + plen);
        by_code.assign(256, plen);
        for (int j = 0, skip = plen; j < plen; ++j) by_code[pat[j]] = --skip;
        by_code[23] = by_code[18];                               // SER2 as SER
        by_suffix.resize(std::max(plen, 1));
        std::vector<int> link(std::max(plen, 1));
        for (int j = 0, v = 2 * plen; j < plen; ++j) by_suffix[j] = --v;
        int j = plen;
        for (int k = plen; --k >= 0; ) {
            link[k] = j;
            pat[plen] = pat[k];                                  // sentinel
            while (!tab_eq(pat[j], pat[k])) {
                by_suffix[j] = std::min(by_suffix[j], plen - 1 - k);
                j = link[j];
            }
            --j;
        }
        after_hit = std::max(j + 1, 2) * step;
        for (int s = j, v = plen, q = 0; q < plen; ++q) {
            by_suffix[q] = std::min(by_suffix[q], s + v--);
            if (q >= s) s = s >= 0 ? link[s] : 0;
        }
        if (step < 0) {
            std::reverse(pat.begin(), pat.begin() + plen);
            std::reverse(by_suffix.begin(), by_suffix.begin() + plen);
        }
        for (int k = 0; k < 3; ++k) idx[k] = step > 0 ? k : minus_k(tlen, k);
    }
    bool finished() const
    {
        return step > 0 ? std::min(idx[0], std::min(idx[1], idx[2])) >= tlen : std::max(idx[0], std::max(idx[1], idx[2])) <= 0;
    }
    bool scanned(int n) const
    {
        if (!queue.empty()) return false;
        n -= origin;
        return step > 0 ? std::min(idx[0], std::min(idx[1], idx[2])) >= n : std::max(idx[0], std::max(idx[1], idx[2])) <= n;
    }
    // nexthit3(l, r): position in b of the first code of the next occurrence, or -1
    int next(int l, int r)
    {
        if (!queue.empty()) { const int v = queue.back(); queue.pop_back(); return v + origin; }
        if (l >= 0) l = std::max(l - origin, 0);
        if (r >= 0) r = std::max(r - origin, 0);
        for (int k = 0; k < 3; ++k) {
            if (step > 0) {
                int i = l >= 0 ? plus_k(l, k) : idx[k];
                idx[k] = plus_k(r >= 0 ? r : tlen, k);
                for (i += step * (plen - 1); i < r; ) {
                    int j = plen - 1;
                    while (j >= 0 && src_eq(text[i], pat[j])) { i -= step; --j; }
                    if (j < 0) { queue.push_back(i + step); idx[k] = i + after_hit; break; }
                    i += step * std::max(by_code[text[i]], by_suffix[j]);
                }
            } else {
                int i = r >= 0 ? minus_k(r, k) : idx[k];
                idx[k] = minus_k(l >= 0 ? l : 0, k);
                for (i += step * (plen - 1); i >= l; ) {
                    int j = 0;
                    while (j < plen && src_eq(text[i], pat[j])) { i -= step; ++j; }
                    if (j >= plen) { queue.push_back(i + step * plen); idx[k] = i + after_hit; break; }
                    i += step * std::max(by_code[text[i]], by_suffix[j]);
                }
            }
        }
        if (queue.empty()) return -1;
        // the reference's heap hands out the smallest first going right, the largest first going left
        std::sort(queue.begin(), queue.end());
        if (step > 0) std::reverse(queue.begin(), queue.end());
        const int v = queue.back(); queue.pop_back();
        return v + origin;
    }
};

class SeedWalkH {
public:
    // inputs (borrowed)
    const uint8_t* a = nullptr; int a_len = 0; int a_pad = 0;
    const uint8_t* b = nullptr; int b_len = 0;
    const int16_t *sig5 = nullptr, *sig3 = nullptr, *sigS = nullptr, *sigT = nullptr, *sigE = nullptr;
    const uint8_t* dinc = nullptr;
    const int32_t* cip = nullptr;               // Cip_score::cip_score(c), c = 0 .. 3 a_len + 1, or null
    PhaseMarks phs5, phs3;                      // SGPT6::phs5 / phs3: the walk marks the junctions it accepts (:2511-2516)
    uint8_t f5[16] = {0}, f3[16] = {0};         // INT53::cano5 / cano3 levels by dinucleotide class; the classes exist for
    int lv_left = 0, lv_right = 0;              // [exin_left, exin_right): dinc5 of position i - 1 and dinc3 of i + 1 come from base i
    int lvl5(int n) const { return (n >= lv_left - 1 && n < lv_right - 1 && n >= 0 && n <= b_len) ? f5[dinc[n] >> 4] : 0; }
    int lvl3(int n) const { return (n >= lv_left + 1 && n <= lv_right && n >= 0 && n <= b_len) ? f3[dinc[n] & 15] : 0; }
    const SpdpScoringH* sc = nullptr;
    const SpdpSeedParams* sp = nullptr;
    DpBackendH* dp = nullptr;
    int lowest_level = 0;
    std::vector<Hsp> top_hsps;
    uint8_t mid[32], tron_of[64];               // the standard genetic code in the reference's tron alphabet

    Span cur{};
    std::vector<SpdpSkl> rec;
    bool is3end = false;
    bool unsupported = false;
    int why = 0;                                // the line that marked the walk as not served
    void mark(int line) { unsupported = true; if (!why) { why = line; if (getenv("SPDP_WALK_DEBUG")) fprintf(stderr, "walk_h: not served, line %d\n", line); } }
    int ss[2] = {0, 0};                         // Aln2h1::ss: the sites nearest_sites found

    enum { J_DIAGONAL, J_HEAD_NOGENOME, J_HEAD_CDS, J_HEAD_EXON, J_TAIL_NOGENOME, J_TAIL_CDS, J_TAIL_EXON, J_JUNCTION,
           J_MICRO_EXON, J_SHORTCUT, J_BACKFORTH, J_SMALL_DP, J_RECURSE, J_DP, J_GIVEUP_HEAD, J_GIVEUP_TAIL, J_GIVEUP_INNER,
           J_PICK_UNIT, J_EXACT_HEAD, J_EXACT_TAIL, J_COUNT };     // (the last two: terminal exons the exact search found)
    int joins[J_COUNT] = {0};

    // TraceBackDir, src/aln.h:30-35
    enum { DEAD = 0, DIAG = 2, NEWD = 3, VERT = 4, SLA1 = 5, SLA2 = 6, HORI = 8, HOR1 = 9, HOR2 = 10 };
    static bool is_diag(int d) { d &= 15; return d == DIAG || d == NEWD; }
    static bool is_vert(int d) { d &= 15; return (d >= 4 && d <= 7) || d == 12; }

    int NEV() const { return SPDP_NEVSEL; }
    bool Local() const { return (sp->lcl & 16) != 0; }
    bool LocalC() const { return Local() && (sp->lcl & 32); }
    int sim(int i, int n) const { return sc->mtx[a[i] * sc->mtx_cols + b[n]]; }
    int simc(int i, int tron) const { return sc->mtx[a[i] * sc->mtx_cols + tron]; }
    int int_pen(int len) const
    {
        if (len < 0) return SHRT_MIN;
        if (len >= sc->intpen_len) len = sc->intpen_len - 1;
        return sc->intpen[len];
    }
    int t53(int m, int n) const { return sc->t53[16 * (dinc[m] >> 4) + (dinc[n] & 15)]; }
    int sig53_5p3(int m, int n) const { return sig5[m] + sig3[n] + t53(m, n); }            // Exinon::sig53(.., IE5P3)
    int junction_score(int n5, int n3) const { return int_pen(n3 - n5) + sig3[n3] + t53(n5, n3); }   // SpJunc::spjscr
    int is_canon(int d, int ac) const
    {
        const int c5 = lvl5(d), c3 = lvl3(ac);
        return ((c5 == 3 && c3 == 3) || (c5 == 2 && c3 == 2) || (c5 == 1 && c3) || (c5 && c3 == 1)) ? c5 + c3 : 0;
    }
    // PwdB::GapPenalty3(i), src/aln2.cc:41-52
    int gap_penalty3(int i) const
    {
        if (i == 0) return 0;
        const int d = i / 3;
        const int x = i % 3 == 1 ? sc->gape1 : (i % 3 == 2 ? sc->gape2 : 0);
        return x + (i > sc->codonk1 ? sc->lgop * sc->gop / sc->gop + d * sc->lgep : sc->gop + d * sc->gep);
    }
    int gap_penalty(int i) const { return i == 0 ? 0 : (i > sc->codonk1 ? sc->lgop + i * sc->lgep : sc->gop + i * sc->gep); }
    int gap_ext_pen3(int i) const { return i > sc->codonk1 ? sc->lgep : sc->gep; }
    void put(int m, int n) { rec.push_back({m, n}); }
    int end_margin() const { return (int) (sp->vthr / sc->gep); }
    int slmt() const { return sp->vthr / 2; }
    // SpJunc::spjseq(n5, n3): the two codons an intron between n5 and n3 can split, as tron codes {phase 1, phase 2}.  A codon is
    // defined when its own three bases are: an ambiguous second or third base of the four leaves neither, an ambiguous
    // first (last) one the second (first) codon (spj_amb_tron_tab / spj_tron_amb_tab, src/codepot.cc:84-106)
    bool split_codon(int n5, int n3, int cs[2])
    {
        cs[0] = cs[1] = 2;                                                              // spj_tron_tab[256]: {AMB, AMB}
        if (n5 < cur.bl || n3 >= cur.br) return true;
        int w[4];
        const int at[4] = {n5 - 2, n5 - 1, n3, n3 + 1};
        for (int k = 0; k < 4; ++k) {
            if (at[k] < 0 || at[k] > b_len) { mark(__LINE__); return false; }
            w[k] = mid[b[at[k]] & 31];
        }
        if (n3 == 0) w[2] = w[3] = 3;                                                   // (`pyrim`: PHE PHE stands in for position 0)
        if (w[1] > 3 || w[2] > 3) return true;
        if (w[0] <= 3) cs[0] = tron_of[16 * w[0] + 4 * w[1] + w[2]];
        if (w[3] <= 3) cs[1] = tron_of[16 * w[1] + 4 * w[2] + w[3]];
        return true;
    }
    static bool avst_equal(int x, int y) { return x == y || (x == 18 && y == 23); }     // SER / SER2

    // stripe31(seqs, &wdw, shld, cmode), src/aln2.cc:178-198
    SpdpWindow stripe31(int shld, int cmode = 0) const
    {
        SpdpWindow w;
        if (shld < 0) shld = -shld * std::min(cur.ar - cur.al, cur.br - cur.bl) / 100;
        shld *= 3;
        w.up = cur.br - 3 * cur.ar;
        w.lw = cur.bl - 3 * cur.al;
        if (cmode == 1) w.lw = w.up;
        else if (cmode == 2) w.up = w.lw;
        else if (w.up < w.lw) std::swap(w.up, w.lw);
        w.up += shld; w.lw -= shld;
        w.up = std::min(w.up, cur.br - 3 * cur.al);
        w.lw = std::max(w.lw, cur.bl - 3 * cur.ar);
        w.width = w.up - w.lw + 7;
        return w;
    }

    int diagonal()
    {
        const bool LL = Local() && cur.a_exgl && cur.b_exgl, LR = Local() && cur.a_exgr && cur.b_exgr;
        int scr = 0, best = NEV(), mL = cur.al, mR = cur.ar;
        for (int m = cur.al, n = cur.bl + 1; m < cur.ar; n += 3) {
            scr += sim(m, n) + sigE[n];
            ++m;
            if (LL && scr < 0) { scr = 0; mL = m; }
            if (LR && scr > best) { best = scr; mR = m; }
        }
        put(mL, 3 * (mL - cur.al) + cur.bl);
        put(mR, 3 * (mR - cur.al) + cur.bl);
        return LR ? best : scr;
    }

    int creep_back(int ovr, int bscr, const Bound& lub)
    {
        int d = 0;
        while (cur.al > lub.la && cur.bl > lub.lb && (ovr < 0 || std::abs(d) <= bscr)) {
            d += sim(cur.al - 1, cur.bl - 2) + sigE[cur.bl - 2];
            --cur.al; cur.bl -= 3;
            if ((ovr += 3) == 0) bscr += d;
        }
        return d;
    }
    int creep_fwrd(int& ovr, int bscr, const Bound& lub)
    {
        int d = 0;
        while (cur.ar < lub.ua && cur.br < lub.ub && (ovr < 0 || std::abs(d) <= bscr)) {
            d += sim(cur.ar, cur.br + 1) + sigE[cur.br + 1];
            ++cur.ar; cur.br += 3;
            if ((ovr += 3) == 0) bscr += d;
        }
        return d;
    }

    int back_and_forth(int ovr, const Bound& lub)
    {
        const int oc = ovr / 3;
        std::vector<int> acc(oc + 1, 0);
        int scr = 0, i = oc, m = cur.al, n = cur.bl;
        int pa = cur.al, pb = cur.bl + 1;
        for (;;) {
            if (--i < 0) break;
            if ((m -= 3) < lub.la) break;               // (sic: the query position moves by three here)
            if ((n -= 3) < lub.lb) break;
            --pa; pb -= 3;
            acc[i] = scr += sim(pa, pb) + sigE[pb];
        }
        int best = scr;
        scr = 0;
        int where = ++i;
        m = cur.ar + i; n = cur.br + 3 * i;
        pa = m; pb = n;
        for ( ; ; n += 3, pb += 3) {
            if (!(i++ < oc)) break;
            if (!(m++ < lub.ua)) break;
            if (!(n < lub.ub)) break;
            scr += sim(pa++, pb);
            if ((acc[i] += scr) > best) { best = acc[i]; where = i; }
        }
        SpdpSkl k = {cur.ar + where, cur.br + 3 * where};
        rec.push_back(k);
        int dr = (cur.br - 3 * cur.ar) - (cur.bl - 3 * cur.al);
        if (dr >= 0) k.n -= dr; else k.m -= (dr = -dr) / 3;
        if (k.n >= 0) rec.push_back(k);
        return best + gap_penalty3(dr);
    }

    // the open reading frame runs on beyond an HSP at the 5' end: backwards to the start codon (sigS)
    int cds_end5(const SpdpSkl& at, int wmode)
    {
        int x = at.m, y = at.n;
        int pb = y + 1;
        int best = 0, scr = 0;
        SpdpSkl k = at;
        int pa = x;
        if (wmode && x == 0) rec.push_back(at);
        for ( ; y > cur.bl; y -= 3) {
            if (sigS[pb] > 0) scr += sigS[pb];
            if (scr > best) { best = scr; k.m = x; k.n = y; }
            if (sigS[pb] > 0 || scr + sp->vthr < 0) break;
            pb -= 3;
            scr += sigE[pb];
            if (x > 0) { --x; --pa; scr += sim(pa, pb); }       // (bs moves with bb: both are at y + 1 - 3)
            else scr += sc->gep;
        }
        if (wmode && best > 0) { rec.push_back(at); rec.push_back(k); }
        else if (wmode == 2) rec.push_back(at);
        return best;
    }
    int cds_end3(const SpdpSkl& at, int maxscr, int wmode)
    {
        int x = at.m, y = at.n;
        int pb = y + 1;
        int scr = maxscr;
        SpdpSkl k = at;
        int pa = x, ps = y + 1;
        if (wmode) rec.push_back(at);
        for ( ; y < cur.br; y += 3, pb += 3) {
            if (sigT[pb] > 0) scr += sigT[pb];
            else scr += sigE[pb] + sc->gep;
            if (scr > maxscr) { maxscr = scr; k.m = x; k.n = y + 3; }
            if (sigT[pb] > 0 || scr + sp->vthr < 0) break;
            if (x < a_len) { ++x; scr += sim(pa++, ps); ps += 3; }
        }
        if (wmode && k.n != at.n) { rec.push_back(at); rec.push_back(k); }
        else if (wmode == 2) rec.push_back(at);
        return maxscr;
    }

    bool indel_free_junction(int agap, int& iscr, bool write)
    {
        SpdpSkl k = {0, 0};
        const int dgap = cur.br - cur.bl - 3 * agap;
        const int d5 = cur.bl + 3 * agap - 2;
        const int d3 = cur.bl + 2;
        const int a5 = cur.br - 2;
        const int ntry = (sp->crs || agap) ? 1 : 2;
        agap = 2 - agap;
        const int reach = std::min(std::min(cur.al, cur.bl / 3), agap + 16);
        std::vector<int> bw(std::max(reach, 0) + 2, 0);
        int i = 0, v = 0;
        int pa = cur.al, pb = cur.bl + 1, pd = cur.bl + 1 + dgap, pe = cur.bl + 1;
        for (;;) {
            if (!(++i < reach)) break;
            pb -= 3; pd -= 3;
            if (pb < 0 || pd < 0 || pd > b_len) { mark(__LINE__); return false; }
            if (!(b[pb] == b[pd] || i < agap)) break;
            --pa; pe -= 3;
            bw[i] = v += sim(pa, pb) + sigE[pe];
        }
        if (i > 0) std::reverse(bw.begin(), bw.begin() + i);
        int phs53 = 0;
        iscr = NEV();
        bool all_mch = true;
        for (int nt = 0; nt < ntry && iscr == NEV(); ++nt) {
            int phs = 1;
            int m = cur.ar - 1;
            int qa = m, qb = a5;
            int n = d5;
            int t = 0;
            int qacc = a5;                              // ba: the SGPT6 entry that walks with n on the acceptor side
            for (v = 0; n <= d3; ++n, ++qacc) {
                if (n < 0) continue;
                if (nt || is_canon(n, n + dgap)) {
                    bool mch = true;
                    int y = sig53_5p3(n, n + dgap) - v - bw[t] + (cip ? cip[3 * m + phs] : 0);
                    if (phs) {
                        int cs[2];
                        if (!split_codon(n, n + dgap, cs)) return false;
                        const int c = cs[phs == 1 ? 1 : 0];
                        if (qa < 0 || qa >= a_len) { mark(__LINE__); return false; }
                        mch = sp->crs || avst_equal(a[qa], c);
                        if (mch) y += simc(qa, c); else y = NEV();
                    }
                    if (y > iscr) { k.n = n - phs; k.m = m; iscr = y; phs53 = -phs; all_mch = mch; }
                }
                if (++phs == 0) ++t;
                else if (phs == 2) { ++m; phs = -1; }
                else { ++qa; qb += 3; v += sim(qa, qb) + sigE[qacc + 1]; }
            }
        }
        if (!all_mch || iscr <= NEV()) return false;
        if (write) {
            rec.push_back(k);
            phs5.set(k.n, (int8_t) phs53);
            k.n += dgap;
            rec.push_back(k);
            phs3.set(k.n, (int8_t) phs53);
            iscr += int_pen(dgap);
        }
        return true;
    }

    // the (up to two) splice sites nearest to the open end of the genomic span, into ss[]; returns how many
    template <int SIDE>
    int nearest_sites(const Bound& bab)
    {
        const int from = SIDE == 5 ? cur.bl : cur.br;
        const int a0 = SIDE == 5 ? cur.al : cur.ar;
        auto strong = [&](int n, bool retry) {
            return SIDE == 5 ? (sig5[n] > sp->gc_sig5 || (retry && phs5[n] == 0)) : (sig3[n] > 0 || (retry && phs3[n] == 0));
        };
        auto sig = [&](int n) { return SIDE == 5 ? sig5[n] : sig3[n]; };
        int found[2], nss = 0;
        for (int retry = 0; ; ) {
            nss = 0;
            int pa = a0, ta = std::max(bab.la, pa - 5), nn = from, bb = from;
            for (int p = 0; pa > ta && nn > bab.lb; --nn, --bb) {
                if (strong(bb, retry != 0)) { if (nss < 2) found[nss++] = bb; else break; }
                else if ((++p % 3) == 0) --pa;
            }
            const int dd = nss ? from - found[nss - 1] : 5;
            pa = a0; ta = std::min(pa + 5, bab.ua); nn = from; bb = from;
            for (int p = 1; pa < ta && ++nn < bab.ub; ) {
                ++bb;
                if (strong(bb, retry != 0)) {
                    if (nss < 2) found[nss++] = bb;
                    else if (p < dd || sig(bb) > sig(found[nss - 1])) found[nss - 1] = bb;
                    if (nss == 2) break;
                } else if ((++p % 3) == 0) ++pa;
            }
            if (nss == 0) { if (!retry++) continue; return 0; }
            break;
        }
        if (nss == 2) {
            const int d0 = std::abs(found[0] - from), d1 = std::abs(found[1] - from);
            if (d0 > d1 || (d0 == d1 && sig(found[0]) < sig(found[1]))) std::swap(found[0], found[1]);
            if (sig(found[0]) > sig(found[1])) --nss;
        }
        for (int n = 0; n < nss; ++n) ss[n] = found[n];
        return nss;
    }

    // (n >= 0 ? n + 1 : n - 1) / 3: how many whole codons a shift of n nucleotides moves the query end
    static int codons_of(int d) { return (d >= 0 ? d + 1 : d - 1) / 3; }

    int micro_exon(const Bound& bab)
    {
        if (!nearest_sites<5>(bab)) return NEV();
        const int l = ss[0];
        if (!nearest_sites<3>(bab)) return NEV();
        const int r = ss[0];
        const Span keep = cur;
        auto restore = [&]() { cur.al = keep.al; cur.ar = keep.ar; cur.bl = keep.bl; cur.br = keep.br; };
        int d5 = codons_of(cur.bl - l);
        cur.al -= d5; cur.bl -= 3 * d5;
        d5 = cur.bl - l;
        int d3 = codons_of(cur.br - r);
        cur.ar -= d3; cur.br -= 3 * d3;
        d3 = cur.br - r;
        const int alen = cur.ar - cur.al;
        if (alen <= 0) {
            int scr = 0;
            if (indel_free_junction(alen, scr, true)) return scr;
            restore();
            return NEV();
        }
        int ts = cur.ar, s0 = cur.al;
        if (d5 == 1) --s0;
        if (d3 == 1) --ts;
        int best = NEV(), f = -1;
        const int cds = 3 * alen + d5 - d3;
        const int n9 = cur.br - cds - sp->minl;
        for (int n5 = cur.bl + sp->minl, n3 = n5 + cds; n5 < n9; ++n5, ++n3) {
            if (phs3[n5] || phs5[n3]) continue;
            int as = s0, bs = n5 + d5 + 1, ms = 0;
            if (d5) {
                int cs[2];
                if (!split_codon(l, n5, cs)) return NEV();
                int c = cs[0];
                if (d5 == -1) { c = cs[1]; bs += 3; }
                ms += simc(as++, c);
            }
            if (d3) {
                int cs[2];
                if (!split_codon(n5 + cds, r, cs)) return NEV();
                ms += simc(ts, cs[d3 == -1 ? 1 : 0]);
            }
            for ( ; as < ts; bs += 3) ms += sim(as++, bs);
            const float fs = sp->w2 * ms + sig53_5p3(l, n5) + sig53_5p3(n3, r) + int_pen(n5 - l) + int_pen(r - n3);
            const int scr = (int) fs;
            if (scr > best) { best = scr; f = n5; }
        }
        if (f < 0) { restore(); return NEV(); }
        SpdpSkl k = {cur.al, cur.bl};
        rec.push_back(k);
        k.n = f + d5; rec.push_back(k);
        k.n = f + cds + d3; k.m += alen; rec.push_back(k);
        k.n = cur.br; rec.push_back(k);
        return best;
    }

    // ungapped placement of a short first exon at a canonical site upstream of the acceptor (first_exon_wmm)
    int first_exon_wmm(int d3, int& retscr, bool& pm, int nss)
    {
        const int na = cur.br - d3;
        int n = std::max(cur.bl, cur.br - 3 * cur.ar - sp->minl);
        int nd = n + 3 * cur.ar - d3;
        int ts = cur.ar;
        const int s0 = cur.al;
        int pmch = 0, best = NEV();
        const float dfact = nss > 1 ? sp->w2 - 1 : 0.f;
        for (int as = s0; as < ts; ++as) pmch += simc(as, a[as] < sc->mtx_cols ? a[as] : 0);
        if (d3 == -1) pmch += simc(ts, a[ts]);
        else if (d3 == 1) --ts;
        int f = -1;
        for ( ; n >= cur.bl; --n, --nd) {
            if (sigS[n + 1] <= 0 || !is_canon(nd, na)) continue;
            int bs = n + 1, ms = 0;
            for (int as = s0; as < ts; bs += 3) ms += sim(as++, bs);
            if (d3) {
                int cs[2];
                if (!split_codon(nd, na, cs)) return -1;
                ms += simc(ts, cs[d3 == -1 ? 1 : 0]);
            }
            const int scr = (int) (sp->w2 * ms + sigS[n + 1] + sig5[nd] + junction_score(nd, na));
            if (scr > best) {
                f = n; best = scr;
                retscr = (int) (best - dfact * ms);
                pm = ms == pmch;
                if (pm && (na - nd) > sp->ip_mode) break;
            }
            if (sp->ip_maxl && ((na - nd) % sp->ip_maxl) == 0 && best > NEV()) break;
        }
        return f;
    }
    int first_exon(const Bound& bab)
    {
        const int nss = nearest_sites<3>(bab);
        if (nss == 0) { const SpdpSkl k = {cur.ar, cur.br}; return cds_end5(k, 2); }
        const int sites[2] = {ss[0], ss[1]};
        Span first = cur, second = cur;
        int maxf = -1, maxscr = NEV(), nn = 1;
        for (int n = 0; n < nss; ++n) {
            if (n) { cur.al = first.al; cur.ar = first.ar; cur.bl = first.bl; cur.br = first.br; }
            const int r = sites[n];
            int d3 = codons_of(cur.br - r);
            cur.ar -= d3; cur.br -= 3 * d3;
            if (cur.ar == 0 || cur.br < 3) { const SpdpSkl k = {cur.ar, cur.br}; return cds_end5(k, 2); }
            if (cur.al >= cur.ar || cur.bl >= cur.br) continue;
            d3 = cur.br - r;
            if (sp->crs || cur.ar < 2) {
                if (n == 0) second = cur;
                int scr = NEV(); bool pm = false;
                const int f = first_exon_wmm(d3, scr, pm, nss);
                if (unsupported) return NEV();
                if (scr > maxscr) { maxscr = scr; maxf = f; nn = n; if (pm) break; }
            } else {                                        // same-species mode: exact occurrences of the terminal stretch
                const int cds = 3 * cur.ar - d3;
                if (d3 == 1) --cur.ar;
                if (cur.ar - cur.al < 1) { mark(__LINE__); return NEV(); }
                ExactFinder3 bm(b, cur.bl, cur.br, a, cur.al, cur.ar, -1);
                int l = std::max(cur.bl, cur.br - sp->ip_maxl);
                const int as = cur.ar;
                while (!bm.finished()) {
                    int f = bm.next(l, -1) - 1;
                    if (f >= 0) {
                        const int nd = f + cds;
                        if (nd < 0 || nd > b_len || f + 1 > b_len + 2) { mark(__LINE__); return NEV(); }
                        if (is_canon(nd, r)) {
                            if (d3) {
                                int cs[2];
                                if (!split_codon(nd, r, cs)) return NEV();
                                if (as >= a_len + 1) { mark(__LINE__); return NEV(); }
                                if (!avst_equal(as < a_len ? a[as] : a_pad, d3 == -1 ? cs[1] : cs[0])) f = -1;
                            }
                            if (f >= 0) {
                                const int scr = sigS[f + 1] + sig5[nd] + junction_score(nd, r);
                                if (scr > maxscr) { maxscr = scr; maxf = f; }
                            }
                        }
                    }
                    if (bm.scanned(l)) {
                        if (maxf < 0) l = std::max(cur.bl, l - sp->ip_maxl);
                        else break;
                    }
                }
                if (d3 == 1) ++cur.ar;
                if (maxf >= 0) { ++joins[J_EXACT_HEAD]; nn = 1; break; }
            }
        }
        if (maxf < 0) { cur.al = first.al; cur.ar = first.ar; cur.bl = first.bl; cur.br = first.br; return NEV(); }
        if (nn == 0) { cur.al = second.al; cur.ar = second.ar; cur.bl = second.bl; cur.br = second.br; }
        cur.bl = maxf;
        put(cur.al, cur.bl);
        put(cur.ar, cur.bl + 3 * cur.ar);
        put(cur.ar, cur.br);
        return maxscr;
    }

    int last_exon_wmm(int d5, int& retscr, bool& pm, int nss)
    {
        const int l = cur.bl - d5;
        const int alen = cur.ar - cur.al;
        const int rr = cur.br - 3 * alen - d5 - 1;
        int n = cur.bl + sp->minl;
        const int ts = cur.ar;
        int s0 = cur.al;
        if (d5 == 1) --s0;
        int best = NEV();
        const float dfact = nss > 1 ? sp->w2 - 1 : 0.f;
        int pmch = 0;
        for (int as = s0; as < ts; ++as) pmch += simc(as, a[as]);
        int f = INT_MIN / 2;
        for (int bt = n + 3 * alen + d5 + 1, bss = n + d5 + 1; n < rr; ++n, ++bt, ++bss) {
            if (sigT[bt] <= 0 || !is_canon(l, n)) continue;
            int ms = 0, as = s0, bs = bss;
            if (d5) {
                int cs[2];
                if (!split_codon(l, n, cs)) return INT_MIN / 2;
                int c = cs[0];
                if (d5 == -1) { c = cs[1]; bs += 3; }
                ms += simc(as++, c);
            }
            for ( ; as < ts; bs += 3) ms += sim(as++, bs);
            const int scr = (int) (sp->w2 * ms + sigT[bt] + sig5[l] + junction_score(l, n));
            if (scr > best) {
                f = n; best = scr;
                retscr = (int) (best - dfact * ms);
                pm = ms == pmch;
                if (pm && (n - l) > sp->ip_mode) break;
            }
            if (sp->ip_maxl && ((n - l) % sp->ip_maxl) == 0 && best > NEV()) break;
        }
        return f + d5;
    }
    int last_exon(Bound& bab, const SpdpSkl& at)
    {
        if (cur.ar == a_len) ++bab.ua;
        const int nss = nearest_sites<5>(bab);
        if (nss == 0) return cds_end3(at, 0, 2);
        const int sites[2] = {ss[0], ss[1]};
        const Span first = cur;
        Span second = cur;
        int maxf = -1, maxscr = NEV(), alen = 0, alen_0 = 0, nn = 1;
        for (int n = 0; n < nss; ++n) {
            if (n) { cur.al = first.al; cur.ar = first.ar; cur.bl = first.bl; cur.br = first.br; }
            const int l = sites[n];
            int d5 = codons_of(cur.bl - l);
            cur.al -= d5; cur.bl -= 3 * d5;
            d5 = cur.bl - l;
            alen = cur.ar - cur.al;
            if (alen < 0) {                             // xxx'tg|gt...ag|a'
                // acodon[code] == 'W': tryptophan (src/seq.cc:59)
                if (b[first.bl + 1] != 20) return cds_end3(at, 0, 2);
                rec.push_back(at);
                return 0;
            } else if (alen == 0 && d5 == 0) { rec.push_back(at); return 0; }
            else if (alen == 0 && d5 == -1) {
                // ncodon[code] == 'T': the codes whose middle base is T (src/seq.cc:60)
                static const char ncodon[] = "--NCGAAGAAGATTATTCCCGATGGA";
                if (ncodon[b[first.bl] < 26 ? b[first.bl] : 0] != 'T') return cds_end3(at, 0, 2);
                rec.push_back(at);
                return 0;
            }
            if (sp->crs || alen < 2) {
                if (n == 0) { second = cur; alen_0 = alen; }
                int scr = NEV(); bool pm = false;
                const int f = last_exon_wmm(d5, scr, pm, nss);
                if (unsupported) return NEV();
                if (scr > maxscr) { maxscr = scr; maxf = f; nn = n; if (pm) break; }
            } else {                                        // same-species mode: exact occurrences of the terminal stretch
                int as = cur.al;
                if (d5 < 0) { ++cur.al; --alen; d5 += 3; }
                if (cur.ar - cur.al < 1) { mark(__LINE__); return NEV(); }
                ExactFinder3 bm(b, cur.bl, cur.br, a, cur.al, cur.ar, 1);
                int r = std::min(cur.br, l + sp->ip_maxl);
                if (d5 == 1) --as;
                while (!bm.finished()) {
                    int f = bm.next(-1, r) - 1;
                    if (f >= 0) {
                        const int na = f - d5;
                        if (is_canon(l, na)) {
                            if (d5) {
                                int cs[2];
                                if (!split_codon(l, na, cs)) return NEV();
                                if (as < 0) { mark(__LINE__); return NEV(); }
                                if (!avst_equal(a[as], d5 != 1 ? cs[1] : cs[0])) f = -1;
                            }
                            if (f >= 0) {
                                const int t = f + 3 * alen + 1;
                                if (t < 0 || t > b_len + 2) { mark(__LINE__); return NEV(); }
                                if (sigT[t] > 0) { maxscr = sig5[l] + junction_score(l, na); maxf = f; break; }
                            }
                        }
                    }
                    if (bm.scanned(r)) {
                        if (maxf < 0) r = std::min(cur.br, r + sp->ip_maxl);
                        else break;
                    }
                }
                if (d5 == 2) { --cur.al; ++alen; d5 -= 3; maxf -= 3; }
                if (maxf >= 0) { ++joins[J_EXACT_TAIL]; nn = 1; break; }
            }
        }
        if (maxf < 0) { cur.al = first.al; cur.ar = first.ar; cur.bl = first.bl; cur.br = first.br; return NEV(); }
        if (nn == 0) { cur.al = second.al; cur.ar = second.ar; cur.bl = second.bl; cur.br = second.br; alen = alen_0; }
        put(cur.al, cur.bl);
        put(cur.al, maxf);
        const SpdpSkl k = {cur.ar, maxf + 3 * alen};
        return cds_end3(k, maxscr, 2);
    }

    // ---- the intron-less X-drop extensions of an open end in three frames (back2ward5endH_ng / for2ward3endH_ng) -------
    // Row by row away from the last HSP; a cell takes the codon match, deletions of one / two / three nucleotides, or an
    // insertion from the cells one / two / three columns back (three rotating insertion states); the row ends where all
    // three frames have dropped Vthr below the best score seen, the column range follows the previous row's peaks.
    struct Cell { int val, ptr, dir; };
    struct Trail { std::vector<int> m, n, prev;
                   int add(int m_, int n_, int p) { m.push_back(m_); n.push_back(n_); prev.push_back(p); return (int) m.size() - 1; } };
    template <bool TOWARDS5>
    int end_extension(int* last, const SpdpWindow& w, Trail& vmf)
    {
        const int S = TOWARDS5 ? -1 : 1;
        const Cell black = {NEV(), 0, 0};
        const int width = w.width;
        if (width < 7) { mark(__LINE__); *last = 0; return NEV(); }
        std::vector<Cell> buf(2 * (size_t) width, black);
        auto H = [&](int r) -> Cell& { return buf[r - w.lw + 3]; };
        auto F = [&](int r) -> Cell& { return buf[width + r - w.lw + 3]; };
        auto inbuf = [&](int r) { const long i = (long) r - w.lw + 3; return i >= 0 && i < width; };
        const int m_corner = TOWARDS5 ? cur.ar : cur.al, m_last = TOWARDS5 ? cur.al : cur.ar;
        const int n_corner = TOWARDS5 ? cur.br : cur.bl;
        int best_val = 0, best_m = m_corner, best_n = n_corner, best_p = 0;
        int maxval = 0;
        vmf.add(0, 0, 0);
        {   // pbinitH_ng / pfinitH_ng
            int r = n_corner - 3 * m_corner;
            H(r).val = 0; H(r).dir = DIAG; H(r).ptr = vmf.add(m_corner, n_corner, 0);
            const int rr = TOWARDS5 ? std::min(w.up, cur.br - 3 * cur.al) : std::max(w.lw, cur.bl - 3 * cur.ar);
            for (int i = 1; TOWARDS5 ? ++r <= rr : --r >= rr; ++i) {
                if (!inbuf(r)) { mark(__LINE__); return NEV(); }
                if (i <= 3) {
                    H(r) = H(r + S * i);
                    H(r).val += gap_penalty(i);
                    if (i < 3) H(r).val += sc->extragop;
                    H(r).dir = VERT;
                } else {
                    H(r) = H(r + S * 3);
                    H(r).val += gap_ext_pen3(i);
                }
            }
        }
        int m = m_corner;
        if (TOWARDS5 ? !cur.a_exgr : !cur.a_exgl) m -= S;
        int n1, n2;
        if (TOWARDS5) { n1 = 3 * m + w.lw; n2 = 3 * m + w.up + 1; }
        else { n1 = 3 * m + w.lw - 1; n2 = 3 * m + w.up; best_val = maxval = NEV(); }
        // mxd: the best diagonal cell of the current block, by reference: an H / F entry or one of the three insertion states
        enum { R_NONE, R_H, R_F, R_E };
        for (;;) {
            m += S;
            if (TOWARDS5 ? m < cur.al : m > cur.ar) break;
            n1 += 3 * S; n2 += 3 * S;
            const int n0 = TOWARDS5 ? std::min(n2, cur.br) : std::max(n1, cur.bl);
            const int n9 = TOWARDS5 ? std::max(n1, cur.bl) : std::min(n2, cur.br);
            int n = n0;
            int r = n - 3 * m;
            int count3 = 0;
            Cell e1[3] = {black, black, black};
            if (!inbuf(r)) { mark(__LINE__); return NEV(); }
            if ((TOWARDS5 ? !cur.b_exgr : !cur.b_exgl) && n == n_corner && m == m_corner) { e1[2] = H(r); e1[2].val = sc->gapw3; }
            int nr[3];
            for (int p = 0; p < 3; ++p) nr[((n + S * -p) % 3 + 3) % 3] = n - S * p;      // nr[(n -/+ p) % 3] = n -/+ p
            int ref_kind = (H(r).val + sp->vthr < maxval) ? R_NONE : R_H, ref_idx = r;
            nr[(n % 3 + 3) % 3] = n - 3 * S;
            auto ref_cell = [&]() -> const Cell& {
                return ref_kind == R_H ? H(ref_idx) : ref_kind == R_F ? F(ref_idx) : ref_kind == R_E ? e1[ref_idx] : black;
            };
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
                // TOWARDS5: sigE of the position the sweep leaves (bb before its decrement); else of position n - 2
                const int se = TOWARDS5 ? sigE[n + 1] : sigE[n - 2];
                int mx = 0;                             // 0: h, 1: f, 2: eq
                if (!corner_row) {
                    if (TOWARDS5 ? n > cur.br - 3 : n < cur.bl + 3) h = black;
                    else {
                        const bool was_diag = is_diag(h.dir);
                        h.val += sc->mtx[am * sc->mtx_cols + b[TOWARDS5 ? n + 1 : n - 2]] + se;
                        h.dir = was_diag ? DIAG : NEWD;
                    }
                    const int y = F(r + 3 * S).val + sc->gep;
                    {   // one nucleotide deleted
                        const Cell& fr = H(r + S);
                        const int x = fr.val + (is_vert(fr.dir) ? sc->gape1 : sc->gapw1);
                        if (x > y) { f = fr; f.val = x; f.dir = SLA2; } else f.val = y;
                    }
                    {   // two
                        const Cell& fr = H(r + 2 * S);
                        const int x = fr.val + (is_vert(fr.dir) ? sc->gape2 : sc->gapw2);
                        if (x > f.val) { f = fr; f.val = x; f.dir = SLA1; }
                    }
                    {   // a codon
                        const Cell& fr = H(r + 3 * S);
                        const int x = fr.val + sc->gapw3;
                        if (x >= f.val) { f = fr; f.val = x; f.dir = VERT; }
                        else if (y >= f.val) { f = F(r + 3 * S); f.val = y; f.dir = VERT; }
                    }
                    if (f.val >= h.val) mx = 1;
                }
                // insertions: three, two, one nucleotide(s) back along the row
                if (TOWARDS5 ? n < n0 - 2 : n > n0 + 2) {
                    const Cell& fr = H(r - 3 * S);
                    const bool stop = !TOWARDS5 && m == cur.ar && sigT[n - 2] > 0;      // the stop codon ends the forward form
                    int x = fr.val + (stop ? sigT[n - 2] : sc->gapw3);
                    const int y = eq.val += sc->gep;
                    if (x > y) {
                        eq = fr; eq.val = x;
                        if (TOWARDS5) { if (eq.dir) eq.dir = HORI; }
                        else eq.dir = stop ? DEAD : HORI;
                    }
                    if (!stop) eq.val += se;
                }
                if (TOWARDS5 ? n < n0 - 1 : n > n0 + 1) {
                    const Cell& fr = H(r - 2 * S);
                    const int x = fr.val + sc->gapw2;
                    if (x > eq.val) { eq = fr; eq.val = x; eq.dir = HOR2; }
                }
                {
                    const Cell& fr = H(r - S);
                    const int x = fr.val + sc->gapw1;
                    if (x > eq.val) { eq = fr; eq.val = x; eq.dir = HOR1; }
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
            if (TOWARDS5) {
                if (!ref_cell().dir) break;
                if (peak) n1 = n + 3;
            } else {
                if (peak) n2 = n - 3;
                if (!ref_cell().dir) break;
            }
        }
        *last = vmf.add(best_m, best_n, best_p);
        if (!TOWARDS5) is3end = true;
        return best_val;
    }

    int open_end(int cmode)
    {
        Trail vmf;
        int ptr = 0;
        if (cmode == 3) {
            const int room = a_len - cur.ar;
            if (cur.al > room) { cmode = 2; cur.ar = a_len; }
            else { cmode = 1; cur.al = 0; rec.clear(); }
        }
        const SpdpWindow w = stripe31(sc->sh, cmode);
        const int scr = cmode == 1 ? end_extension<true>(&ptr, w, vmf) : end_extension<false>(&ptr, w, vmf);
        for (int p = ptr; p; p = vmf.prev[p]) put(vmf.m[p], vmf.n[p]);
        return scr;
    }

    int interpolate(unsigned level, const int cmode, const Hsp* wjxt, Bound& bab)
    {
        if (is3end) return 0;
        int agap = cur.ar - cur.al, bgap = cur.br - cur.bl;
        int ovr = std::min(3 * agap, bgap);
        const int dgap = bgap - 3 * agap;
        int wlmt = level <= 3 ? sp->wl_width[level] : 0;
        ++level;
        if (sp->crs == 0 && cmode != 3) wlmt *= 3;
        const bool no_rec = ovr <= wlmt;
        int iscore = NEV(), scr = 0;
        std::vector<SpdpSkl> saved;
        bool have_saved = false;

        if (dgap == 0 && agap) {
            ++joins[J_DIAGONAL];
            scr += diagonal();
            iscore = 0;
        } else if (cmode == 1 && no_rec) {
            if (bgap < 0) {
                ++joins[J_HEAD_NOGENOME];
                cur.al -= bgap / 3; cur.bl -= bgap;
                put(cur.al, cur.bl);
                iscore = 0;
            } else {
                if (wjxt && (sp->crs || cur.ar == cur.al)) { ++joins[J_HEAD_CDS]; const SpdpSkl k = {wjxt->jx, wjxt->jy}; iscore = cds_end5(k, 1); }
                if (iscore <= 0) {
                    std::vector<SpdpSkl> before = rec;
                    const int kscore = first_exon(bab);
                    if (kscore > iscore) { ++joins[J_HEAD_EXON]; iscore = kscore; }
                    else rec.swap(before);
                }
            }
        } else if (cmode == 2 && no_rec) {
            if (bgap <= 0) {
                ++joins[J_TAIL_NOGENOME];
                cur.al -= bgap / 3; cur.bl -= bgap;
                put(cur.al, cur.bl);
                iscore = 0;
            } else {
                const SpdpSkl k = {cur.al, cur.bl};
                if (sp->crs || cur.ar == cur.al) { ++joins[J_TAIL_CDS]; iscore = cds_end3(k, 0, 1); }
                if (iscore <= 0) {
                    std::vector<SpdpSkl> before = rec;
                    const int kscore = last_exon(bab, k);
                    if (kscore > iscore) { ++joins[J_TAIL_EXON]; iscore = kscore; }
                    else rec.swap(before);
                }
            }
        } else if (cmode == 3 && agap <= 1 && dgap >= sp->minl && indel_free_junction(agap, iscore, true)) {
            ++joins[J_JUNCTION];
        } else if (cmode == 3 && no_rec && dgap >= sp->minl) {
            if (sp->crs == 0) { iscore = micro_exon(bab); if (iscore != NEV()) ++joins[J_MICRO_EXON]; }
            if (iscore == NEV() && agap < sp->elmt) { ++joins[J_SHORTCUT]; iscore = shortcut(ovr, bab); }
        } else if (ovr <= 0 && dgap < sp->minl) {
            ++joins[J_BACKFORTH];
            iscore = back_and_forth(-ovr, bab);
        } else if (dgap < sp->minl) {
            ++joins[J_SMALL_DP];
            scr -= creep_back(ovr, slmt(), bab);
            scr -= creep_fwrd(ovr, slmt(), bab);
            const bool abnormal = cur.bl > cur.br;
            if (abnormal) std::swap(cur.bl, cur.br);
            const SpdpWindow w = stripe31(std::min(sc->sh, std::abs(dgap) + 3));
            iscore = dp->trcbk(cur, w, false, nullptr, rec);
            if (abnormal) std::swap(cur.bl, cur.br);
        } else if ((int) level < sp->qck) {
            ++joins[J_RECURSE];
            saved = rec; have_saved = true;
            iscore = seeded(level, cmode, bab);
        }
        if (unsupported) return NEV();
        const int max_agap = (sp->desert && cmode < 3) ? sp->desert * (4 - (int) level) : INT_MAX;
        if (iscore == NEV() && (no_rec || (int) level == sp->qck) && agap < max_agap && !(LocalC() && sp->qck == 3 && cmode < 3)) {
            const Span before = cur;
            if (cmode & 1) scr -= creep_fwrd(ovr, slmt(), bab);
            if (cmode & 2) scr -= creep_back(ovr, slmt(), bab);
            agap += before.al - cur.al + cur.ar - before.ar;
            bgap += before.bl - cur.bl + cur.br - before.br;
            if (have_saved) rec = saved; else { saved = rec; have_saved = true; }
            const SpdpWindow w = stripe31(sc->sh);
            ++joins[J_DP];
            iscore = dp->lsp(cur, w, rec);
        }
        if (iscore == NEV()) {
            if (have_saved) rec = saved;
            if (cmode == 1) {
                ++joins[J_GIVEUP_HEAD];
                if (wjxt) { const int bl = wjxt->jy + end_margin(); if (bl > cur.bl) cur.bl = bl; }
                iscore = open_end(cmode);
            } else if (cmode == 2) {
                ++joins[J_GIVEUP_TAIL];
                if (wjxt) { const int br = bgap - wjxt->jy - end_margin(); if (br > cur.bl && br < cur.br) cur.br = br; }
                iscore = open_end(cmode);
            } else {
                ++joins[J_GIVEUP_INNER];
                if (Local()) iscore = open_end(cmode);
                else { ++joins[J_SHORTCUT]; iscore = shortcut(ovr, bab); }
            }
        }
        return scr + iscore;
    }

    // ---- Aln2h1::shortcutH_ng (:2232-2260): one traceback sweep over both flanks of a gap the HSPs leave open, the
    // genomic middle (all but minl at either side, whole codons) jumped over as one insertion
    int shortcut(int ovr, const Bound& bab)
    {
        const int margin = sp->minl;
        int interval = cur.br - cur.bl - 2 * margin;
        interval = interval > 0 ? interval / 3 * 3 : 0;
        const int cut[2] = {cur.bl + margin, cur.bl + margin + interval};
        int scr = 0;
        ovr = (ovr > 0 ? 0 : ovr) - 3;
        scr -= creep_back(ovr, slmt(), bab);
        scr -= creep_fwrd(ovr, slmt(), bab);
        const int alen = cur.ar - cur.al;
        int sh = alen / 2;
        if (sc->sh < 0) {
            float f = (float) -sc->sh;
            if (f > 1.f) f /= 100;
            if (f < 0.5f) sh = (int) (alen * f);
        } else if (sc->sh < sh) sh = sc->sh;
        sh = std::max(sh, alen - margin / 3);
        const SpdpWindow w = stripe31(sh);
        cur.a_exgl = cur.b_exgl = cur.a_exgr = cur.b_exgr = 0;      // stay so: the callers put their own flags back
        scr += dp->trcbk(cur, w, true, interval ? cut : nullptr, rec);
        return scr;
    }

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
            if (!sp->crs && agap > (cmode == 1 ? 0 : 1)) continue;
            int iscore = NEV(), jscore = 0;
            if (cmode == 1) { const SpdpSkl k = {jxt->jx, jxt->jy}; jscore = w.scr + cds_end5(k, 0); }
            else if (indel_free_junction(agap, iscore, false)) jscore = w.scr + iscore;
            else continue;
            jxt = w.jxt.data() + w.num - 1;
            cur.al = jxt->jx + jxt->jlen; cur.bl = jxt->jy + 3 * jxt->jlen;
            cur.ar = keep.ar; cur.br = keep.br;
            agap = jxt[1].jx - cur.al;
            if (!sp->crs && agap > (cmode == 2 ? 0 : 1)) continue;
            if (cmode == 2) {
                const SpdpSkl k = {cur.al, cur.bl};
                iscore = cds_end3(k, 0, 0);
                if (iscore > 0) jscore += iscore; else continue;
            } else if (indel_free_junction(agap, iscore, false)) jscore += iscore;
            else continue;
            if (jscore > best) { best = jscore; which = (int) u; }
        }
        cur.al = keep.al; cur.ar = keep.ar; cur.bl = keep.bl; cur.br = keep.br;
        return best > NEV() ? which : -1;
    }

    int seeded(unsigned level, int eimode, const Bound& lub)
    {
        const Span at_entry = cur;
        int cmode = eimode, scr = 0;
        std::vector<Unit> units;
        std::vector<Hsp>* list = nullptr;
        int num = 0;
        const int wlmt = level <= 3 ? sp->wl_width[level] : 0;
        Bound bab = lub;
        if ((int) level == lowest_level && !top_hsps.empty()) {
            list = &top_hsps;
            num = (int) top_hsps.size() - 1;
            for (int k = 0; k < num; ++k) {             // addsigEjxt: the coding potential along every HSP joins its score
                int s = 0;
                for (int i = 0, n = top_hsps[k].jy + 1; i < top_hsps[k].jlen; ++i, n += 3) s += sigE[n];
                top_hsps[k].jscr += s;
            }
        } else {
            if (!dp->wilip((int) level, cur, units)) { mark(__LINE__); return NEV(); }
            const int nwlu = (int) units.size();
            int pick = nwlu ? 0 : -1;
            if (nwlu > 1 && cur.br - cur.bl >= sp->minl) { ++joins[J_PICK_UNIT]; pick = best_unit(units, cmode); }
            if (unsupported) return NEV();
            if (pick >= 0) { list = &units[pick].jxt; num = units[pick].num; }
            else if (nwlu > 1) level = sp->qck - 1;
        }
        const Hsp* wjxt = nullptr;
        if (num) {
            std::vector<Hsp>& jxt = *list;
            jxt[num].jx = cur.ar;
            jxt[num].jy = cur.br;
            cur.a_exgr = 0; cur.b_exgr = 0;
            for (int k = 0; k < num; ++k) {
                const Hsp& h = jxt[k];
                scr += h.jscr;
                cur.ar = h.jx; cur.br = h.jy;
                bab.ua = std::min(h.jx + h.jlen, jxt[k + 1].jx) - wlmt;
                bab.ua = std::max(bab.ua, h.jx + h.jlen / 2);
                bab.ub = h.jy + 3 * (bab.ua - h.jx);
                if (cmode == 2) cmode = 3;
                const int iscore = interpolate(level, cmode, &h, bab);
                if (unsupported) return NEV();
                if (iscore > NEV()) {
                    cmode = 3;
                    scr += iscore;
                    cur.al = h.jx + h.jlen; cur.bl = h.jy + 3 * h.jlen;
                    cur.a_exgl = 0; cur.b_exgl = 0;
                    bab.la = cur.ar; bab.lb = cur.br;
                }
            }
            wjxt = &jxt[num];
            cur.a_exgr = at_entry.a_exgr; cur.b_exgr = at_entry.b_exgr;
            cur.ar = at_entry.ar; cur.br = at_entry.br;
            bab.ua = lub.ua; bab.ub = lub.ub;
            if (eimode == 2 || ((int) level == lowest_level && eimode == 1)) cmode = 2;
        }
        const int iscore = interpolate(level, cmode, wjxt, bab);
        if (unsupported) return NEV();
        if (iscore > NEV()) scr += iscore; else scr = NEV();
        cur.al = at_entry.al; cur.ar = at_entry.ar; cur.bl = at_entry.bl; cur.br = at_entry.br;
        cur.a_exgl = at_entry.a_exgl; cur.b_exgl = at_entry.b_exgl;
        if ((int) level == lowest_level && list == &top_hsps && wjxt) { top_hsps[num].jx = a_len; top_hsps[num].jy = b_len; }
        return scr;
    }

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

inline bool bind_problem_h(SeedWalkH& w, const SpdpScoringH* sc, const SpdpSeedParams* sp, const SpdpProblemH* p,
                           const SpdpJuxt* hsps, int n_hsps, int lowest_level)
{
    if (!sc || !sp || !p || !p->a || !p->b || !p->sig5 || !p->sig3 || !p->sigS || !p->sigT || !p->sigE || !p->phs5 || !p->phs3 ||
        !p->dinc || !sc->intpen || sc->intpen_len <= 0 || sp->qck < 1 || sp->qck > 3) return false;
    w.a = p->a; w.a_len = p->a_len; w.a_pad = p->a_pad; w.b = p->b; w.b_len = p->b_len;
    w.sig5 = p->sig5; w.sig3 = p->sig3; w.sigS = p->sigS; w.sigT = p->sigT; w.sigE = p->sigE; w.dinc = p->dinc; w.cip = p->cip;
    w.sc = sc; w.sp = sp; w.lowest_level = lowest_level;
    const int N = p->b_len + 3;
    w.phs5.bind(p->phs5); w.phs3.bind(p->phs3);
    (void) N;
    spdp_genetic_code_tables(w.mid, w.tron_of);
    {   // canonical-site levels by dinucleotide class (Exinon::intron53_c, src/codepot.cc:435-475), as in bind_problem
        static const uint8_t lac[4] = {0, 2, 3, 1}, lgt[4] = {0, 0, 3, 1};
        const int any = sp->any & 3;
        const uint8_t base = any == 3 ? 1 : 0, gt = lgt[any], ac = lac[any], bo = sp->both_ori ? 1 : 0;
        uint8_t* f5 = w.f5;
        uint8_t* f3 = w.f3;
        for (int c = 0; c < 16; ++c) f5[c] = f3[c] = base;
        enum { AA, AC, AG, AT, CA, CC, CG, CT, GA, GC, GG, GT, TA, TC, TG, TT };
        f3[AA] = ac; f3[AC] = 2; if (bo) f5[AC] = 1;
        f3[AG] = 3; f5[AT] = 2; f3[AT] = ac; f3[CG] = gt; f5[CT] = gt; if (bo) f3[CT] = 1;
        f5[GA] = gt; f5[GC] = 3; f5[GG] = gt; f3[GG] = gt; f5[GT] = 3; if (bo) f3[GT] = 1; f3[TG] = gt; f5[TT] = gt;
        w.lv_left = p->exin_left; w.lv_right = p->exin_right;
    }
    w.top_hsps.clear();
    if (hsps && n_hsps > 0)
        for (int j = 0; j <= n_hsps; ++j) w.top_hsps.push_back({hsps[j].jx, hsps[j].jy, hsps[j].jlen, hsps[j].nid, hsps[j].jscr});
    return true;
}

}   // namespace spdp_seed
#endif
