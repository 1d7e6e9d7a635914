// spdp_wilip.h -- the word-lookup HSP search of the reference (Wilip / Wlp, ogotoh/spaln v3.0.7 src/wln.cc), host side.
//
// SURVEY 8 row f4, second slice (round 5).  The seeded walks (spdp_walk.h) ask for the HSPs of a sub-range at a recursion
// level; FindHsp of the block search asks for them on a candidate region (level -1).  Until round 5 the library took them
// from the caller through SpdpHspSource (the reference's own Wilip behind a callback); with a SpdpWilipModel it finds them
// itself.  What is restated:
//   Wlp::Wlp / foldseq / lookup        src/wln.cc:210-232, 291-320, 253-270   the query's k-mers under the level's reduced
//                                                                             alphabet and bit pattern, chained by word
//   Bitpat / Bitpat_wq                 src/bitpat.cc:109-211                  spaced / contiguous words with a flaw state
//   Wlp::dmsnno / dmsnno31 / scan_b    src/wln.cc:554-678                     the scan of the genomic side, scores by diagonal
//   Wlp::enter / storedh               src/wln.cc:471-500, 537-552            a diagonal's run becomes an HSP record
//   Wlp::reeval / eval                 src/wln.cc:358-469                     extension, Kadane trimming, end bonuses
//   Wlp::mkhsps / LinkHspScr / jxtcore src/wln.cc:680-926                     sparse chaining DP, units, their bounds
//   Wlp::willip, Wilip::Wilip          src/wln.cc:955-992
// The reference sorts with qsort; glibc's is a merge sort for these sizes, i.e. stable: std::stable_sort here.
// Header only (std::vector, no device code): compiled into libspdp_hip.so, and by the host compiler into the tests' checker
// (oracle/walk_check.cpp), which pins it to every Wilip reply the reference's own runs recorded (tests/golden/q*_*.spdg).
#ifndef SPDP_WILIP_H_
#define SPDP_WILIP_H_

#include <stdint.h>
#include <limits.h>
#include <algorithm>
#include <vector>
#include "../../include/spdp.h"

namespace spdp_wl {

constexpr uint32_t BAD_RES = 0xff;              // src/bitpat.h:28
constexpr uint32_t BAD_WORD = 0xffffffffu;      // BadWord, src/bitpat.h:32
constexpr int NEVSEL_ = INT32_MIN / 16 * 7;     // src/cmn.h:79

struct Juxt { int jx, jy, jlen, nid, jscr; };   // JUXT, src/seq.h:174
struct Unit { int num, nid, tlen, llmt, ulmt, scr; std::vector<Juxt> jxt; };    // WLUNIT + its num + 1 records

// the two sequences as the reference's Seq objects present them to Wilip
struct Pair {
    const uint8_t* a; int a_len, a_left, a_right, a_exgl, a_exgr;
    const uint8_t* b; int b_len, b_left, b_right;
    int bbt;                                    // 1: nucleotide query, 3: protein query against tron codes
    const int16_t* sigS; const int16_t* sigE; const int16_t* sigT;      // protein only: Exinon (SGPT6) by position, or null
    const int16_t* intpen; int intpen_len;      // IntronPenalty::Penalty(len)
    int gop, gep, lgop, lgep, codonk1;          // PwdB::GapPenalty
};

// Bitpat_wq(elms, nframe, false, bitmask(width), spat)
struct Words {
    int weight = 0, width = 0, nalpha = 0, nframe = 1, noq = 1;
    uint32_t msb = 0, tabsize = 0;
    std::vector<int> exam;
    std::vector<uint32_t> queue, fstat;
    std::vector<int> qp;
    bool spaced() const { return width > weight; }
    void init(int elms, int nf, int wid, const uint8_t* spat, int spat_len)
    {
        nalpha = elms; nframe = nf;
        exam.clear();
        if (spat_len > 0) {
            width = spat_len; weight = 0;
            for (int w = 0; w < width; ++w) if (spat[w]) ++weight;
            for (int w = 0; w < width; ++w) if (spat[w]) exam.push_back(w);
            for (int w = 0; w < width; ++w) if (spat[width - w - 1]) exam.push_back(w);
        } else {                                // npat = bitmask(width): a contiguous seed
            width = weight = wid;
            exam.resize(2 * weight);
            for (int w = 0; w < weight; ++w) exam[w] = exam[w + weight] = w;
        }
        msb = 1u << (weight - 1);
        noq = spaced() ? nframe : (nframe > 1 ? 2 : 1);
        uint64_t t = 1;
        for (int i = 0; i < weight; ++i) t *= (uint64_t) nalpha;
        tabsize = (uint32_t) t;
        queue.assign((size_t) std::max(noq, nframe) * width, 0);
        fstat.assign(nframe, 0);
        qp.assign(std::max(noq, nframe), 0);
        clear();
    }
    void clear()
    {
        std::fill(queue.begin(), queue.end(), spaced() ? BAD_RES : 0u);
        std::fill(fstat.begin(), fstat.end(), msb);
        std::fill(qp.begin(), qp.end(), 0);
    }
    bool good(uint32_t c) const { return c < (uint32_t) nalpha; }
    bool flawless(int f = 0) const { return fstat[f] == 0; }
    void flaw(int f = 0)
    {
        if (spaced()) {
            queue[qp[f] + f * width] = BAD_RES;
            if (++qp[f] == width) qp[f] = 0;
        } else { queue[f] = 0; fstat[f] = msb; }
    }
    uint32_t word(uint32_t c, int f = 0)
    {
        if (!spaced()) {
            fstat[f] >>= 1;
            queue[f] = (uint32_t) (((uint64_t) queue[f] * nalpha + c) % tabsize);
            return fstat[f] ? BAD_WORD : queue[f];
        }
        const int offset = f * width;
        queue[qp[f] + offset] = c;
        if (++qp[f] == width) qp[f] = 0;
        uint32_t w = 0;
        for (int k = 0; k < weight; ++k) {
            int q = qp[f] + exam[k];
            if (q >= width) q -= width;
            if (queue[q + offset] == BAD_RES) { fstat[f] = 1; return BAD_WORD; }
            w = w * nalpha + queue[q + offset];
        }
        fstat[f] = 0;
        return w;
    }
};

struct Jxtd { int score, mxscr, prevj, lastj, maxj; };      // JXTD: {score | nhit, mxscr | ml, prevj | mr, lastj, maxj}

struct Wlp {
    const SpdpWilipModel* M; const Pair* P;
    SpdpWilipLevel L;                           // a copy: level -1 on a short query scales cutoff and vthr
    int bbt, mm, sect_l, tplwt, awspan, bwspan, precutoff, min_lnkscr;
    bool dhit;
    std::vector<uint32_t> position, header;
    std::vector<Jxtd> jxtd;
    std::vector<Juxt> mfd;
    Words bpp;
    bool ng = true;
    int conv(int code) const { return L.convtab[code & 31]; }
    int sim2(int ac, int bc) const { return M->mtx[ac * M->mtx_cols + bc]; }
    int gap_penalty(int i) const { return i == 0 ? 0 : (i > P->codonk1 ? P->lgop + i * P->lgep : P->gop + i * P->gep); }
    int intpen_plus(int n) const               // IntronPenalty::PenaltyPlus, src/codepot.h:248
    {
        if (n < M->llmt) return SHRT_MIN;
        const int k = std::min(n, P->intpen_len - 1);
        return (int) (int16_t) (P->intpen[k] + M->avrsig);
    }

    Wlp(const SpdpWilipModel* m, const Pair* p, int level) : M(m), P(p)
    {
        L = m->level[std::max(level, 0)];
        bbt = p->bbt; mm = p->a_right - p->a_left; sect_l = bbt * mm;
        tplwt = L.tpl * L.gain; awspan = L.width - 1; bwspan = 3 * L.width - 1;
        precutoff = L.cutoff - L.gain * L.tpl;
        min_lnkscr = -L.vthr / 2;
        dhit = m->crs && level > 1;
        if (mm <= awspan) return;
        if (level < 0 && p->a_len < m->shortquery) {
            L.cutoff = L.cutoff * p->a_len / m->shortquery;
            L.vthr = L.vthr * p->a_len / m->shortquery;
            precutoff = precutoff * p->a_len / m->shortquery;
        }
        if (!foldseq()) return;
        lookup();
        ng = false;
    }
    bool foldseq()
    {
        const int nk = mm - awspan;
        if (nk <= 0) return false;
        position.assign(nk + 1, 0);
        bpp.init(L.elem, 1, L.width, L.bitpat, L.bitpat_len);
        int ps = P->a_left;
        const int ts = P->a_left + awspan;
        while (ps < ts) {
            const uint32_t c = conv(P->a[ps++]);
            if (bpp.good(c)) bpp.word(c); else bpp.flaw();
        }
        const int te = P->a_right;
        for (int s = 0; ps < te; ++s) {
            const uint32_t c = conv(P->a[ps++]);
            if (bpp.good(c)) {
                const uint32_t w = bpp.word(c);
                position[s] = bpp.flawless() ? w : (uint32_t) L.mask + 1;
            } else { position[s] = (uint32_t) L.mask + 1; bpp.flaw(); }
        }
        position[nk] = position[0];
        return true;
    }
    void lookup()
    {
        header.assign((size_t) L.mask, 0);
        const int kk = mm - (L.width - 1);
        for (int k = 0, s = 0; k++ < kk; ++s) {
            if (position[s] < (uint32_t) L.mask) {
                const uint32_t m = header[position[s]];
                header[position[s]] = (uint32_t) k;
                position[s] = m;
            } else position[s] = 0;
        }
    }
    void enter(const Jxtd& w, int r)
    {
        Juxt j;
        j.jx = w.lastj; j.jy = bbt * w.lastj + r; j.jlen = w.maxj - w.lastj + L.width; j.nid = 0; j.jscr = w.mxscr;
        mfd.push_back(j);
    }
    void storedh(int r, int ml, int mr)
    {
        ml -= L.width; mr += 2 * L.width;
        int x = (r < 0) ? (bbt - r - 1) / bbt : ml;
        if (x < 0) x = 0;
        const int y = r + bbt * x + (bbt == 3 ? 1 : 0);
        int as = P->a_left + x, bs = P->b_left + y;
        const int at = std::min(P->a_right, P->a_left + mr), bt = P->b_right;
        int scr = 0, maxscr = 0;
        int ms = ml = mr = x;
        for (int m = x; as < at && bs < bt; ++as, bs += bbt, ++m) {
            scr += sim2(P->a[as], P->b[bs]);
            if (scr <= 0) { scr = 0; ms = m; }
            else if (scr > maxscr) { maxscr = scr; mr = m; ml = ms; }
        }
        if (maxscr > L.vthr) { Juxt j = {ml + 1, r + bbt * (ml + 1), mr - ml, 0, 0}; mfd.push_back(j); }
    }
    void scan_b(uint32_t m, uint32_t n)
    {
        for ( ; m; m = position[m]) {
            const int r = (int) n - (int) --m * bbt;
            Jxtd& w = jxtd[(r + sect_l) % sect_l];
            if (dhit) {
                if (!w.score) w.mxscr = (int) m;            // nhit / ml / mr share score / mxscr / prevj
                w.prevj = (int) m;
                ++w.score;
                continue;
            }
            const int intvl = (int) m - w.prevj - L.width;
            if (intvl > 0) {
                const int land = w.mxscr - L.cutoff;
                w.score -= L.gain * intvl;
                if (land > w.score || w.score < 0) {
                    if (land > 0) enter(w, r);
                    w.score = tplwt;
                    if ((int) m < L.width) w.score += L.gain * (L.width - (int) m);
                    w.mxscr = w.score;
                    w.maxj = w.lastj = (int) m;
                } else w.score += tplwt;
            } else if ((int) m - w.lastj == 1) w.score += L.gain1;
            else w.score += L.gain;
            if (w.score > w.mxscr) { w.mxscr = w.score; w.maxj = (int) m; }
            w.prevj = (int) m;
        }
    }
    void close_diag(Jxtd& w, int r)
    {
        if (dhit) { if (w.score >= M->min_hit) storedh(r, w.mxscr, w.prevj); }
        else if (w.mxscr > precutoff) {
            const int d = w.maxj + 2 * L.width - mm;
            if (d > 0) w.mxscr += L.gain * d;
            if (w.mxscr > L.cutoff) enter(w, r);
        }
    }
    void dmsnno()
    {
        const Jxtd ixtd = {0, 0, -(L.width + 1), 0, 0};
        const int nn = P->b_right - P->b_left - awspan;
        int bs = P->b_left;
        const int ts = P->b_left + awspan;
        jxtd.assign(mm, ixtd);
        while (bs < ts) {
            const uint32_t c = conv(P->b[bs++]);
            if (bpp.good(c)) bpp.word(c); else bpp.flaw();
        }
        for (int n = 0; n < nn; ) {
            const uint32_t c = conv(P->b[bs++]);
            if (bpp.good(c)) {
                const uint32_t w = bpp.word(c);
                const uint32_t m = bpp.flawless() ? header[w] : 0;
                if (m) scan_b(m, (uint32_t) n);
            } else bpp.flaw();
            const int r = ++n - mm;
            Jxtd& w = jxtd[n % mm];
            close_diag(w, r);
            w = ixtd;
        }
        for (int r = nn - mm; r < nn; ++r) close_diag(jxtd[(r + mm) % mm], r);
    }
    void dmsnno31()
    {
        static const int next_p[3] = {1, 2, 0};
        const Jxtd ixtd = {0, 0, -(L.width + 1), 0, 0};
        const int nn = P->b_right - P->b_left - bwspan;
        int bs = P->b_left;
        const int ts = P->b_left + bwspan - 1;
        jxtd.assign(sect_l, ixtd);
        int p = 0;
        for ( ; bs < ts; p = next_p[p]) {
            const uint32_t c = conv(P->b[bs++]);
            if (bpp.good(c)) bpp.word(c, p); else bpp.flaw(p);
        }
        for (int n = 0; n < nn; p = next_p[p]) {
            const uint32_t c = conv(P->b[bs++]);
            if (bpp.good(c)) {
                const uint32_t w = bpp.word(c, p);
                const uint32_t m = bpp.flawless(p) ? header[w] : 0;
                if (m) scan_b(m, (uint32_t) n);
            } else bpp.flaw(p);
            const int r = ++n - sect_l;
            Jxtd& w = jxtd[n % sect_l];
            close_diag(w, r);
            w = ixtd;
        }
        for (int r = nn - sect_l; r < nn; ++r) close_diag(jxtd[(r + sect_l) % sect_l], r);
    }
    // the records of the scan + the closing {mm, nn, 0}; false: nothing found
    bool run_dmsnno(std::vector<Juxt>& jxt)
    {
        jxt.clear();
        if (ng || P->a_right - P->a_left < L.width || P->b_right - P->b_left < bbt * L.width) return false;
        mfd.clear();
        if (bbt == 1) { bpp.clear(); dmsnno(); }
        else { bpp.init(L.elem, 3, L.width, L.bitpat, L.bitpat_len); dmsnno31(); }
        if (mfd.empty()) return false;
        jxt = mfd;
        Juxt end = {P->a_right - P->a_left, P->b_right - P->b_left, 0, 0, 0};
        jxt.push_back(end);
        return true;
    }
    int eval(Juxt& j) const
    {
        const bool prot = bbt == 3;
        int scr = 0;
        int as = j.jx, bs = j.jy;
        const int at0 = as, bt0 = bs;
        bool has_bb = false; int bb = 0;        // the SGPT6 pointer (protein with an Exinon) as a position
        auto sg = [&](const int16_t* arr, int i) -> int { return (i < 0 || i > P->b_len + 2) ? 0 : arr[i]; };
        if (prot) {
            ++bs;
            if (P->sigS) {
                has_bb = true; bb = j.jy + 1;
                if (j.jx == 0 && P->a[as] == M->met && sg(P->sigS, bb) > 0) scr = L.vthr / 2;
            }
        }
        if (scr <= 0 && P->a_exgl) {
            const int lend = L.tpl - j.jx;
            if (lend > 0) scr += M->end_bonus * std::min(lend, L.tpl);
        }
        while (--as >= 0 && (bs -= bbt) >= 0) {
            if ((as < at0 || bs < bt0) && conv(P->a[as]) != conv(P->b[bs])) break;
            j.jx--; j.jy -= bbt;
            ++j.jlen;
            if (P->a[as] == P->b[bs] || (P->a[as] == M->ser && P->b[bs] == M->ser2)) ++j.nid;
            if (has_bb) bb -= bbt;
        }
        if (as < 0) bs -= bbt;
        const int at = std::min(j.jx + j.jlen, P->a_right);
        const int bt = std::min(j.jy + bbt * j.jlen, P->b_right);
        const int ax = P->a_len;                // (a->tlen)
        int bx = P->b_len;
        if (prot) --bx;
        j.jlen = j.nid = 0;
        int maxscr = scr;
        const int al = as;
        int start = 0, restart = 0, end = 0, nid = 0;
        while (++as < ax && (bs += bbt) < bx) {
            if ((as >= at || bs >= bt) && conv(P->a[as]) != conv(P->b[bs])) break;
            ++j.jlen;
            scr += sim2(P->a[as], P->b[bs]);
            if (P->a[as] == P->b[bs] || (P->a[as] == M->ser && P->b[bs] == M->ser2)) ++j.nid;
            if (has_bb) { scr += sg(P->sigE, bb); bb += bbt; }
            if (scr < 0) { scr = 0; restart = as - al; j.jlen = j.nid = 0; }       // Kadane-Gries
            if (scr > maxscr) { maxscr = scr; start = restart; end = j.jlen; nid = j.nid; }
        }
        j.jx += start;
        j.jy += bbt * start;
        j.jlen = end;
        j.nid = nid;
        const int nmmc = std::min(j.jlen - j.nid, 3);
        if (M->crs == 0 && prot && nmmc) scr -= nmmc * L.vthr;
        if (as == P->a_len && has_bb && sg(P->sigT, bb) > 0) scr += L.vthr / 2;
        else {
            const int rend = L.tpl - P->a_right + j.jx + j.jlen;
            if (P->a_exgr && rend > 0) scr += M->end_bonus * std::min(rend, L.tpl);
            else if (nid == end) scr += M->end_bonus * 4;
        }
        return scr;
    }
    // restores coordinates, re-scores, drops what stays at or below vthr; jxt[num] (the closing record) moves up
    void reeval(std::vector<Juxt>& jxt, int& num) const
    {
        int k = 0;
        for (int i = 0; i < num; ++i) {
            Juxt& w = jxt[i];
            w.jx += P->a_left;
            w.jy += P->b_left;
            w.jscr = eval(w);
            if (w.jscr > L.vthr) jxt[k++] = w;
        }
        if (k < num) { jxt[k] = jxt[num]; num = k; }
    }

    struct Hsp { int lx, ly, rx, ry, ux, rr, nid, len, irno, jscr, sscr, sumh, ulnk; };
    int link_hsp_scr(const Hsp& m, const Hsp& n) const
    {
        int dr = n.rr - m.rr;
        const int dd = std::min(n.lx - m.rx, n.ly - m.ry);
        int scr = NEVSEL_;
        if (dr < 0) dr = -dr;
        else if (dr && M->lsg) {
            if ((M->hard_maxl && dr > M->maxl) || (M->hard_minl && dr < M->minl)) return scr;
            scr = intpen_plus(dr);
        }
        dr /= bbt;
        const int pen = gap_penalty(dr);
        if (pen > scr) scr = pen;
        if (dd < 0) scr += (m.jscr + n.jscr) * dd / (m.len + n.len);
        return scr;
    }
    std::vector<Hsp> mkhsps(const std::vector<Juxt>& jxt, int n) const
    {
        std::vector<Hsp> hsp(std::max(n, 1));
        int wcl = 0, pcl = 0;
        for (int i = 0; i < n; ++i) {
            const Juxt& j = jxt[i];
            Hsp& w = hsp[wcl];
            w.lx = j.jx; w.ly = j.jy; w.rx = j.jx + j.jlen; w.ry = j.jy + bbt * j.jlen; w.rr = j.jy - bbt * j.jx;
            w.nid = j.nid; w.len = j.jlen; w.jscr = j.jscr; w.sscr = 0; w.ulnk = -1; w.irno = 0; w.sumh = 0; w.ux = INT_MAX;
            const Hsp& p = hsp[pcl];
            if (wcl == 0 || w.ly > p.ry || w.rx > p.rx || w.rx < p.lx) { pcl = wcl++; continue; }
            const int aovr = w.rx - std::max(w.lx, p.lx);
            const Hsp& mcl = ((int64_t) w.jscr * p.len > (int64_t) p.jscr * w.len) ? w : p;
            const int ovrscr = mcl.jscr * aovr / mcl.len + gap_penalty(std::abs(p.rr - w.rr));
            if (ovrscr > 0) pcl = wcl++;
            else if (w.jscr > p.jscr) hsp[pcl] = w;
        }
        hsp.resize(wcl);
        return hsp;
    }
    // jxtcore: chains of HSPs -> units, their bounds, sorted by score
    void jxtcore(std::vector<Juxt>& jxt, int num, std::vector<Unit>& out) const
    {
        std::stable_sort(jxt.begin(), jxt.begin() + num, [](const Juxt& a, const Juxt& b) {
            const int dr = a.jx + a.jy - b.jx - b.jy;
            return dr ? dr < 0 : a.jy < b.jy; });
        std::vector<Hsp> ccl = mkhsps(jxt, num);
        num = (int) ccl.size();
        std::vector<int> phcl(num + 1, -1);
        int irno = 0, sumh = 0;
        for (int n = 0; n < num; ++n) {
            Hsp& ncl = ccl[n];
            ncl.ulnk = -1;
            int sscr = 0, q = n;
            for (int m = n; --m >= 0; ) {
                const Hsp& mcl = ccl[m];
                if (ncl.rx <= mcl.rx || ncl.ry < mcl.ry || ncl.lx <= mcl.lx || mcl.ux <= ncl.lx ||
                    (mcl.rx - ncl.lx) * 2 > ncl.rx - mcl.lx) continue;
                const int h = mcl.sscr + link_hsp_scr(mcl, ncl);
                if (h > sscr) { sscr = h; q = m; }
            }
            ncl.sscr = sscr += ncl.jscr;
            if (q != n) {
                Hsp& qcl = ccl[q];
                ncl.ulnk = q;
                ncl.irno = qcl.irno;
                if (ncl.sscr > ccl[phcl[qcl.irno]].sscr) phcl[qcl.irno] = n;
                sumh = sscr + (qcl.sumh - qcl.sscr);
                if (qcl.ux > ncl.rx) qcl.ux = ncl.rx;
            } else {
                phcl[irno] = n;
                ncl.irno = irno++;
                sumh += ncl.jscr;
            }
            ncl.sumh = sumh;
        }
        std::stable_sort(phcl.begin(), phcl.begin() + irno, [&](int x, int y) { return ccl[y].sscr - ccl[x].sscr < 0; });
        phcl[irno] = -1;
        int wh = 0;
        int q = phcl[wh++];
        const int maxh = (!M->lsg && M->mlt < 2) ? ccl[q].sscr - L.vthr : L.vthr;
        std::vector<Unit> units;
        while (q >= 0 && ccl[q].sscr >= maxh) {
            const int head = q;
            int cnt = 0;
            for ( ; q >= 0 && ccl[q].sscr > 0; q = ccl[q].ulnk) ++cnt;
            if (q >= 0) {                       // partial overlap with a chain already taken
                for (q = head; q >= 0 && ccl[q].sscr > 0; q = ccl[q].ulnk) ccl[q].sscr = 0;
            } else {
                Unit u;
                u.num = cnt; u.scr = ccl[head].sscr; u.nid = u.tlen = 0; u.llmt = u.ulmt = 0;
                u.jxt.resize(cnt + 1);
                u.jxt[cnt] = Juxt{P->a_right, P->b_right, 0, 0, 0};
                // (the reference leaves nid of the closing record as the array held it: fresh memory of `new JUXT[]`;
                //  nothing reads it)
                int at = cnt;
                for (q = head; q >= 0; q = ccl[q].ulnk) {
                    Hsp& c = ccl[q];
                    u.jxt[--at] = Juxt{c.lx, c.ly, c.len, c.nid, c.jscr};
                    u.nid += c.nid; u.tlen += c.len;
                    c.sscr = 0;
                }
                units.push_back(u);
            }
            q = phcl[wh++];
        }
        int n_u = (int) units.size();
        for (Unit& u : units) { u.llmt = u.jxt[0].jy; const Juxt& r = u.jxt[u.num - 1]; u.ulmt = r.jy + bbt * r.jlen; }
        std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) {
            const int d = a.llmt - b.llmt;
            return d ? d < 0 : a.ulmt < b.ulmt; });
        int llmt = P->b_left;
        for (int l = 0; l < n_u; ++l) {
            Unit& wl = units[l];
            if (!wl.num) continue;
            wl.llmt = llmt;
            for (int u = l + 1; u < n_u; ++u) {
                Unit& wu = units[u];
                if (wl.ulmt < wu.llmt) { llmt = wl.ulmt; wl.ulmt = wu.llmt; break; }
                int jl = 0;
                const int jr = wu.num;
                while (++jl < jr) if (wl.ulmt < wu.jxt[jl].jy) { wl.ulmt = wu.jxt[jl].jy; break; }
                if (jl == jr) {
                    if (wl.scr >= wu.scr) { wu.num = 0; continue; }
                    wl.num = 0;
                }
                for (int k = wl.num; --k >= 0; ) {
                    const int uu = wl.jxt[k].jx + bbt * wl.jxt[k].jlen;
                    if (uu < wu.llmt) { llmt = uu; break; }
                }
                break;
            }
        }
        int wlum = 0, wlur = n_u;
        for (int l = 0; l < wlur; ) {
            if (units[l].num) { if (units[l].ulmt > units[wlum].ulmt) wlum = l; ++l; }
            else std::swap(units[l], units[--wlur]);
        }
        units.resize(wlur);
        if (!units.empty()) units[wlum].ulmt = P->b_right;
        std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) {
            if (a.scr == b.scr) return b.nid - a.nid < 0;
            return a.scr > b.scr; });
        out.swap(units);
    }
};

// Wilip::Wilip(seqs, pwd, level): the units, best first; empty: none
inline void run(const SpdpWilipModel* m, const Pair* p, int level, std::vector<Unit>& units)
{
    units.clear();
    Wlp w(m, p, level);
    if (w.ng) return;
    std::vector<Juxt> jxt;
    if (!w.run_dmsnno(jxt)) return;
    int n = (int) jxt.size() - 1;
    w.reeval(jxt, n);
    if (!n) return;
    if (n == 1) {
        Unit u;
        u.num = 1; u.scr = jxt[0].jscr; u.nid = jxt[0].nid; u.tlen = jxt[0].jlen; u.llmt = p->b_left; u.ulmt = p->b_right;
        u.jxt.assign(jxt.begin(), jxt.begin() + 2);
        units.push_back(u);
    } else w.jxtcore(jxt, n, units);
}

// the flat form SpdpHspSource::units hands over: n_units, then per unit {num, nid, tlen, llmt, ulmt, scr} + (num + 1) x {jx, jy, jlen, nid, jscr}
inline void flatten(const std::vector<Unit>& units, std::vector<int32_t>& flat)
{
    flat.clear();
    flat.push_back((int32_t) units.size());
    for (const Unit& u : units) {
        const int32_t hd[6] = {u.num, u.nid, u.tlen, u.llmt, u.ulmt, u.scr};
        flat.insert(flat.end(), hd, hd + 6);
        for (int j = 0; j <= u.num; ++j) {
            const Juxt& t = u.jxt[j];
            const int32_t r[5] = {t.jx, t.jy, t.jlen, t.nid, t.jscr};
            flat.insert(flat.end(), r, r + 5);
        }
    }
}

}   // namespace spdp_wl
#endif
