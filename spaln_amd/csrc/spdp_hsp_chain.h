// spdp_hsp_chain.h -- from the HSPs of one search to gene candidates: what the reference's Wilip does after its scan
// (ogotoh/spaln v3.0.7 src/wln.cc: mkhsps :680-726, LinkHspScr :390-405 of the restated numbering = :644-678, jxtcore :776-926,
// Wilip::Wilip :955-992).  Host code; its input is a list of scored HSPs -- from the device search (spdp_hsp.hip) or from the
// host's form of it (spdp_hsp_host.h) -- its output the `units` FindHsp and the seeded walks read: chains of HSPs, best first,
// each with the stretch of the genomic side it may claim.
//
// Four steps, each a plain pass over small arrays (tens of HSPs):
//   order   by anti-diagonal, then genomic start (ties keep the order the scan produced: same diagonal, rising position)
//   thin    an HSP that ends inside its predecessor's shadow either stands beside it (their overlap still pays) or the better
//           of the two survives
//   link    every HSP takes the predecessor that maximises chain score + link cost (gap or intron cost, overlap correction);
//           a predecessor whose successor ends at x is closed for later HSPs that start at or behind x
//   harvest families of chains by their best tip; a tip whose chain runs into an already harvested HSP is dropped; the chains'
//           genomic claims are fenced against each other, contained chains lose to better ones; best score first
#ifndef SPDP_HSP_CHAIN_H_
#define SPDP_HSP_CHAIN_H_

#include <stdint.h>
#include <limits.h>
#include <algorithm>
#include <vector>
#include "../../include/spdp.h"

namespace spdp_hsp {

struct Hsp { int jx, jy, jlen, nid, jscr; };            // SpdpJuxt's layout (JUXT, src/seq.h:174)
struct Unit { int num, nid, tlen, llmt, ulmt, scr; std::vector<Hsp> hsp; };     // WLUNIT and its num + 1 records (the last one closes the list)

struct ChainCost {                                      // what the link cost reads
    const SpdpWilipModel* M;
    const int16_t* intpen; int intpen_len;              // IntronPenalty::Penalty(len)
    int gop, gep, lgop, lgep, codonk1;                  // PwdB::GapPenalty
    int bbt, vthr;                                      // 3: protein query; the level's threshold (scaled where the search scaled it)
    int gap(int n) const { return n == 0 ? 0 : (n > codonk1 ? lgop + n * lgep : gop + n * gep); }
    int intron(int n) const                             // IntronPenalty::PenaltyPlus (src/codepot.h:248)
    {
        if (n < M->llmt) return SHRT_MIN;
        return (int) (int16_t) (intpen[std::min(n, intpen_len - 1)] + M->avrsig);
    }
};

namespace chain_detail {

constexpr int NEVER = INT32_MIN / 16 * 7;               // NEVSEL, src/cmn.h:79

struct Node {
    int lx, ly, rx, ry, diag, nid, len, scr;            // the HSP: [lx, rx) x [ly, ry) on diagonal `diag`
    int open_to = INT_MAX;                              // successors must start before this query position
    int chain = 0, back = -1, family = 0;               // best chain score ending here, its predecessor, the family it belongs to
};

inline Node node_of(const Hsp& h, int bbt)
{
    Node n;
    n.lx = h.jx; n.ly = h.jy; n.rx = h.jx + h.jlen; n.ry = h.jy + bbt * h.jlen; n.diag = h.jy - bbt * h.jx;
    n.nid = h.nid; n.len = h.jlen; n.scr = h.jscr;
    return n;
}

inline std::vector<Node> thin(const std::vector<Hsp>& sorted, const ChainCost& C)
{
    std::vector<Node> keep;
    size_t shadow = 0;                                  // the kept HSP later ones are held against
    for (const Hsp& h : sorted) {
        const Node w = node_of(h, C.bbt);
        if (keep.empty()) { keep.push_back(w); shadow = 0; continue; }
        const Node& p = keep[shadow];
        const bool clear_of_it = w.ly > p.ry || w.rx > p.rx || w.rx < p.lx;
        if (clear_of_it) { shadow = keep.size(); keep.push_back(w); continue; }
        // w ends inside p's query span: what is their overlap worth at the denser one's rate, less the shift between the diagonals?
        const int overlap = w.rx - std::max(w.lx, p.lx);
        const Node& denser = ((int64_t) w.scr * p.len > (int64_t) p.scr * w.len) ? w : p;
        const int worth = denser.scr * overlap / denser.len + C.gap(std::abs(p.diag - w.diag));
        if (worth > 0) { shadow = keep.size(); keep.push_back(w); }
        else if (w.scr > p.scr) keep[shadow] = w;
    }
    return keep;
}

// cost of going from m to n (m before n)
inline int link_cost(const Node& m, const Node& n, const ChainCost& C)
{
    int shift = n.diag - m.diag, cost = NEVER;
    if (shift < 0) shift = -shift;
    else if (shift && C.M->lsg) {                       // the genomic side runs ahead: an intron, if its length is allowed
        if ((C.M->hard_maxl && shift > C.M->maxl) || (C.M->hard_minl && shift < C.M->minl)) return cost;
        cost = C.intron(shift);
    }
    cost = std::max(cost, C.gap(shift / C.bbt));
    const int apart = std::min(n.lx - m.rx, n.ly - m.ry);
    if (apart < 0) cost += (m.scr + n.scr) * apart / (m.len + n.len);       // they overlap: that part was counted twice
    return cost;
}

inline bool may_follow(const Node& m, const Node& n)
{
    if (n.rx <= m.rx || n.ry < m.ry || n.lx <= m.lx) return false;          // not further along
    if (m.open_to <= n.lx) return false;                                    // m already has a successor that starts earlier
    return (m.rx - n.lx) * 2 <= n.rx - m.lx;                                // overlap at most half of their joint span
}

}   // namespace chain_detail

// hsps: in scan order (same diagonal: rising position).  a_left .. b_right: the ranges of the search.
inline void chain(std::vector<Hsp> hsps, const ChainCost& C, int a_left, int a_right, int b_left, int b_right, std::vector<Unit>& units)
{
    using namespace chain_detail;
    units.clear();
    if (hsps.empty()) return;
    if (hsps.size() == 1) {                             // a single HSP is its own unit and claims the whole range
        Unit u;
        u.num = 1; u.scr = hsps[0].jscr; u.nid = hsps[0].nid; u.tlen = hsps[0].jlen; u.llmt = b_left; u.ulmt = b_right;
        u.hsp = {hsps[0], Hsp{a_right - a_left, b_right - b_left, 0, 0, 0}};        // (here the reference closes the list with the LENGTHS of the ranges)
        units.push_back(u);
        return;
    }
    std::stable_sort(hsps.begin(), hsps.end(), [](const Hsp& x, const Hsp& y) {
        const int dx = x.jx + x.jy, dy = y.jx + y.jy;
        return dx != dy ? dx < dy : x.jy < y.jy; });
    std::vector<Node> node = thin(hsps, C);
    const int n_nodes = (int) node.size();
    // ---- link
    std::vector<int> tip;                               // per family: its best-scoring member so far
    for (int n = 0; n < n_nodes; ++n) {
        Node& cur = node[n];
        int best = 0, from = -1;
        for (int m = n - 1; m >= 0; --m) {
            if (!may_follow(node[m], cur)) continue;
            const int h = node[m].chain + link_cost(node[m], cur, C);
            if (h > best) { best = h; from = m; }
        }
        cur.chain = best + cur.scr;
        cur.back = from;
        if (from < 0) { cur.family = (int) tip.size(); tip.push_back(n); continue; }
        Node& pre = node[from];
        cur.family = pre.family;
        if (cur.chain > node[tip[pre.family]].chain) tip[pre.family] = n;
        pre.open_to = std::min(pre.open_to, cur.rx);
    }
    // ---- harvest
    std::stable_sort(tip.begin(), tip.end(), [&](int x, int y) { return node[x].chain > node[y].chain; });
    const int floor_ = (!C.M->lsg && C.M->mlt < 2) ? node[tip[0]].chain - C.vthr : C.vthr;
    for (size_t f = 0; f < tip.size() && node[tip[f]].chain >= floor_; ++f) {
        const int head = tip[f];
        int q = head, count = 0;
        for ( ; q >= 0 && node[q].chain > 0; q = node[q].back) ++count;
        if (q >= 0) {                                   // it runs into an HSP a better chain has taken: the rest is spent as well
            for (q = head; q >= 0 && node[q].chain > 0; q = node[q].back) node[q].chain = 0;
            continue;
        }
        Unit u;
        u.num = count; u.scr = node[head].chain; u.nid = u.tlen = 0; u.llmt = u.ulmt = 0;
        u.hsp.assign((size_t) count + 1, Hsp{a_right, b_right, 0, 0, 0});      // (the last one closes the list)
        int at = count;
        for (q = head; q >= 0; q = node[q].back) {
            Node& c = node[q];
            u.hsp[--at] = Hsp{c.lx, c.ly, c.len, c.nid, c.scr};
            u.nid += c.nid; u.tlen += c.len;
            c.chain = 0;                                // taken
        }
        units.push_back(u);
    }
    // ---- fences: every unit's claim on the genomic side, by position
    for (Unit& u : units) { u.llmt = u.hsp[0].jy; const Hsp& r = u.hsp[u.num - 1]; u.ulmt = r.jy + C.bbt * r.jlen; }
    std::stable_sort(units.begin(), units.end(), [](const Unit& x, const Unit& y) { return x.llmt != y.llmt ? x.llmt < y.llmt : x.ulmt < y.ulmt; });
    const int n_u = (int) units.size();
    int fence = b_left;
    for (int l = 0; l < n_u; ++l) {
        Unit& lo = units[l];
        if (!lo.num) continue;
        lo.llmt = fence;
        for (int k = l + 1; k < n_u; ++k) {
            Unit& hi = units[k];
            if (lo.ulmt < hi.llmt) { fence = lo.ulmt; lo.ulmt = hi.llmt; break; }      // apart: lo may reach up to hi's first HSP
            // they overlap: lo may reach up to the first HSP of hi that starts behind lo's end (the first one apart)
            int j = 1;
            for ( ; j < hi.num; ++j) if (lo.ulmt < hi.hsp[j].jy) { lo.ulmt = hi.hsp[j].jy; break; }
            if (j == hi.num) {                          // none: one of them lies inside the other; the better one stays
                if (lo.scr >= hi.scr) { hi.num = 0; continue; }
                lo.num = 0;
            }
            for (int i = lo.num - 1; i >= 0; --i) {
                const int reach = lo.hsp[i].jx + C.bbt * lo.hsp[i].jlen;      // (query start + genomic length: the reference's own mix)
                if (reach < hi.llmt) { fence = reach; break; }
            }
            break;
        }
    }
    // the dropped ones go out (the reference swaps them to the end: the order of the others changes with it), the unit that
    // reaches furthest claims the rest of the range
    int last = n_u, furthest = 0;
    for (int l = 0; l < last; ) {
        if (units[l].num) { if (units[l].ulmt > units[furthest].ulmt) furthest = l; ++l; }
        else std::swap(units[l], units[--last]);
    }
    units.resize(last);
    if (!units.empty()) units[furthest].ulmt = b_right;
    std::stable_sort(units.begin(), units.end(), [](const Unit& x, const Unit& y) { return x.scr != y.scr ? x.scr > y.scr : x.nid > y.nid; });
}

// the flat form SpdpHspSource::units hands over: n_units, then per unit {num, nid, tlen, llmt, ulmt, scr} + (num + 1) x {jx, jy, jlen, nid, jscr}
inline void flatten(const std::vector<Unit>& units, std::vector<int32_t>& flat)
{
    flat.assign(1, (int32_t) units.size());
    for (const Unit& u : units) {
        flat.insert(flat.end(), {u.num, u.nid, u.tlen, u.llmt, u.ulmt, u.scr});
        for (int j = 0; j <= u.num; ++j) flat.insert(flat.end(), {u.hsp[j].jx, u.hsp[j].jy, u.hsp[j].jlen, u.hsp[j].nid, u.hsp[j].jscr});
    }
}

}   // namespace spdp_hsp
#endif
