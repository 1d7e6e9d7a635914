// spdp_host.cpp -- the reference's dispatch around the DP engines, host side.
//
// What it mirrors (ogotoh/spaln v3.0.7):
//   alignS_ng (ori = 1, seeding off)        src/fwd2s1.cc:2746-2760
//   Aln2s1::globalS_ng                      src/fwd2s1.cc:2674-2694
//   Aln2s1::lspS_ng   (decision ladder)     src/fwd2s1.cc:1801-1897
//   Aln2s1::trcbkalignS_ng                  src/fwd2s1.cc:1667-1710
//   Aln2s1::mimd_postwork / rcsv_postwork   src/fwd2s1.cc:1714-1799
//   Aln2s1::diagonalS_ng                    src/fwd2s1.cc:1629-1665
//   stdskl / trimskl                        src/gaps.cc:140-180, 254-273
//
// The reference runs this ladder synchronously per query, deep inside its
// worker threads.  Here the same decisions are taken for a whole batch in
// rounds: every pending sub-problem is classified, all UDH sweeps of the round
// go to the GPU in one launch, their cpos rows are turned into slabs (or, in
// recursive mode, into two half problems for the next round), and finally all
// traceback sub-problems of all queries run in one forward launch + one walk.
// Sub-problems are descriptors into the resident inputs (no re-upload).

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>

#include "spdp_internal.h"

namespace {

const int kEndOfUlk = SPDP_END_OF_ULK;
const float kCoefB = 2.0f;                  // sizeof(short), src/fwd2s1.cc:58
const int kScalarRows = 8;                  // trcbkalignS_ng: m < 8 goes to the scalar engine

struct Rng { int al, ar, bl, br; uint8_t a_exgl, a_exgr, b_exgl, b_exgr; };

struct Job {                                // one query through alignS_ng
    int score = SPDP_NEVSEL;
    bool score_set = false;
    bool failed = false;
    bool edge = false;                      // a linear-space call of this query ran along the free left edge (SPDP_ALN_LEFT_EDGE)
    std::vector<SpdpSkl> rec;               // Mfile records after the dummy one
};

struct LspItem { int job; Rng r; SpdpWindow w; bool top; };
struct TbItem  { int job; Rng r; SpdpWindow w; bool top; int cut_l = 0, cut_r = 0; };     // trcbkalignS_ng call (cut_r > cut_l: with a cut range)
struct UdhItem { int job; Rng r; SpdpWindow w; bool top; int n_imd; bool recursive; int imd_intvl; };

void stripe_of(const Rng& r, int sh, SpdpWindow* w)
{
    SpdpProblem p;
    memset(&p, 0, sizeof p);
    p.a_left = r.al; p.a_right = r.ar; p.b_left = r.bl; p.b_right = r.br;
    spdp_stripe(&p, sh, w);
}

RunItem run_item(int parent, const Rng& r, const SpdpWindow& w, int n_im)
{
    RunItem it;
    it.parent = parent;
    it.a_left = r.al; it.a_right = r.ar; it.b_left = r.bl; it.b_right = r.br;
    it.a_exgl = r.a_exgl; it.a_exgr = r.a_exgr; it.b_exgl = r.b_exgl; it.b_exgr = r.b_exgr;
    it.w = w; it.n_im = n_im;
    return it;
}

// PwdB::GapPenalty / GapExtPen / UnpPenalty with codonk1 = LARGEN (Noll = 2), src/aln.h:275-287
int gap_penalty(const SpdpScoring& sc, int i) { return i == 0 ? 0 : sc.gop + i * sc.gep; }
int gap_ext_pen(const SpdpScoring& sc, int) { return sc.gep; }
int unp_penalty(const SpdpScoring& sc, int d) { return d * sc.gep; }

// Aln2s1::diagonalS_ng (non-local and local), src/fwd2s1.cc:1629-1665
int diagonal_s(const SpdpScoring& sc, const SpdpProblem& p, const Rng& r, std::vector<SpdpSkl>& rec)
{
    const bool Local = sc.local;
    const bool LocalL = Local && r.a_exgl && r.b_exgl;
    const bool LocalR = Local && r.a_exgr && r.b_exgr;
    const int dlt = Local ? 0 : ((r.br - r.bl) - (r.ar - r.al));
    // if dlt < 0 the roles of a and b are swapped for the walk
    const uint8_t* as = dlt < 0 ? p.b : p.a;
    const uint8_t* bs = dlt < 0 ? p.a : p.b;
    const int al = dlt < 0 ? r.bl : r.al, ar = dlt < 0 ? r.br : r.ar;
    const int bl = dlt < 0 ? r.al : r.bl;
    int mL = al, mR = ar;
    int scr = 0, maxh = SPDP_NEVSEL;
    for (int m = al; m++ < ar; ) {
        const int x = as[m - 1], y = bs[bl + (m - 1 - al)];
        scr += dlt < 0 ? sc.mtx[y * sc.mtx_dim + x] : sc.mtx[x * sc.mtx_dim + y];
        if (LocalL && scr < 0) { scr = 0; mL = m; }
        if (LocalR && scr > maxh) { maxh = scr; mR = m; }
    }
    int rr = bl - al;
    if (dlt < 0) rr -= dlt;
    // NB: the reference writes (m, n) of the *swapped* pair; reproduce literally
    rec.push_back({mL, mL + rr});
    rec.push_back({mR, mR + rr});
    return LocalR ? maxh : scr;
}

}   // namespace

// Corner list of a set of path records (what stdskl / stdskl3, src/gaps.cc:140-227, return; in / out without header).
// The records, sorted by (m, n), are the vertices of a monotone polyline.  Every step between two of them is a
// diagonal leg (when it advances in both sequences) followed by a gap leg (when the advances differ); a vertex is
// a corner when the leg leaving it does not continue the leg arriving -- plus, as in the reference, every vertex
// that starts a step without progress in m.  M_UNIT = 3 for protein rows against nucleotide columns: a gap leg that
// is not a whole number of codons rounds the row of its corner up (row gaps) or splits off the frame shift (column gaps).
template <int M_UNIT>
std::vector<SpdpSkl> corner_list(std::vector<SpdpSkl> pts)
{
    if (pts.size() < 2) return pts;
    std::sort(pts.begin(), pts.end(), [](const SpdpSkl& x, const SpdpSkl& y) { return x.m != y.m ? x.m < y.m : x.n < y.n; });
    std::vector<SpdpSkl> corners;
    corners.reserve(2 * pts.size() + 1);
    enum { DIAG = 0, NONE = 2 };                     // headings: 0 diagonal, +1 gap along n, -1 gap along m
    int heading = NONE;
    size_t at = 0;
    for (size_t nxt = 1; nxt < pts.size(); ++nxt) {
        const int adv_m = (pts[nxt].m - pts[at].m) * M_UNIT, adv_n = pts[nxt].n - pts[at].n;
        if (adv_n < 0 || (!adv_m && !adv_n)) continue;           // a step back or a repeat: not a vertex
        const int diag = std::min(adv_m, adv_n), slack = adv_n - adv_m;
        const int gap = (slack > 0) - (slack < 0);
        const bool two_legs = diag && gap;
        if ((two_legs ? (int) DIAG : gap) != heading || !adv_m) corners.push_back(pts[at]);
        if (two_legs) {
            SpdpSkl c;
            c.n = pts[at].n + diag;
            c.m = pts[at].m + (diag + ((M_UNIT > 1 && slack < 0 && slack % M_UNIT) ? M_UNIT - 1 : 0)) / M_UNIT;
            corners.push_back(c);
            if (M_UNIT > 1 && slack > 0 && slack % M_UNIT) { c.n += slack % M_UNIT; corners.push_back(c); }
        }
        heading = gap;
        at = nxt;
    }
    corners.push_back(pts[at]);
    return corners;
}
template std::vector<SpdpSkl> corner_list<1>(std::vector<SpdpSkl>);
template std::vector<SpdpSkl> corner_list<3>(std::vector<SpdpSkl>);

// host-only entry (no device work): the corner list of n records, out[] has room for 2 n + 1; returns the count
extern "C" int spdp_corner_list(const SpdpSkl* recs, int n, int m_unit, SpdpSkl* out)
{
    if (!recs || !out || n < 0 || (m_unit != 1 && m_unit != 3)) return -1;
    std::vector<SpdpSkl> v(recs, recs + n);
    v = m_unit == 1 ? corner_list<1>(std::move(v)) : corner_list<3>(std::move(v));
    std::copy(v.begin(), v.end(), out);
    return (int) v.size();
}

namespace {

// trimskl, src/gaps.cc:254-273: delete terminal gaps at free ends
void trim_skl(std::vector<SpdpSkl>& s, const SpdpProblem& p)
{
    if (s.size() >= 2) {
        const int i = s[1].m - s[0].m, j = s[1].n - s[0].n;
        if ((p.a_exgl && !i) || (p.b_exgl && !j)) s.erase(s.begin());
    }
    if (s.size() >= 2) {
        const size_t k = s.size() - 1;
        const int i = s[k].m - s[k - 1].m, j = s[k].n - s[k - 1].n;
        if ((p.a_exgr && !i) || (p.b_exgr && !j)) s.pop_back();
    }
}

}   // namespace
void trim_skl_of(std::vector<SpdpSkl>& s, const SpdpProblem& p) { trim_skl(s, p); }
namespace {

// Chunks of one batch run as a software pipeline on lanes of the context (own streams and pools): chunk c starts
// its first linear-space sweep when that of chunk c - 1 has finished, so the host work, the slab tracebacks and the
// walk of one chunk run beside the big sweep of the next instead of leaving the GPU idle between launches.
struct ChunkGate {
    std::mutex m;
    std::condition_variable cv;
    int state = 0;                              // 0: not there yet, 1: event recorded, 2: nothing to wait for
    hipEvent_t ev = nullptr;
    void open(int st_) { { std::lock_guard<std::mutex> g(m); if (state == 0) state = st_; } cv.notify_all(); }
    int wait() { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return state != 0; }); return state; }
};

struct Aligner {
    SpdpContext* ctx;                           // the lane this chunk runs on
    const DevStore* st;
    const SpdpProblem* probs;                   // first problem of the chunk
    int n;
    int base = 0;                               // its index in the store (parent of job j = base + j)
    ChunkGate* gate_in = nullptr;               // the chunk before mine / mine
    ChunkGate* gate_out = nullptr;
    SpdpAlignment* out = nullptr;               // where the chunk's alignments go (may be null)
    bool raw = false;                           // lspS_ng level: hand the Mfile records over as they are
    const SpdpRequests* req = nullptr;          // explicit engine calls on sub-ranges of store entries (the seeded path)
    int parent(int job) const { return req ? req->parents[base + job] : base + job; }
    std::vector<TbItem> ctbs;                   // forwardS_ng calls with a cut range
    std::vector<Job> jobs;
    std::vector<LspItem> pending;
    std::vector<TbItem> tbs;                    // forwardS1_wip calls
    std::vector<TbItem> stbs;                   // scalar forwardS_ng calls (< 8 rows)
    std::vector<TbItem> xtbs;                   // -A1: forwardS1 calls
    float kernel_ms = 0.f;
    int64_t kernel_cells = 0;
    int unsupported = 0;
    int overflowed = 0;                         // queries given up because a record list did not fit (scalar / -A1 walkers)
    double stats[SPDP_N_STATS] = {0};           // see include/spdp.h

    void set_score(int job, bool top, int scr) { if (top) { jobs[job].score = scr; jobs[job].score_set = true; } }

    // Aln2s1::trcbkalignS_ng
    // a sub-range the links produced that does not lie inside the sequences: the reference would run
    // its engine on it anyway (out-of-bounds reads); here the query is reported as not aligned
    bool bad_range(int job, const Rng& r) const
    {
        return r.al < 0 || r.bl < 0 || r.ar > probs[job].a_len || r.br > probs[job].b_len || r.ar < r.al || r.br < r.bl;
    }

    void trcbk(int job, const Rng& r, const SpdpWindow& w, bool top)
    {
        if (w.width < 0) { set_score(job, top, SPDP_NEVSEL); return; }
        if (bad_range(job, r) || w.width < 3) { ++unsupported; jobs[job].failed = true; return; }
        if (st->sc.scalar_engines == 1 || r.ar - r.al < kScalarRows) {   // -A0, or fewer than 8 rows: scalar forwardS_ng (src/fwd2s1.cc:1677)
            if (!st->has_exact) { ++unsupported; jobs[job].failed = true; return; }
            stbs.push_back({job, r, w, top});
            return;
        }
        if (st->sc.scalar_engines == 2) {       // -A1: forwardS1 (mode 3 / 5) + Vmf::traceback
            if (!st->has_exact) { ++unsupported; jobs[job].failed = true; return; }
            xtbs.push_back({job, r, w, top});
            return;
        }
        tbs.push_back({job, r, w, top});
    }

    // Aln2s1::lspS_ng up to the point where an engine is needed
    bool classify(const LspItem& it, std::vector<UdhItem>& udh)
    {
        const SpdpScoring& sc = st->sc;
        const Rng& r = it.r;
        Job& J = jobs[it.job];
        const int m = r.ar - r.al, n = r.br - r.bl;
        if (!m && !n) { set_score(it.job, it.top, 0); return true; }
        if (!m || !n) {
            J.rec.push_back({r.al, r.bl});
            J.rec.push_back({r.ar, r.br});
            int scr;
            if (m) scr = (r.a_exgl || r.a_exgr) ? gap_ext_pen(sc, m) : gap_penalty(sc, m);
            else   scr = (r.b_exgl || r.b_exgr) ? gap_ext_pen(sc, n) : unp_penalty(sc, n);
            set_score(it.job, it.top, scr);
            return true;
        }
        if (it.w.up == it.w.lw) {
            set_score(it.job, it.top, diagonal_s(sc, probs[it.job], r, J.rec));
            return true;
        }
        if (abs(n - m) < 8 || m == 1 || n == 1) { trcbk(it.job, r, it.w, it.top); return true; }
        int n_imd = 1;
        bool recursive = sc.recursive != 0;     // algmode.alg & 4 (-A4 .. -A7)
        float cvol = float(m) * (n + m);        // rhombic, simd >= 2
        if (sc.scalar_engines >= 1) {           // hexagonal, simd < 2 (src/fwd2s1.cc:1830-1833)
            const float k = it.w.lw - r.bl + r.ar;
            const float q = r.br - r.al - it.w.up;
            cvol = float(m) * n - (k * k + q * q) / 2;
        }
        if (kCoefB * cvol < sc.max_vmf_space) { trcbk(it.job, r, it.w, it.top); return true; }
        int intvl = (m + 1) / 2;
        if (!recursive) {
            const float coef_C = (sc.noll + 1) * sizeof(int);
            const double z = 2. * m * kCoefB / coef_C;
            const int imd1 = int(pow(z, 1. / 3) + 0.5) - 1;
            const float spc = coef_C * n * imd1 + kCoefB * cvol / (imd1 + 1) / (imd1 + 1);
            if (spc > sc.max_vmf_space) recursive = true;
            else {
                const int imd3 = m / SPDP_NELEM;
                n_imd = sc.ubh ? sc.ubh : std::min(imd1, imd3);
                const int imd_intvl = intvl = (m + n_imd) / (n_imd + 1);
                if (imd_intvl * n_imd == m) --n_imd;
                if (n_imd == 0) { trcbk(it.job, r, it.w, it.top); return true; }
            }
        }
        if (bad_range(it.job, r) || it.w.width < 3) { ++unsupported; J.failed = true; return true; }
        if (sc.scalar_engines == 1 && !st->has_exact) { ++unsupported; J.failed = true; return true; }
        if (sc.scalar_engines == 2 && !st->has_exact) { ++unsupported; J.failed = true; return true; }
        udh.push_back({it.job, r, it.w, it.top, n_imd, recursive, intvl});
        return true;
    }

    // window of a slab: stripe() under SIMD; under -A0 the diagonal bounds hirschbergS_ng recorded in
    // the cpos row (src/fwd2s1.cc:1735-1740, 1778-1797)
    void slab_window(const Rng& r, int sh, const int32_t* row, SpdpWindow* w) const
    {
        if (st->sc.scalar_engines != 1) { stripe_of(r, sh, w); return; }
        w->lw = row[8]; w->up = row[9];
        w->width = w->up - w->lw + 3;
    }

    // Aln2s1::mimd_postwork on the written-back ranges
    void mimd(const UdhItem& u, const int32_t* cpos, Rng cur)
    {
        Job& J = jobs[u.job];
        const int sh = st->sc.sh;
        const int aleft = cur.al, bleft = cur.bl;
        cur.a_exgl = cur.a_exgr = cur.b_exgl = cur.b_exgr = 0;
#define CP(i, c) cpos[(i) * 10 + (c)]
        int i = u.n_imd;
        while (--i >= 0 && CP(i, 0) == kEndOfUlk) ;
        for ( ; i >= 0 && CP(i, 0) != kEndOfUlk; --i) {
            int c = 0;
            cur.al = CP(i, c);
            cur.b_exgl = CP(i, ++c) ? 1 : 0;
            cur.bl = CP(i, ++c);
            if (cur.bl < 0 || cur.bl > cur.br) break;
            while (c < 9 && CP(i, ++c) < kEndOfUlk) J.rec.push_back({cur.al, CP(i, c)});
            SpdpWindow vw;
            slab_window(cur, sh, &CP(i + 1, 0), &vw);
            trcbk(u.job, cur, vw, false);
            cur.ar = cur.al;
            cur.br = CP(i, c - 1);
        }
        if ((i < 0 && CP(0, 0) != kEndOfUlk) || CP(0, 2) != kEndOfUlk) {
            cur.al = aleft; cur.bl = bleft;
            SpdpWindow vw;
            slab_window(cur, sh, &CP(0, 0), &vw);
            trcbk(u.job, cur, vw, false);
        }
#undef CP
    }

    // Aln2s1::rcsv_postwork: two half problems for the next round
    void rcsv(const UdhItem& u, const int32_t* cpos, Rng cur)
    {
        Job& J = jobs[u.job];
        const int sh = st->sc.sh;
        cur.a_exgl = cur.a_exgr = cur.b_exgl = cur.b_exgr = 0;
        int c = 0;
        if (cpos[c++] < kEndOfUlk) {
            while (c < 9 && cpos[++c] < kEndOfUlk) J.rec.push_back({cpos[0], cpos[c]});
            Rng first = cur;
            first.ar = cpos[0]; first.br = cpos[c - 1];
            SpdpWindow w;
            slab_window(first, sh, cpos, &w);
            pending.push_back({u.job, first, w, false});
            Rng second = cur;
            second.al = cpos[0]; second.b_exgl = cpos[1] ? 1 : 0; second.bl = cpos[2];
            slab_window(second, sh, cpos + 10, &w);
            pending.push_back({u.job, second, w, false});
        } else if (st->sc.local) {
            SpdpWindow w;
            stripe_of(cur, sh, &w);
            trcbk(u.job, cur, w, false);
        }
    }

    double t_mark = 0;
    bool timing = getenv("SPDP_TIMING") != nullptr;
    void lap(const char* what)
    {
        if (!timing) return;
        const double now = std::chrono::duration<double, std::milli>(
            std::chrono::steady_clock::now().time_since_epoch()).count();
        if (t_mark > 0) fprintf(stderr, "[spdp timing] %-28s %8.2f ms\n", what, now - t_mark);
        t_mark = now;
    }

    int run()
    {
        const int rc = run_chunk();
        if (gate_out) gate_out->open(2);        // no linear-space round (or an error): the next chunk need not wait
        if (!rc && out) for (int i = 0; i < n; ++i) finish(i, out + i);     // stdskl / trimskl, beside the other chunks
        return rc;
    }

    int run_chunk()
    {
        const SpdpScoring& sc = st->sc;
        lap("start");
        jobs.assign(n, Job());
        for (int i = 0; i < n; ++i) {           // alignS_ng: stripe(alprm.sh), globalS_ng -> lspS_ng
            const SpdpProblem& p = probs[i];
            Rng r{p.a_left, p.a_right, p.b_left, p.b_right, p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr};
            SpdpWindow w;
            stripe_of(r, sc.sh, &w);
            if (req) {                          // a caller that made the reference's decisions up to the engine call itself
                w = req->windows[base + i];
                const int kind = req->kinds[base + i];
                if (kind == 1) { trcbk(i, r, w, true); continue; }
                if (kind == 2) {                // trcbkalignS_ng(wdw, spj, mc): always the scalar engine (src/fwd2s1.cc:1674-1678)
                    if (w.width < 0) { set_score(i, true, SPDP_NEVSEL); continue; }
                    if (bad_range(i, r) || !st->has_exact) { ++unsupported; jobs[i].failed = true; continue; }
                    TbItem t{i, r, w, true};
                    t.cut_l = req->cuts[2 * (base + i)]; t.cut_r = req->cuts[2 * (base + i) + 1];
                    ctbs.push_back(t);
                    continue;
                }
            }
            pending.push_back({i, r, w, true});
        }
        // Queries that go straight to the traceback do not wait for the linear-space rounds of the others: their
        // forward sweep starts on the side stream as soon as the first classification is done (an EST batch
        // has a handful of stragglers in the linear-space engine; that launch is latency-bound and would
        // otherwise hold up everything).  SPDP_OVERLAP=0 restores the single-stream order.
        DevRun side;
        side.use_ctx = ctx;
        std::vector<TbItem> side_tbs;
        const char* ov = getenv("SPDP_OVERLAP");
        bool may_overlap = ctx->stream2 != nullptr && !(ov && atoi(ov) == 0);
        bool first_round = true;
        while (!pending.empty()) {
            std::vector<LspItem> cur;
            cur.swap(pending);
            std::vector<UdhItem> udh;
            for (const LspItem& it : cur) classify(it, udh);
            lap("classify");
            if (may_overlap && !udh.empty() && tbs.size() >= 64) {
                side_tbs.swap(tbs);
                std::vector<RunItem> items;
                for (const TbItem& t : side_tbs) { items.push_back(run_item(parent(t.job), t.r, t.w, 0)); items.back().vmf_scale = 1 << 20; }   // (a few rows each)
                side.side = true;
                if (side.build(st, items, 1) || side.launch()) return -1;
                lap("side fwd build+launch");
            }
            may_overlap = false;                        // first round only
            if (udh.empty()) continue;
            std::vector<RunItem> items;
            for (const UdhItem& u : udh) {
                items.push_back(run_item(parent(u.job), u.r, u.w, u.n_imd));
                items.back().imd_intvl = u.imd_intvl;
            }
            DevRun run;
            run.use_ctx = ctx;
            run.beside = !side_tbs.empty();
            const bool a0 = sc.scalar_engines == 1;
            if (run.build(st, items, a0 ? 5 : (sc.scalar_engines == 2 ? 8 : 2))) return -1;
            lap("udh build");
            if (first_round && gate_in && gate_in->wait() == 1) HIPCHK(hipStreamWaitEvent(run.strm(), gate_in->ev, 0));
            if (run.launch()) return -1;
            if (first_round && gate_out) {
                HIPCHK(hipEventRecord(gate_out->ev, run.strm()));
                gate_out->open(1);
            }
            first_round = false;
            if (run.sync()) return -1;
            lap("udh launch+sync");
            kernel_ms += run.kernel_ms; kernel_cells += run.total_cells;
            stats[0] += run.kernel_ms; stats[1] += (double) run.total_cells; stats[2] += (double) items.size();
            stats[6] += 1;
            std::vector<int32_t> scores, cpos, ranges;
            std::vector<int32_t> edge;
            if (run.fetch_udh(scores, cpos, ranges, &edge)) return -1;
            std::vector<DevResult> ures;
            if (a0 && run.fetch_results(ures)) return -1;
            lap("udh fetch");
            const int stride = 10 * (run.max_n_im + 1);
            for (size_t k = 0; k < udh.size(); ++k) {
                const UdhItem& u = udh[k];
                const int scr = scores[k];
                if (a0 && ures[k].pad[0]) { ++unsupported; jobs[u.job].failed = true; continue; }   // undefined in the reference
                set_score(u.job, u.top, scr);
                if (edge[k]) jobs[u.job].edge = true;
                if (scr <= SPDP_NEVSEL) continue;
                Rng curr = u.r;                 // ranges as written back by the engine
                curr.al = ranges[4 * k]; curr.ar = ranges[4 * k + 1];
                curr.bl = ranges[4 * k + 2]; curr.br = ranges[4 * k + 3];
                const int32_t* cp = cpos.data() + k * stride;
                if (cp[0] == kEndOfUlk) {
                    jobs[u.job].rec.push_back({curr.al, curr.bl});
                    jobs[u.job].rec.push_back({curr.ar, curr.br});
                } else if (u.recursive) rcsv(u, cp, curr);
                else mimd(u, cp, curr);
            }
        }
        lap("postwork (slab lists)");
        // forwardS_ng (-A0, and any sub-problem below 8 rows) and forwardS1 (-A1) keep their traceback as Vmf records,
        // up to two per cell: a whole batch of slabs can ask for more memory than the card has, so they run in
        // groups whose record space stays below SPDP_VMF_GB (default 32) gigabytes
        auto run_vmf = [&](std::vector<TbItem>& list, int flav, const char* what) -> int {
            size_t limit = (size_t) 32 << 30;
            if (const char* e = getenv("SPDP_VMF_GB")) limit = (size_t) std::max(1, atoi(e)) << 30;
            auto recs_of = [](const TbItem& t, int scale) {
                const size_t rows = t.r.ar - t.r.al + 1, cols = t.r.br - t.r.bl + 1;
                const size_t band = rows * std::min<size_t>(cols, (size_t) t.w.width);
                return std::min(2 * rows * cols + 64, std::max(band / 2, 64 * rows) * (size_t) scale + 64);
            };
            std::vector<size_t> todo(list.size());
            for (size_t k = 0; k < list.size(); ++k) todo[k] = k;
            for (int scale = 1; !todo.empty(); scale *= 8) {
                std::vector<size_t> again;
                for (size_t lo = 0; lo < todo.size(); ) {
                    size_t hi = lo, sum = 0;
                    while (hi < todo.size() && (hi == lo || sum + recs_of(list[todo[hi]], scale) * sizeof(int3) <= limit))
                        sum += recs_of(list[todo[hi++]], scale) * sizeof(int3);
                    std::vector<RunItem> items;
                    for (size_t k = lo; k < hi; ++k) {
                        const TbItem& t = list[todo[k]];
                        items.push_back(run_item(parent(t.job), t.r, t.w, 0));
                        items.back().vmf_scale = scale;
                        items.back().cut_l = t.cut_l; items.back().cut_r = t.cut_r;
                    }
                    DevRun run;
                    run.use_ctx = ctx;
                    if (run.build(st, items, flav) || run.launch() || run.sync()) return -1;
                    kernel_ms += run.kernel_ms; kernel_cells += run.total_cells;
                    stats[3] += run.kernel_ms; stats[4] += (double) run.total_cells; stats[5] += (double) items.size();
                    std::vector<DevResult> res;
                    std::vector<int> nskl;
                    std::vector<int64_t> off;
                    std::vector<SpdpSkl> skl;
                    if (run.fetch_results(res) || run.fetch_skl(nskl, off, skl)) return -1;
                    for (size_t k = lo; k < hi; ++k) {
                        const TbItem& t = list[todo[k]];
                        const int c = nskl[k - lo];
                        if (c == -3 && scale < 4096) { again.push_back(todo[k]); continue; }     // outgrew its record budget: once more, larger
                        if (c == -1) { jobs[t.job].failed = true; ++overflowed; continue; }    // record list beyond its slot: this query only
                        if (c < 0) { ctx->err = what; return -1; }
                        set_score(t.job, t.top, res[k - lo].score);
                        const SpdpSkl* sk = skl.data() + off[k - lo];
                        jobs[t.job].rec.insert(jobs[t.job].rec.end(), sk, sk + c);
                    }
                    lo = hi;
                }
                todo.swap(again);
            }
            return 0;
        };
        if (!stbs.empty() && run_vmf(stbs, 3, "forwardS_ng traceback failed")) return -1;
        if (!ctbs.empty() && run_vmf(ctbs, 3, "forwardS_ng (cut range) traceback failed")) return -1;
        if (!xtbs.empty() && run_vmf(xtbs, 7, "forwardS1 traceback failed")) return -1;
        // all (remaining) trcbkalignS_ng calls of all queries: one forward sweep + one walk (beside the side run, if any)
        // A few tall slabs among many short ones (the recursion on a long cDNA leaves slabs of thousands of rows whose band
        // is narrow enough for the traceback): the launch would end with its tallest problem on the four waves of one
        // block.  They go to the side stream as a launch of their own -- all of them tall, so each gets a 16-wave block
        // (DevRun::build) -- beside the launch of the rest.  SPDP_SPLIT_FWD=0: one launch.
        if (side_tbs.empty() && ctx->stream2 != nullptr && tbs.size() > 1 && !(getenv("SPDP_SPLIT_FWD") && atoi(getenv("SPDP_SPLIT_FWD")) == 0)) {
            const int tall = 4 * 32 * SPDP_NELEM;               // >= 32 passes: DevRun::build's condition for 16-wave blocks
            std::vector<TbItem> rest;
            for (const TbItem& t : tbs) (t.r.ar - t.r.al >= tall ? side_tbs : rest).push_back(t);
            if (side_tbs.empty() || rest.empty() || (int) side_tbs.size() > 2 * ctx->n_cu) side_tbs.clear();
            else {
                tbs.swap(rest);
                std::vector<RunItem> items;
                for (const TbItem& t : side_tbs) items.push_back(run_item(parent(t.job), t.r, t.w, 0));
                side.side = true;
                if (side.build(st, items, 1) || side.launch()) return -1;
                lap("tall fwd build+launch");
            }
        }
        if (!tbs.empty()) {
            std::vector<RunItem> items;
            for (const TbItem& t : tbs) items.push_back(run_item(parent(t.job), t.r, t.w, 0));
            DevRun run;
            run.use_ctx = ctx;
            run.beside = !side_tbs.empty();
            if (run.build(st, items, 1)) return -1;
            lap("fwd build");
            if (run.launch() || run.sync()) return -1;
            lap("fwd launch+sync (+walk)");
            kernel_ms += run.kernel_ms; kernel_cells += run.total_cells;
            stats[3] += run.kernel_ms; stats[4] += (double) run.total_cells; stats[5] += (double) items.size();
            stats[7] += (double) run.tb_bytes;
            std::vector<DevResult> res;
            std::vector<int> nskl;
            std::vector<int64_t> off;
            std::vector<SpdpSkl> skl;
            if (run.fetch_results(res) || run.fetch_skl(nskl, off, skl)) return -1;
            lap("fwd fetch");
            for (size_t k = 0; k < tbs.size(); ++k) {
                const TbItem& t = tbs[k];
                if (nskl[k] == -1) { jobs[t.job].failed = true; ++overflowed; continue; }    // record list beyond its slot: this query only
                if (nskl[k] < 0) { ctx->err = "traceback walk failed"; return -1; }
                set_score(t.job, t.top, res[k].score);
                const SpdpSkl* s = skl.data() + off[k];
                jobs[t.job].rec.insert(jobs[t.job].rec.end(), s, s + nskl[k]);
            }
        }
        if (!side_tbs.empty()) {                        // collect the side run (its own pool: the runs above went on beside it)
            if (side.sync()) return -1;
            lap("side fwd sync");
            kernel_ms += side.kernel_ms; kernel_cells += side.total_cells;
            stats[3] += side.kernel_ms; stats[4] += (double) side.total_cells; stats[5] += (double) side_tbs.size();
            stats[7] += (double) side.tb_bytes;
            std::vector<DevResult> res;
            std::vector<int> nskl;
            std::vector<int64_t> off;
            std::vector<SpdpSkl> skl;
            if (side.fetch_results(res) || side.fetch_skl(nskl, off, skl)) return -1;
            for (size_t k = 0; k < side_tbs.size(); ++k) {
                const TbItem& t = side_tbs[k];
                if (nskl[k] == -1) { jobs[t.job].failed = true; ++overflowed; continue; }    // record list beyond its slot: this query only
                if (nskl[k] < 0) { ctx->err = "traceback walk failed"; return -1; }
                set_score(t.job, t.top, res[k].score);
                const SpdpSkl* sk = skl.data() + off[k];
                jobs[t.job].rec.insert(jobs[t.job].rec.end(), sk, sk + nskl[k]);
            }
        }
        lap("assemble records");
        return 0;
    }

    // globalS_ng tail: header, stdskl, trimskl
    void finish(int i, SpdpAlignment* out) const
    {
        const Job& J = jobs[i];
        out->score = J.score_set ? J.score : SPDP_NEVSEL;
        out->n_skl = 0; out->skl = nullptr;
        out->flags = J.edge ? SPDP_ALN_LEFT_EDGE : 0;
        if (raw) {                              // what lspS_ng appended to the caller's Mfile (no header, any order)
            if (J.failed && req) { out->n_skl = -1; return; }   // (explicit requests: the caller must tell "no records" from "not served")
            if (J.failed || J.rec.empty()) return;
            out->n_skl = (int) J.rec.size();
            out->skl = (SpdpSkl*) malloc(sizeof(SpdpSkl) * J.rec.size());
            memcpy(out->skl, J.rec.data(), sizeof(SpdpSkl) * J.rec.size());
            return;
        }
        if (J.failed || (int) J.rec.size() < 2) return;
        std::vector<SpdpSkl> s = corner_list<1>(J.rec);
        trim_skl(s, probs[i]);
        out->n_skl = (int) s.size() + 1;
        out->skl = (SpdpSkl*) malloc(sizeof(SpdpSkl) * out->n_skl);
        out->skl[0].m = 1;                      // AlgnTrb
        out->skl[0].n = (int) s.size();
        memcpy(out->skl + 1, s.data(), sizeof(SpdpSkl) * s.size());
    }
};

}   // namespace

// ---- resident batches -----------------------------------------------------------------------
struct SpdpBatch {
    SpdpContext* ctx = nullptr;
    DevStore store;
    std::vector<SpdpProblem> probs;
    DevRun score;                                // HomScoreS_ng leg, built once
    bool score_built = false;
    double stats[SPDP_N_STATS] = {0};
};

SpdpBatch* spdp_batch_upload(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n)
{
    if (!ctx || n <= 0) return nullptr;
    SpdpBatch* b = new SpdpBatch();
    b->ctx = ctx;
    b->probs.assign(probs, probs + n);
    if (b->store.upload(ctx, sc, probs, n)) { delete b; return nullptr; }
    return b;
}

void spdp_batch_free(SpdpBatch* bt) { delete bt; }

int64_t spdp_batch_cells(const SpdpBatch* bt)
{
    if (!bt) return 0;
    int64_t c = 0;
    for (const SpdpProblem& p : bt->probs) {
        SpdpWindow w;
        spdp_stripe(&p, bt->store.sc.sh, &w);
        c += spdp_cells(&p, &w);
    }
    return c;
}

int spdp_batch_homscore(SpdpBatch* bt, int32_t* scores, float* kernel_ms)
{
    if (!bt) return -1;
    {   // descriptors are rebuilt per call: work buffers live in the context's pool
        std::vector<RunItem> items;
        for (size_t i = 0; i < bt->probs.size(); ++i)
            items.push_back(spdp_item_of(bt->probs[i], (int) i, bt->store.sc.sh));
        if (bt->score.build(&bt->store, items, 0)) return -1;
    }
    if (bt->score.launch() || bt->score.sync()) return -1;
    if (kernel_ms) *kernel_ms = bt->score.kernel_ms;
    if (scores) {
        std::vector<DevResult> r;
        if (bt->score.fetch_results(r)) return -1;
        for (size_t i = 0; i < r.size(); ++i) scores[i] = r[i].score;
    }
    return 0;
}

static int align_on_store(SpdpContext* ctx, const DevStore* st, const SpdpProblem* probs, int n,
                          SpdpAlignment* out, float* kernel_ms, int64_t* kernel_cells,
                          double* stats = nullptr, bool raw = false, const SpdpRequests* req = nullptr)
{
    // big batches run as chunks on lanes of the context, a software pipeline (ChunkGate); SPDP_CHUNKS=1 turns it off
    int n_chunks = n >= 4096 ? 2 : 1;
    if (const char* e = getenv("SPDP_CHUNKS")) n_chunks = std::max(1, std::min(atoi(e), 8));
    n_chunks = std::min(n_chunks, std::max(1, n / 64));
    // a request batch runs on the dispatcher lane that took it, as ONE chunk: chunk lanes are spdp_lane(ctx, c), and for
    // the seeded dispatchers those are the contexts their sibling dispatchers run on from their own threads
    if (req) n_chunks = 1;
    std::vector<Aligner> al(n_chunks);
    std::vector<ChunkGate> gates(n_chunks);
    std::vector<int> rc(n_chunks, 0);
    for (int c = 0; c < n_chunks; ++c) {
        Aligner& a = al[c];
        a.ctx = spdp_lane(ctx, c);
        if (!a.ctx) { ctx->err = "cannot create a lane context"; return -1; }
        a.st = st;
        a.base = (int) ((int64_t) n * c / n_chunks);
        a.n = (int) ((int64_t) n * (c + 1) / n_chunks) - a.base;
        a.probs = probs + a.base;
        a.out = out ? out + a.base : nullptr;
        a.raw = raw;
        a.req = req;
        if (n_chunks > 1) {
            if (hipEventCreateWithFlags(&gates[c].ev, hipEventDisableTiming) != hipSuccess) {
                for (int k = 0; k < c; ++k) if (gates[k].ev) (void) hipEventDestroy(gates[k].ev);
                ctx->err = "hipEventCreate"; return -1;
            }
            a.gate_out = &gates[c];
            a.gate_in = c > 0 ? &gates[c - 1] : nullptr;
        }
    }
    std::vector<std::thread> workers;
    for (int c = 1; c < n_chunks; ++c)
        workers.emplace_back([&, c, lane_copies = t_lane_copies]() { (void) hipSetDevice(ctx->device); t_lane_copies = lane_copies; rc[c] = al[c].run(); });
    rc[0] = al[0].run();
    for (std::thread& t : workers) t.join();
    for (int c = 0; c < n_chunks; ++c) if (gates[c].ev) (void) hipEventDestroy(gates[c].ev);
    int unsupported = 0, overflowed = 0;
    float kms = 0.f; int64_t kc = 0;
    double st_sum[SPDP_N_STATS] = {0};
    for (int c = 0; c < n_chunks; ++c) {
        if (rc[c]) {                            // alignments other chunks have already handed over do not outlive the failed call
            if (c > 0) ctx->err = al[c].ctx->err;
            if (out) spdp_free_alignments(out, n);
            return -1;
        }
        kms += al[c].kernel_ms; kc += al[c].kernel_cells; unsupported += al[c].unsupported; overflowed += al[c].overflowed;
        for (int k = 0; k < SPDP_N_STATS; ++k) st_sum[k] += al[c].stats[k];
    }
    if (kernel_ms) *kernel_ms = kms;
    if (kernel_cells) *kernel_cells = kc;
    if (stats) memcpy(stats, st_sum, sizeof st_sum);
    if (unsupported) {
        ctx->err = "sub-problems with fewer than 8 query rows need the scalar exact engine: supply "
                   "SpdpScoring.intpen / t53 and SpdpProblem.cano5 / cano3 / dinc";
        return 1;                               // partial: those queries are returned without alignment
    }
    if (overflowed) {
        ctx->err = "traceback record list of a scalar / -A1 engine call exceeds its slot; those queries come back without an alignment";
        return 1;
    }
    return 0;
}

int spdp_batch_align(SpdpBatch* bt, SpdpAlignment* out, float* kernel_ms, int64_t* kernel_cells)
{
    if (!bt) return -1;
    return align_on_store(bt->ctx, &bt->store, bt->probs.data(), (int) bt->probs.size(), out,
                          kernel_ms, kernel_cells, bt->stats);
}

int spdp_batch_stats(const SpdpBatch* bt, double* out, int n)
{
    if (!bt || !out) return -1;
    for (int i = 0; i < n && i < SPDP_N_STATS; ++i) out[i] = bt->stats[i];
    return 0;
}

int spdp_align_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                 SpdpAlignment* out)
{
    if (!ctx) return -1;
    for (int i = 0; i < n_probs; ++i) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0; }
    if (n_probs <= 0) return 0;
    DevStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    return align_on_store(ctx, &st, probs, n_probs, out, nullptr, nullptr);
}

// Aln2s1::lspS_ng (src/fwd2s1.cc:1817-1880) for a caller that keeps the record file itself (seededS_ng / interpolateS):
// the whole ladder below it, records as written (out[i].skl has no header record; globalS_ng's stdskl sorts them)
int spdp_lsp_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs, SpdpAlignment* out)
{
    if (!ctx || !sc || !out) return -1;
    for (int i = 0; i < n_probs; ++i) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0; }
    if (n_probs <= 0) return 0;
    DevStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    return align_on_store(ctx, &st, probs, n_probs, out, nullptr, nullptr, nullptr, true);
}

// engine calls a caller has already dispatched itself (the seeded walk, spdp_seeded.cpp): request k runs lspS_ng (kind 0)
// or trcbkalignS_ng (kind 1; kind 2 with a cut range) on probs[k]'s ranges with windows[k], on the resident inputs of
// store entry parents[k]; out[k]: score + records as written, n_skl = -1 where the request could not be served
int spdp_run_requests(SpdpContext* ctx, const DevStore* st, const SpdpProblem* probs, int n, const SpdpRequests* req,
                      SpdpAlignment* out)
{
    for (int i = 0; i < n; ++i) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0; }
    if (n <= 0) return 0;
    LaneCopies own_stream;                      // (request batches are small: one chunk, this thread)
    return align_on_store(ctx, st, probs, n, out, nullptr, nullptr, nullptr, true, req);
}

// alignS_ng(seqs, pwd, gsi, ori = 3) with seeding off (src/fwd2s1.cc:2746-2760): infer_orientation
// (:2718-2730) scores the query as given and its reverse complement against the opposite strand with
// HomScoreS_ng, keeps the reverse only if it scores strictly higher, then aligns once.
int spdp_align_s_ori3(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* fwd, const SpdpProblem* rev,
                      int n_probs, SpdpAlignment* out, int32_t* orient)
{
    if (!ctx || !sc || !fwd || !rev || !out || !orient) return -1;
    if (n_probs <= 0) return 0;
    std::vector<int32_t> s1(n_probs), s2(n_probs);
    const int r1 = spdp_homscore_s(ctx, sc, fwd, n_probs, s1.data());
    if (r1 < 0) return -1;
    const int r2 = spdp_homscore_s(ctx, sc, rev, n_probs, s2.data());
    if (r2 < 0) return -1;
    std::vector<SpdpProblem> pick(n_probs);
    for (int i = 0; i < n_probs; ++i) {
        orient[i] = s2[i] > s1[i] ? 1 : 0;
        pick[i] = orient[i] ? rev[i] : fwd[i];
    }
    const int r3 = spdp_align_s(ctx, sc, pick.data(), n_probs, out);
    if (r3 < 0) return -1;
    // globalS_ng marks an alignment of the flipped pair: skl->m |= A_RevCom when a->inex.sens is set, which
    // comrev() does (src/fwd2s1.cc:2691-2692, src/aln.h:89)
    for (int i = 0; i < n_probs; ++i)
        if (orient[i] && out[i].skl && out[i].n_skl > 0) out[i].skl[0].m |= 0x10;
    return r1 | r2 | r3;
}
