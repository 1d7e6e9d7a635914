// walk_check.cpp -- CPU checker for the seeded walk.  TEST INFRASTRUCTURE ONLY (same rules as spdp_oracle.c: only
// tests/ load this library; the product never does).
//
// The product's host walk (spaln_amd/csrc/spdp_walk.h: seededS_ng / interpolateS and the closed-form joins,
// src/fwd2s1.cc:1899-2672) is a header shared by the product library, where its DP calls go to the device, and by this
// library, where they go to callbacks -- the tests bind those to the oracle's ladder (oracle/host_logic.py: lsp, trcbk)
// and to the Wilip replies a `ref_dump -Q` fixture recorded.  That way the walk's decisions are checked against the
// reference's seeded alignments without a GPU, and the -m gpu tests check the same source with the device behind it.
#include <atomic>
#include <cstring>
#include <vector>

#include "../spaln_amd/csrc/spdp_walk.h"
#include "../spaln_amd/csrc/spdp_hsp_host.h"
#include "../spaln_amd/csrc/spdp_seeded_rv.h"

extern "C" {
// kind 0: lspS_ng, 1: trcbkalignS_ng, 2: Wilip.  args: a_left, a_right, b_left, b_right, a_exgl, a_exgr, b_exgl, b_exgr,
// lw, up, width, has_cut, cut_left, cut_right, level.  The callee returns 0 and points *out at int32 data it keeps alive
// until its next call: kinds 0 / 1: score, then (m, n) pairs; kind 2: the flat unit record of SpdpHspSource.
typedef int (*WalkCheckFn)(void* user, int kind, const int32_t* args, const int32_t** out, int32_t* n_out);
}

namespace {
using namespace spdp_seed;

struct CallbackBackend : DpBackend {
    WalkCheckFn fn; void* user; bool failed = false;
    int call(int kind, const Span& s, const SpdpWindow* w, const int* cut, int level, const int32_t** out, int32_t* n)
    {
        int32_t args[15] = {s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr, w ? w->lw : 0, w ? w->up : 0,
                            w ? w->width : 0, cut ? 1 : 0, cut ? cut[0] : 0, cut ? cut[1] : 0, level};
        return fn(user, kind, args, out, n);
    }
    int dp(int kind, const Span& s, const SpdpWindow& w, const int* cut, std::vector<SpdpSkl>& rec)
    {
        const int32_t* out = nullptr; int32_t n = 0;
        if (call(kind, s, &w, cut, 0, &out, &n) || n < 1) { failed = true; return SPDP_NEVSEL; }
        for (int i = 1; i + 1 < n; i += 2) rec.push_back({out[i], out[i + 1]});
        return out[0];
    }
    int lsp(const Span& s, const SpdpWindow& w, std::vector<SpdpSkl>& rec) override { return dp(0, s, w, nullptr, rec); }
    int trcbk(const Span& s, const SpdpWindow& w, bool, const int* cut, std::vector<SpdpSkl>& rec) override { return dp(1, s, w, cut, rec); }
    bool wilip(int level, const Span& s, std::vector<Unit>& units) override
    {
        const int32_t* out = nullptr; int32_t n = 0;
        if (call(2, s, nullptr, nullptr, level, &out, &n) || n < 1) { failed = true; return false; }
        return spdp_seed::parse_units(out, n, units);
    }
};
}   // namespace

extern "C" int walk_check_n_joins() { return SeedWalk::J_COUNT; }

// one query through globalS_ng with seeding on; rec receives the record file (dummy record first), *n_rec its size
// (cap entries available).  Returns 0, 1 when the walk met a state it does not serve, -1 on a callback failure.
extern "C" int walk_check_run(const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpProblem* p,
                              const SpdpJuxt* hsps, int n_hsps, int lowest_level, WalkCheckFn fn, void* user,
                              int32_t* score, SpdpSkl* rec, int cap, int* n_rec, int32_t* joins /* J_COUNT or null */)
{
    CallbackBackend be;
    be.fn = fn; be.user = user;
    SeedWalk w;
    if (!spdp_seed::bind_problem(w, sc, sp, p, hsps, n_hsps, lowest_level)) return -1;
    w.dp = &be;
    const Span whole = {p->a_left, p->a_right, p->b_left, p->b_right, p->a_exgl, p->a_exgr, p->b_exgl, p->b_exgr};
    *score = w.run(whole);
    *n_rec = (int) w.rec.size();
    if (joins) for (int k = 0; k < SeedWalk::J_COUNT; ++k) joins[k] = w.joins[k];
    if ((int) w.rec.size() > cap) return -1;
    if (!w.rec.empty()) memcpy(rec, w.rec.data(), sizeof(SpdpSkl) * w.rec.size());
    if (be.failed) return -1;
    return w.unsupported ? 1 : 0;
}

// ---- the protein walk (Walk<ProteinPath>), same scheme: kind 0 lspH_ng, 1 trcbkalignH_ng, 2 Wilip, 3 trcbkalignH_ng
// without introns (spj = false)
namespace {
struct CallbackBackendH : DpBackend {
    WalkCheckFn fn; void* user; bool failed = false;
    int call(int kind, const Span& s, const SpdpWindow* w, const int* cut, int level, const int32_t** out, int32_t* n)
    {
        int32_t args[15] = {s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr, w ? w->lw : 0, w ? w->up : 0,
                            w ? w->width : 0, cut ? 1 : 0, cut ? cut[0] : 0, cut ? cut[1] : 0, level};
        return fn(user, kind, args, out, n);
    }
    int dp(int kind, const Span& s, const SpdpWindow& w, const int* cut, std::vector<SpdpSkl>& rec)
    {
        const int32_t* out = nullptr; int32_t n = 0;
        if (call(kind, s, &w, cut, 0, &out, &n) || n < 1) { failed = true; return SPDP_NEVSEL; }
        for (int i = 1; i + 1 < n; i += 2) rec.push_back({out[i], out[i + 1]});
        return out[0];
    }
    int lsp(const Span& s, const SpdpWindow& w, std::vector<SpdpSkl>& rec) override { return dp(0, s, w, nullptr, rec); }
    int trcbk(const Span& s, const SpdpWindow& w, bool spj, const int* cut, std::vector<SpdpSkl>& rec) override { return dp(spj ? 1 : 3, s, w, cut, rec); }
    bool wilip(int level, const Span& s, std::vector<Unit>& units) override
    {
        const int32_t* out = nullptr; int32_t n = 0;
        if (call(2, s, nullptr, nullptr, level, &out, &n) || n < 1) { failed = true; return false; }
        return spdp_seed::parse_units(out, n, units);
    }
};
}   // namespace

extern "C" int walk_check_n_joins_h() { return SeedWalkH::J_COUNT; }
static thread_local std::vector<int32_t> g_marks_h;    // of the last walk_check_run_h on this thread

extern "C" int walk_check_run_h(const SpdpScoringH* sc, const SpdpSeedParams* sp, const SpdpProblemH* p,
                                const SpdpJuxt* hsps, int n_hsps, int lowest_level, WalkCheckFn fn, void* user,
                                int32_t* score, SpdpSkl* rec, int cap, int* n_rec, int32_t* joins)
{
    CallbackBackendH be;
    be.fn = fn; be.user = user;
    SeedWalkH w;
    if (!spdp_seed::bind_problem_h(w, sc, sp, p, hsps, n_hsps, lowest_level)) return -1;
    w.dp = &be;
    const Span whole = {p->a_left, p->a_right, p->b_left, p->b_right, p->a_exgl, p->a_exgr, p->b_exgl, p->b_exgr};
    *score = w.run(whole);
    *n_rec = (int) w.rec.size();
    if (joins) for (int k = 0; k < SeedWalkH::J_COUNT; ++k) joins[k] = w.joins[k];
    g_marks_h.clear();                              // {position, side, value} of the marks the walk made (spdp_seeded_phase_marks of the product)
    for (const auto& e : w.phs5.edits) { g_marks_h.push_back(e.first); g_marks_h.push_back(5); g_marks_h.push_back(e.second); }
    for (const auto& e : w.phs3.edits) { g_marks_h.push_back(e.first); g_marks_h.push_back(3); g_marks_h.push_back(e.second); }
    if ((int) w.rec.size() > cap) return -1;
    if (!w.rec.empty()) memcpy(rec, w.rec.data(), sizeof(SpdpSkl) * w.rec.size());
    if (be.failed) return -1;
    return w.unsupported ? 1 : 0;
}
extern "C" int walk_check_marks_h(const int32_t** out) { *out = g_marks_h.data(); return (int) g_marks_h.size() / 3; }

// SeedWalkH::split_codon on the whole active range of a problem (tests/test_oracle_spjseq.py)
extern "C" int walk_check_split_codon_h(const SpdpProblemH* p, int n5, int n3, int32_t* cs)
{
    SeedWalkH w;
    w.b = p->b; w.b_len = p->b_len;
    spdp_genetic_code_tables(w.mid, w.tron_of);
    w.cur = {p->a_left, p->a_right, p->b_left, p->b_right, 0, 0, 0, 0};
    int c[2] = {0, 0};
    const bool ok = w.split_codon(n5, n3, c);
    cs[0] = c[0]; cs[1] = c[1];
    return ok ? 0 : 1;
}

// ---- the fiber scheduler of the seeded drivers (spdp_seeded_rv.h) without a device ----------------------------------
// n_walks toy walks: walk q makes 1 + (q * 7919) % max_parks requests in a row from `depth` frames down a recursion (each
// frame keeps 1 KB on the fiber's stack), request k of walk q asks for f(q, k) = q * 131 + k * 17 and its latency class
// is k % 3; the "device" (whatever dispatcher lane picks the batch up) answers.  out[q] = sum of the answers.  Returns the
// number of batches, or -1 when the scheduler reports a failure.
namespace {
int toy_descend(spdp_seed::Fiber& fb, int q, int k, int depth)
{
    volatile char pad[1024];
    pad[0] = (char) depth; pad[1023] = (char) q;
    if (depth > 0) return toy_descend(fb, q, k, depth - 1) + (pad[0] - (char) depth) + (pad[1023] - (char) q);
    spdp_seed::Parked p;
    p.query = q; p.kind = k;
    fb.park(&p);
    return p.failed ? -1000000 : p.score;
}
}   // namespace

extern "C" int walk_check_scheduler(int n_walks, int max_parks, int depth, int l0, int l1, int l2, int64_t* out)
{
    using namespace spdp_seed;
    std::atomic<int> n_batches{0};
    std::mutex mu;
    bool bad_class = false;
    const std::vector<int> class_of_lane = lanes_per_class(n_walks, {l0, l1, l2});
    const int n_cls = class_of_lane.back() + 1;
    auto walk = [&](int q, Fiber& fb) {
        const int parks = 1 + (int) (((int64_t) q * 7919) % max_parks);
        int64_t sum = 0;
        for (int k = 0; k < parks; ++k) sum += toy_descend(fb, q, k, depth);
        out[q] = sum;
    };
    auto device = [&](std::vector<Parked*>& take, int lane) {
        ++n_batches;
        for (Parked* p : take) {
            if (std::min(p->kind % 3, n_cls - 1) != class_of_lane[lane]) { std::lock_guard<std::mutex> g(mu); bad_class = true; }
            p->score = p->query * 131 + p->kind * 17;
        }
    };
    auto cls = [n_cls](const Parked& p) { return std::min(p.kind % 3, n_cls - 1); };
    WalkScheduler ws;
    if (!ws.run(n_walks, walk, device, class_of_lane, cls) || bad_class) return -1;
    return n_batches.load();
}

// ---- the HSP search (spdp_hsp_host.h) on one request, as Wilip::Wilip(seqs, pwd, level) answers it: flat units into out[cap];
// returns the number of ints (-1: cap too small)
extern "C" int walk_check_wilip(const SpdpWilipModel* m, const uint8_t* a, int a_len, int a_left, int a_right, int a_exgl, int a_exgr,
                                const uint8_t* b, int b_len, int b_left, int b_right, int bbt,
                                const int16_t* sigS, const int16_t* sigE, const int16_t* sigT,
                                const int16_t* intpen, int intpen_len, int gop, int gep, int lgop, int lgep, int codonk1,
                                int level, int32_t* out, int cap)
{
    const spdp_hsp::Seqs p = {a, a_len, a_left, a_right, a_exgl, a_exgr, b, b_len, b_left, b_right, bbt, sigS, sigE, sigT};
    const spdp_hsp::GapCosts gc = {intpen, intpen_len, gop, gep, lgop, lgep, codonk1};
    std::vector<spdp_hsp::Unit> units;
    spdp_hsp::search(m, p, gc, level, units);
    std::vector<int32_t> flat;
    spdp_hsp::flatten(units, flat);
    if ((int) flat.size() > cap) return -1;
    memcpy(out, flat.data(), sizeof(int32_t) * flat.size());
    return (int) flat.size();
}
