// spdp_seeded_h.cpp -- alignH_ng with seeding on (algmode.qck = 1 .. 3): the protein walks of a batch of queries over their
// HSPs, every DP call of every walk served by the device in common batches.
//
// What it mirrors (ogotoh/spaln v3.0.7): Aln2h1::globalH_ng -> seededH_ng -> interpolateH (src/fwd2h1.cc:3267-3286,
// 3177-3265, 3023-3131); the walk itself is spdp_seeded_walk_h.h.  Same scheme as spdp_seeded.cpp: a pool of host threads
// runs the walks; a walk that reaches lspH_ng / trcbkalignH_ng parks its request; when every walk in flight sleeps, the
// calling thread runs all parked requests as one pass of the protein ladder on the resident inputs of the batch
// (spdh_run_requests, spdp_h_api.cpp: linear-space sweeps, slab tracebacks, the scalar engine with its cut-range and
// no-intron variants).  No DP cell of a request is computed on the host.
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "spdp_internal.h"
#include "spdp_seeded_walk_h.h"
#include "spdp_seeded_rv.h"
#include "spdp_h_requests.h"

namespace {
using namespace spdp_seed;

struct DeviceBackendH : DpBackendH {
    Rendezvous* rv; int query; const SpdpHspSource* src;
    bool failed = false;
    std::atomic<int64_t>* n_wilip;
    int park(int kind, const Span& s, const SpdpWindow& w, const int* cut, std::vector<SpdpSkl>& rec)
    {
        Parked p;
        p.query = query; p.kind = kind; p.s = s; p.w = w;
        if (cut) { p.cut[0] = cut[0]; p.cut[1] = cut[1]; }
        {
            std::unique_lock<std::mutex> lk(rv->mu);
            rv->parked.push_back(&p);
            --rv->running;
            rv->cv_main.notify_one();
            rv->cv_walk.wait(lk, [&] { return p.done; });
        }
        if (p.failed) { failed = true; return SPDP_NEVSEL; }
        rec.insert(rec.end(), p.rec.begin(), p.rec.end());
        return p.score;
    }
    int lsp(const Span& s, const SpdpWindow& w, std::vector<SpdpSkl>& rec) override { return park(0, s, w, nullptr, rec); }
    int trcbk(const Span& s, const SpdpWindow& w, bool spj, const int* cut, std::vector<SpdpSkl>& rec) override
    {
        return park(spj ? 1 : 3, s, w, cut, rec);
    }
    bool wilip(int level, const Span& s, std::vector<Unit>& units) override
    {
        if (!src || !src->units) return false;
        const int32_t span[8] = {s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr};
        const int32_t* flat = nullptr; int32_t n = 0;
        ++*n_wilip;
        if (src->units(src->user, query, level, span, &flat, &n) || !flat) return false;
        const bool ok = parse_units(flat, n, units);
        if (src->release) src->release(src->user, query, flat);
        return ok;
    }
};
}   // namespace

extern "C" int spdp_align_h_seeded(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpSeedParams* sp,
                                   const SpdpProblemH* probs, int n_probs, const SpdpJuxt* const* hsps, const int32_t* n_hsps,
                                   const int32_t* lowest_level, const SpdpHspSource* src, SpdpAlignment* out)
{
    if (!ctx || !sc || !sp || !probs || !out) return -1;
    for (int i = 0; i < n_probs; ++i) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0; }
    if (n_probs <= 0) return 0;
    memset(ctx->seed_stats, 0, sizeof ctx->seed_stats);
    if (sp->qck < 1 || sp->qck > 3) { ctx->err = "SpdpSeedParams.qck must be 1 .. 3"; return -1; }
    if (!sc->intpen || sc->intpen_len <= 0) { ctx->err = "the seeded path needs SpdpScoringH.intpen / t53"; return -1; }
    for (int i = 0; i < n_probs; ++i)
        if (!probs[i].sig5 || !probs[i].sig3 || !probs[i].sigS || !probs[i].sigT || !probs[i].sigE || !probs[i].phs5 ||
            !probs[i].phs3 || !probs[i].dinc) {
            ctx->err = "the seeded path needs the seven signal arrays and dinc of every problem on the host";
            return -1;
        }
    HStore* st = spdh_store_open(ctx, sc, probs, n_probs);
    if (!st) return -1;

    Rendezvous rv;
    std::atomic<int> next{0};
    std::atomic<int64_t> n_wilip{0};
    std::vector<int> scores(n_probs, SPDP_NEVSEL);
    std::vector<std::vector<SpdpSkl>> recs(n_probs);
    std::vector<uint8_t> status(n_probs, 0);            // 1: the walk met a state it does not serve, 2: a request failed
    int n_threads = 256;
    if (const char* e = getenv("SPDP_SEED_WALKS")) n_threads = std::max(1, atoi(e));
    n_threads = std::min(n_threads, n_probs);
    rv.running = n_threads;
    auto walker = [&]() {
        for (;;) {
            const int q = next.fetch_add(1);
            if (q >= n_probs) break;
            DeviceBackendH be;
            be.rv = &rv; be.query = q; be.src = src; be.n_wilip = &n_wilip;
            SeedWalkH w;
            const int nh = (hsps && n_hsps && hsps[q]) ? n_hsps[q] : 0;
            if (!bind_problem_h(w, sc, sp, &probs[q], nh ? hsps[q] : nullptr, nh, lowest_level ? lowest_level[q] : 0)) { status[q] = 1; continue; }
            w.dp = &be;
            const SpdpProblemH& p = probs[q];
            const Span whole = {p.a_left, p.a_right, p.b_left, p.b_right, p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr};
            scores[q] = w.run(whole);
            recs[q].swap(w.rec);
            status[q] = be.failed ? 2 : (w.unsupported ? 1 : 0);
        }
        std::lock_guard<std::mutex> g(rv.mu);
        --rv.running;
        rv.cv_main.notify_one();
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) pool.emplace_back(walker);

    int rc = 0;
    int64_t n_batches = 0, n_kind[4] = {0, 0, 0, 0}, n_cut = 0;
    for (;;) {
        std::vector<Parked*> take;
        {
            std::unique_lock<std::mutex> lk(rv.mu);
            rv.cv_main.wait(lk, [&] { return rv.running == 0; });
            if (rv.parked.empty()) break;               // every walk has ended
            take.swap(rv.parked);
        }
        const int m = (int) take.size();
        std::vector<SpdhRequest> rq(m);
        for (int k = 0; k < m; ++k) {
            const Parked& q = *take[k];
            SpdhRequest& r = rq[k];
            r.parent = q.query; r.al = q.s.al; r.ar = q.s.ar; r.bl = q.s.bl; r.br = q.s.br;
            r.exg[0] = q.s.a_exgl; r.exg[1] = q.s.a_exgr; r.exg[2] = q.s.b_exgl; r.exg[3] = q.s.b_exgr;
            r.w = q.w; r.kind = q.kind; r.cut_l = q.cut[0]; r.cut_r = q.cut[1];
            ++n_kind[q.kind & 3];
            if (q.cut[1] > q.cut[0]) ++n_cut;
        }
        std::vector<SpdpAlignment> res(m);
        const int brc = rc < 0 ? -1 : spdh_run_requests(st, rq.data(), m, res.data());
        ++n_batches;
        if (brc < 0) rc = -1;                           // the walks still have to be let go: every request fails from here on
        {
            std::lock_guard<std::mutex> g(rv.mu);
            for (int k = 0; k < m; ++k) {
                Parked& q = *take[k];
                if (brc < 0 || res[k].n_skl < 0) q.failed = true;
                else {
                    q.score = res[k].score;
                    if (res[k].n_skl > 0) q.rec.assign(res[k].skl, res[k].skl + res[k].n_skl);
                }
                q.done = true;
            }
            rv.running += m;
        }
        rv.cv_walk.notify_all();
        if (brc >= 0) spdp_free_alignments(res.data(), m);
    }
    for (std::thread& t : pool) t.join();
    spdh_store_close(st);
    ctx->seed_stats[0] = n_batches; ctx->seed_stats[1] = n_kind[0]; ctx->seed_stats[2] = n_kind[1] + n_kind[3];
    ctx->seed_stats[3] = n_cut; ctx->seed_stats[4] = n_wilip.load(); ctx->seed_stats[5] = n_probs;
    if (rc < 0) return -1;

    // globalH_ng's tail (src/fwd2h1.cc:3277-3285): the file without its dummy record -> header + stdskl3
    int partial = 0;
    for (int i = 0; i < n_probs; ++i) {
        if (status[i]) { ++partial; continue; }
        out[i].score = scores[i];
        if (recs[i].size() < 3) continue;               // fewer than two records behind the dummy: no alignment
        std::vector<SpdpSkl> s = corner_list<3>(std::vector<SpdpSkl>(recs[i].begin() + 1, recs[i].end()));
        out[i].n_skl = (int) s.size() + 1;
        out[i].skl = (SpdpSkl*) malloc(sizeof(SpdpSkl) * out[i].n_skl);
        out[i].skl[0].m = 1;
        out[i].skl[0].n = (int) s.size();
        memcpy(out[i].skl + 1, s.data(), sizeof(SpdpSkl) * s.size());
    }
    if (partial) {
        ctx->err = "some protein walks met a state the seeded path does not serve (no HSP source for a recursion "
                   "level, or a DP call the reference itself leaves undefined); those queries "
                   "come back without an alignment";
        return 1;
    }
    return 0;
}
