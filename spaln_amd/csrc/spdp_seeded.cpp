// spdp_seeded.cpp -- alignS_ng with seeding on (algmode.qck = 1 .. 3): the walks of a batch of queries over their HSPs,
// with every DP call of every walk served by the device in common batches.
//
// What it mirrors (ogotoh/spaln v3.0.7): Aln2s1::globalS_ng -> seededS_ng -> interpolateS (src/fwd2s1.cc:2587-2694,
// 2405-2539); the walk itself is spdp_seeded_walk.h.  The reference runs one walk per worker thread and calls its DP
// engines synchronously from deep inside it (spaln -t, src/spaln.cc:1389-1468).  Here a pool of host threads runs the
// walks of a call side by side; a walk that reaches lspS_ng / trcbkalignS_ng parks its request and sleeps; when every
// walk in flight sleeps (or has ended), the calling thread runs all parked requests as ONE set of device launches on
// the resident inputs of the batch (spdp_run_requests: the same rounds as spdp_align_s -- linear-space sweeps, slab
// tracebacks, walks), hands the records back and wakes the walks.  No DP cell of a request is computed on the host.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "spdp_internal.h"
#include "spdp_seeded_walk.h"
#include "spdp_seeded_rv.h"

namespace {
using namespace spdp_seed;

struct DeviceBackend : DpBackend {
    Rendezvous* rv; int query; const SpdpHspSource* src;
    bool failed = false;
    int flags = 0;                              // of all DP calls of the walk
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
            rv->cv_walk.wait(lk, [&] { return p.done; });       // (the dispatcher counts me as running again before it wakes me)
        }
        if (p.failed) { failed = true; return SPDP_NEVSEL; }
        flags |= p.flags;
        rec.insert(rec.end(), p.rec.begin(), p.rec.end());
        return p.score;
    }
    int lsp(const Span& s, const SpdpWindow& w, std::vector<SpdpSkl>& rec) override { return park(0, s, w, nullptr, rec); }
    int trcbk(const Span& s, const SpdpWindow& w, const int* cut, std::vector<SpdpSkl>& rec) override
    {
        return park(cut ? 2 : 1, s, w, cut, rec);
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

// the walks of `n_probs` (query, strand) pairs: scores[i] = what globalS_ng's seededS_ng returns, recs[i] = its record file
// (dummy record first), status[i] = 0, or 1 / 2 for a walk that was not served
static int seeded_core(SpdpContext* ctx, const SpdpScoring* sc, const SpdpSeedParams* sp,
                       const SpdpProblem* probs, int n_probs, const SpdpJuxt* const* hsps, const int32_t* n_hsps,
                       const int32_t* lowest_level, const SpdpHspSource* src,
                       std::vector<int>& scores, std::vector<std::vector<SpdpSkl>>& recs, std::vector<uint8_t>& status)
{
    memset(ctx->seed_stats, 0, sizeof ctx->seed_stats);
    if (sp->qck < 1 || sp->qck > 3) { ctx->err = "SpdpSeedParams.qck must be 1 .. 3"; return -1; }
    if (!sc->intpen || sc->intpen_len <= 0) { ctx->err = "the seeded path needs SpdpScoring.intpen / t53"; return -1; }
    for (int i = 0; i < n_probs; ++i)
        if (!probs[i].sig5 || !probs[i].sig3 || !probs[i].cano5 || !probs[i].cano3 || !probs[i].dinc) {
            ctx->err = "the seeded path needs sig5 / sig3 / cano5 / cano3 / dinc of every problem on the host";
            return -1;
        }
    DevStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;

    Rendezvous rv;
    std::atomic<int> next{0};
    std::atomic<int64_t> n_wilip{0};
    scores.assign(n_probs, SPDP_NEVSEL);
    recs.assign(n_probs, std::vector<SpdpSkl>());
    status.assign(n_probs, 0);                          // 1: the walk met a state it does not serve, 2: a request failed
    int n_threads = 256;
    if (const char* e = getenv("SPDP_SEED_WALKS")) n_threads = std::max(1, atoi(e));
    n_threads = std::min(n_threads, n_probs);
    rv.running = n_threads;
    auto walker = [&]() {
        for (;;) {
            const int q = next.fetch_add(1);
            if (q >= n_probs) break;
            DeviceBackend be;
            be.rv = &rv; be.query = q; be.src = src; be.n_wilip = &n_wilip;
            SeedWalk w;
            const int nh = (hsps && n_hsps && hsps[q]) ? n_hsps[q] : 0;
            if (!bind_problem(w, sc, sp, &probs[q], nh ? hsps[q] : nullptr, nh, lowest_level ? lowest_level[q] : 0)) { status[q] = 1; continue; }
            w.dp = &be;
            const SpdpProblem& p = probs[q];
            const Span whole = {p.a_left, p.a_right, p.b_left, p.b_right, p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr};
            scores[q] = w.run(whole);
            recs[q].swap(w.rec);
            status[q] = be.failed ? 2 : (w.unsupported ? 1 : 0);
            if (be.flags & SPDP_ALN_LEFT_EDGE) status[q] |= 16;
        }
        std::lock_guard<std::mutex> g(rv.mu);
        --rv.running;
        rv.cv_main.notify_one();
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) pool.emplace_back(walker);

    int rc = 0;
    int64_t n_batches = 0, n_kind[3] = {0, 0, 0};
    for (;;) {
        std::vector<Parked*> take;
        {
            std::unique_lock<std::mutex> lk(rv.mu);
            rv.cv_main.wait(lk, [&] { return rv.running == 0; });
            if (rv.parked.empty()) break;               // every walk has ended
            take.swap(rv.parked);
        }
        // one device batch for everything parked
        const int m = (int) take.size();
        std::vector<SpdpProblem> rp(m);
        std::vector<int> parents(m), cuts(2 * m);
        std::vector<SpdpWindow> wins(m);
        std::vector<uint8_t> kinds(m);
        for (int k = 0; k < m; ++k) {
            const Parked& q = *take[k];
            rp[k] = probs[q.query];
            rp[k].a_left = q.s.al; rp[k].a_right = q.s.ar; rp[k].b_left = q.s.bl; rp[k].b_right = q.s.br;
            rp[k].a_exgl = q.s.a_exgl; rp[k].a_exgr = q.s.a_exgr; rp[k].b_exgl = q.s.b_exgl; rp[k].b_exgr = q.s.b_exgr;
            parents[k] = q.query; wins[k] = q.w; kinds[k] = (uint8_t) q.kind;
            cuts[2 * k] = q.cut[0]; cuts[2 * k + 1] = q.cut[1];
            ++n_kind[q.kind];
        }
        std::vector<SpdpAlignment> res(m);
        const SpdpRequests rq = {parents.data(), wins.data(), kinds.data(), cuts.data()};
        int brc = rc < 0 ? -1 : spdp_run_requests(ctx, &st, rp.data(), m, &rq, res.data());
        ++n_batches;
        if (brc < 0) rc = -1;                           // the walks still have to be let go: every request fails from here on
        {
            std::lock_guard<std::mutex> g(rv.mu);
            for (int k = 0; k < m; ++k) {
                Parked& q = *take[k];
                if (brc < 0 || res[k].n_skl < 0) q.failed = true;
                else {
                    q.score = res[k].score; q.flags = res[k].flags;
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
    ctx->seed_stats[0] = n_batches; ctx->seed_stats[1] = n_kind[0]; ctx->seed_stats[2] = n_kind[1] + n_kind[2];
    ctx->seed_stats[3] = n_kind[2]; ctx->seed_stats[4] = n_wilip.load(); ctx->seed_stats[5] = n_probs;
    return rc < 0 ? -1 : 0;
}

namespace {
// globalS_ng's tail: the file without its dummy record -> header, stdskl, trimskl (src/fwd2s1.cc:2684-2693)
void finish_walk(const SpdpProblem& p, int score, const std::vector<SpdpSkl>& rec, bool rev, SpdpAlignment* out)
{
    out->score = score;
    if (rec.size() < 3) return;                         // fewer than two records behind the dummy: no alignment
    std::vector<SpdpSkl> s = corner_list<1>(std::vector<SpdpSkl>(rec.begin() + 1, rec.end()));
    trim_skl_of(s, p);
    out->n_skl = (int) s.size() + 1;
    out->skl = (SpdpSkl*) malloc(sizeof(SpdpSkl) * out->n_skl);
    out->skl[0].m = 1 | (rev ? 0x10 : 0);               // AlgnTrb (| A_RevCom: a->inex.sens after comrev)
    out->skl[0].n = (int) s.size();
    memcpy(out->skl + 1, s.data(), sizeof(SpdpSkl) * s.size());
}
const char* kPartial = "some walks met a state the seeded path does not serve (no HSP source for a recursion level, "
                       "or an engine call outside the sequences); those queries come back without an alignment";
}   // namespace

extern "C" int spdp_align_s_seeded(SpdpContext* ctx, const SpdpScoring* sc, const SpdpSeedParams* sp,
                                   const SpdpProblem* probs, int n_probs, const SpdpJuxt* const* hsps, const int32_t* n_hsps,
                                   const int32_t* lowest_level, const SpdpHspSource* src, SpdpAlignment* out)
{
    if (!ctx || !sc || !sp || !probs || !out) return -1;
    for (int i = 0; i < n_probs; ++i) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0; }
    if (n_probs <= 0) return 0;
    std::vector<int> scores;
    std::vector<std::vector<SpdpSkl>> recs;
    std::vector<uint8_t> status;
    if (seeded_core(ctx, sc, sp, probs, n_probs, hsps, n_hsps, lowest_level, src, scores, recs, status) < 0) return -1;
    int partial = 0;
    for (int i = 0; i < n_probs; ++i) {
        if (status[i] & 15) { ++partial; continue; }
        finish_walk(probs[i], scores[i], recs[i], false, out + i);
        if (status[i] & 16) out[i].flags |= SPDP_ALN_LEFT_EDGE;
    }
    if (partial) { ctx->err = kPartial; return 1; }
    return 0;
}

// alignS_ng(seqs, pwd, gsi, ori = 3) with seeding on (src/fwd2s1.cc:2762-2777): the walk on the pair as given, then on the
// reverse-complemented query against the other genomic strand with the HSP list turned around (reverse_copy_jxt ->
// Seq::revjxt, src/seq.cc:745-755: jx' = a_len - jx - jlen, jy' = b_len - jy - jlen, order reversed); the forward result
// stays unless the reverse one scores strictly higher.  Both walks of every query run in the same device batches.
extern "C" int spdp_align_s_seeded_ori3(SpdpContext* ctx, const SpdpScoring* sc, const SpdpSeedParams* sp,
                                        const SpdpProblem* fwd, const SpdpProblem* rev, int n_probs,
                                        const SpdpJuxt* const* hsps, const int32_t* n_hsps, const int32_t* lowest_level,
                                        const SpdpHspSource* src, SpdpAlignment* out, int32_t* orient)
{
    if (!ctx || !sc || !sp || !fwd || !rev || !out || !orient) return -1;
    for (int i = 0; i < n_probs; ++i) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0; orient[i] = 0; }
    if (n_probs <= 0) return 0;
    const int n2 = 2 * n_probs;
    std::vector<SpdpProblem> both(fwd, fwd + n_probs);
    both.insert(both.end(), rev, rev + n_probs);
    std::vector<std::vector<SpdpJuxt>> turned(n_probs);
    std::vector<const SpdpJuxt*> lists(n2, nullptr);
    std::vector<int32_t> counts(n2, 0), levels(n2, 0);
    for (int i = 0; i < n_probs; ++i) {
        const int nh = (hsps && n_hsps && hsps[i]) ? n_hsps[i] : 0;
        levels[i] = levels[n_probs + i] = lowest_level ? lowest_level[i] : 0;
        if (!nh) continue;
        lists[i] = hsps[i]; counts[i] = nh;
        std::vector<SpdpJuxt>& t = turned[i];
        t.assign(hsps[i], hsps[i] + nh + 1);
        for (int j = 0; j < nh; ++j) {                  // (the forward walk has left {a->len, b->len} in the slot behind the list)
            t[j].jx = fwd[i].a_len - t[j].jx - t[j].jlen;
            t[j].jy = fwd[i].b_len - t[j].jy - t[j].jlen;
        }
        std::reverse(t.begin(), t.begin() + nh);
        lists[n_probs + i] = t.data(); counts[n_probs + i] = nh;
    }
    std::vector<int> scores;
    std::vector<std::vector<SpdpSkl>> recs;
    std::vector<uint8_t> status;
    if (seeded_core(ctx, sc, sp, both.data(), n2, lists.data(), counts.data(), levels.data(), src, scores, recs, status) < 0) return -1;
    int partial = 0;
    for (int i = 0; i < n_probs; ++i) {
        if ((status[i] | status[n_probs + i]) & 15) { ++partial; continue; }
        const int r = n_probs + i;
        orient[i] = scores[i] >= scores[r] ? 0 : 1;
        if (orient[i]) finish_walk(rev[i], scores[r], recs[r], true, out + i);
        else finish_walk(fwd[i], scores[i], recs[i], false, out + i);
        if (status[orient[i] ? r : i] & 16) out[i].flags |= SPDP_ALN_LEFT_EDGE;
    }
    if (partial) { ctx->err = kPartial; return 1; }
    return 0;
}

extern "C" int spdp_seeded_stats(const SpdpContext* ctx, int64_t* out, int n)
{
    if (!ctx || !out) return -1;
    for (int i = 0; i < n && i < 6; ++i) out[i] = ctx->seed_stats[i];
    return 0;
}
