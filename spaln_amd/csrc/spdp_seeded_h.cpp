// spdp_seeded_h.cpp -- alignH_ng with seeding on (algmode.qck = 1 .. 3): the protein walks of a batch of queries over their
// HSPs, every DP call of every walk served by the device in common batches.
//
// What it mirrors (ogotoh/spaln v3.0.7): Aln2h1::globalH_ng -> seededH_ng -> interpolateH (src/fwd2h1.cc:3267-3286,
// 3177-3265, 3023-3131); the walk itself is spdp_walk.h (Walk<ProteinPath>).  Same scheme as spdp_seeded.cpp: the walks run on
// fibers (spdp_seeded_rv.h); a walk that reaches lspH_ng / trcbkalignH_ng parks its request; the parked requests of a
// latency class run as one pass of the protein ladder on the resident inputs of the batch (spdh_run_requests,
// spdp_h_api.cpp: linear-space sweeps, slab tracebacks, the scalar engine with its cut-range and no-intron variants) on
// that class's dispatcher lane.  No DP cell of a request is computed on the host.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "spdp_internal.h"
#include "spdp_walk.h"
#include "spdp_hsp_host.h"
#include "spdp_seeded_rv.h"
#include "spdp_h_requests.h"

namespace {
using namespace spdp_seed;

struct DeviceBackendH : DpBackend {
    Fiber* fiber; int query; const SpdpHspSource* src;
    bool failed = false;
    std::atomic<int64_t>* n_wilip;
    // the scout pass (spdp_seeded.cpp, seeded_core: the same two runs over one cache)
    RequestCache* cache = nullptr;
    int pass = 1;
    const std::function<bool(const Parked&)>* slow = nullptr;
    int take(const Parked& p, std::vector<SpdpSkl>& rec)
    {
        if (p.failed) { failed = true; return SPDP_NEVSEL; }
        rec.insert(rec.end(), p.rec.begin(), p.rec.end());
        return p.score;
    }
    int park(int kind, const Span& s, const SpdpWindow& w, const int* cut, std::vector<SpdpSkl>& rec)
    {
        if (cache) {
            int key[14];
            RequestCache::make_key(key, kind, s, w, cut);
            if (RequestCache::Entry* e = cache->find(key)) {
                if (e->p.async && !e->p.done) {
                    if (pass == 0) return SPDP_NEVSEL;
                    fiber->wait_for(&e->p);
                }
                return take(e->p, rec);
            }
            if (pass == 0) {
                RequestCache::Entry* e = cache->add(key);
                Parked& p = e->p;
                p.query = query; p.kind = kind; p.s = s; p.w = w;
                if (cut) { p.cut[0] = cut[0]; p.cut[1] = cut[1]; }
                if ((*slow)(p)) { fiber->submit(&p); return SPDP_NEVSEL; }
                fiber->park(&p);
                return take(p, rec);
            }
        }
        Parked p;
        p.query = query; p.kind = kind; p.s = s; p.w = w;
        if (cut) { p.cut[0] = cut[0]; p.cut[1] = cut[1]; }
        fiber->park(&p);                        // back when the request has been served
        if (const char* tf = getenv("SPDP_SEED_TRACE"))         // (looking inside a walk: every DP request with what came back)
            if (FILE* f = fopen(tf, "a")) {
                fprintf(f, "q %d dp kind %d a %d..%d b %d..%d exg %d%d%d%d w %d %d %d cut %d %d -> %s score %d, %d records:", query, kind, s.al, s.ar, s.bl,
                        s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr, w.lw, w.up, w.width, p.cut[0], p.cut[1], p.failed ? "FAILED" : "ok", p.score, (int) p.rec.size());
                for (const SpdpSkl& r : p.rec) fprintf(f, " %d,%d", r.m, r.n);
                fprintf(f, "\n");
                fclose(f);
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
    const SpdpWilipModel* wm = nullptr; const SpdpProblemH* prob = nullptr; const SpdpScoringH* scp = nullptr;
    bool wilip(int level, const Span& s, std::vector<Unit>& units) override
    {
        if ((!src || !src->units) && wm) {      // the library's own HSP search (spdp_hsp_host.h)
            int skey[9] = {level, s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr};
            if (cache)
                for (auto& e : cache->searches) if (!memcmp(e->key, skey, sizeof skey)) { units = e->units; if (pass) ++*n_wilip; return e->ok; }
            if (pass) ++*n_wilip;
            const spdp_hsp::Seqs pr = {prob->a, prob->a_len, s.al, s.ar, s.a_exgl, s.a_exgr, prob->b, prob->b_len, s.bl, s.br, 3,
                                        prob->sigS, prob->sigE, prob->sigT};
            const spdp_hsp::GapCosts gc = {scp->intpen, scp->intpen_len, scp->gop, scp->gep, scp->lgop, scp->lgep, scp->codonk1};
            std::vector<spdp_hsp::Unit> us;
            spdp_hsp::search(wm, pr, gc, level, us);
            std::vector<int32_t> flat;
            spdp_hsp::flatten(us, flat);
            const bool ok = parse_units(flat.data(), (int32_t) flat.size(), units);
            if (cache) {
                cache->searches.emplace_back(new RequestCache::Search);
                memcpy(cache->searches.back()->key, skey, sizeof skey);
                cache->searches.back()->ok = ok; cache->searches.back()->units = units;
            }
            return ok;
        }
        if (!src || !src->units) return false;
        const int32_t span[8] = {s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr};
        const int32_t* flat = nullptr; int32_t n = 0;
        ++*n_wilip;
        if (src->units(src->user, query, level, span, &flat, &n) || !flat) return false;
        const bool ok = parse_units(flat, n, units);
        if (const char* tf = getenv("SPDP_SEED_TRACE"))
            if (FILE* f = fopen(tf, "a")) {
                fprintf(f, "q %d wilip level %d a %d..%d b %d..%d exg %d%d%d%d ->", query, level, s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr);
                for (int k = 0; k < n; ++k) fprintf(f, " %d", flat[k]);
                fprintf(f, "\n");
                fclose(f);
            }
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
    ctx->seed_marks.assign(n_probs, {});
    if (sp->qck < 1 || sp->qck > 3) { ctx->err = "SpdpSeedParams.qck must be 1 .. 3"; return -1; }
    if (!sc->intpen || sc->intpen_len <= 0) { ctx->err = "the seeded path needs SpdpScoringH.intpen / t53"; return -1; }
    for (int i = 0; i < n_probs; ++i)
        if (!probs[i].sig5 || !probs[i].sig3 || !probs[i].sigS || !probs[i].sigT || !probs[i].sigE || !probs[i].phs5 ||
            !probs[i].phs3 || !probs[i].dinc) {
            ctx->err = "the seeded path needs the seven signal arrays and dinc of every problem on the host";
            return -1;
        }
    const auto t_begin = std::chrono::steady_clock::now();
    auto us_since = [](std::chrono::steady_clock::time_point t) {
        return (int64_t) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count(); };
    HStore* st = spdh_store_open(ctx, sc, probs, n_probs);
    if (!st) return -1;
    const int64_t us_upload = us_since(t_begin);
    int64_t us_walks = 0, us_device = 0, us_hand = 0;

    std::atomic<int64_t> n_wilip{0};
    std::vector<int> scores(n_probs, SPDP_NEVSEL);
    std::vector<std::vector<SpdpSkl>> recs(n_probs);
    std::vector<uint8_t> status(n_probs, 0);            // 1: the walk met a state it does not serve, 2: a request failed
    // the scout pass: see seeded_core (spdp_seeded.cpp)
    std::function<int(const Parked&)> cls_fn;
    int slow_class = -1;
    const std::function<bool(const Parked&)> is_slow = [&](const Parked& r) { return slow_class >= 0 && cls_fn(r) >= slow_class; };
    auto walk = [&](int q, Fiber& fb) {
        RequestCache cache;
        const bool two = slow_class >= 0;
        for (int pass = two ? 0 : 1; pass < 2; ++pass) {
            // every way out of the scout pass counts (the dispatcher of the slow class starts once all walks have scouted); the
            // wake-up under the scheduler's mutex, so that it cannot fall between the dispatcher's test and its sleep
            struct ScoutCounted {
                WalkScheduler* s; int n; bool on;
                ~ScoutCounted() { if (on && ++s->scouted >= n) { std::lock_guard<std::mutex> g(s->mu); s->cv_main.notify_all(); } }
            } scout_counted{fb.sched, n_probs, pass == 0};
            try {                                   // (everything a walk allocates is inside: a walk that throws fails alone)
                DeviceBackendH be;
                be.fiber = &fb; be.query = q; be.src = src; be.n_wilip = &n_wilip;
                be.wm = sp->wilip; be.prob = &probs[q]; be.scp = sc;
                if (two) { be.cache = &cache; be.pass = pass; be.slow = &is_slow; }
                SeedWalkH w;
                const int nh = (hsps && n_hsps && hsps[q]) ? n_hsps[q] : 0;
                if (!bind_problem_h(w, sc, sp, &probs[q], nh ? hsps[q] : nullptr, nh, lowest_level ? lowest_level[q] : 0)) { status[q] = 1; break; }
                w.dp = &be;
                const SpdpProblemH& p = probs[q];
                const Span whole = {p.a_left, p.a_right, p.b_left, p.b_right, p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr};
                const int score = w.run(whole);
                if (pass == 0) { fb.flush(); continue; }
                scores[q] = score;
                recs[q].swap(w.rec);
                for (const auto& e : w.phs5.edits) ctx->seed_marks[q].push_back({e.first, 5, e.second, 0});      // (one walk per query: no lock)
                for (const auto& e : w.phs3.edits) ctx->seed_marks[q].push_back({e.first, 3, e.second, 0});
                status[q] = be.failed ? 2 : (w.unsupported ? 1 : 0);
            } catch (...) { if (pass) status[q] = 2; }  // (out of memory inside one walk: that query comes back without an alignment)
        }
        for (auto& e : cache.all) if (e->p.async && !e->p.done) fb.wait_for(&e->p);
    };

    std::atomic<int> rc{0};
    int64_t n_batches = 0, n_kind[4] = {0, 0, 0, 0}, n_cut = 0;
    auto t_idle = std::chrono::steady_clock::now();
    std::mutex stats_mu;
    const std::vector<int> class_of_lane = lanes_per_class(n_probs, {2, 1, 1});
    const int n_lanes = (int) class_of_lane.size();
    if (n_lanes > 1 && !spdp_lane(ctx, n_lanes - 1)) { spdh_store_close(st); return -1; }     // (created here, on one thread)
    int busy_lanes = 0;
    std::vector<int64_t> lane_n(n_lanes, 0), lane_us(n_lanes, 0), lane_req(n_lanes, 0);
    const bool shape_stats = getenv("SPDP_SEED_VERBOSE") && atoi(getenv("SPDP_SEED_VERBOSE")) >= 2;
    int64_t shape_n[6] = {0, 0, 0, 0, 0, 0}, shape_steps[6] = {0, 0, 0, 0, 0, 0}, shape_cells[6] = {0, 0, 0, 0, 0, 0};
    auto device = [&](std::vector<Parked*>& take, int lane) {
        (void) hipSetDevice(ctx->device);
        { std::lock_guard<std::mutex> g(stats_mu); if (!busy_lanes++) us_walks += us_since(t_idle); }
        auto t0 = std::chrono::steady_clock::now();
        const int m = (int) take.size();
        std::vector<SpdhRequest> rq(m);
        for (int k = 0; k < m; ++k) {
            const Parked& q = *take[k];
            SpdhRequest& r = rq[k];
            r.parent = q.query; r.al = q.s.al; r.ar = q.s.ar; r.bl = q.s.bl; r.br = q.s.br;
            r.exg[0] = q.s.a_exgl; r.exg[1] = q.s.a_exgr; r.exg[2] = q.s.b_exgl; r.exg[3] = q.s.b_exgr;
            r.w = q.w; r.kind = q.kind; r.cut_l = q.cut[0]; r.cut_r = q.cut[1];
        }
        if (shape_stats) {                              // SPDP_SEED_VERBOSE=2: what the walks ask of the device, by query rows
            std::lock_guard<std::mutex> g(stats_mu);
            for (int k = 0; k < m; ++k) {
                const int rows = rq[k].ar - rq[k].al;
                const int64_t cols = std::max<int64_t>(0, std::min<int64_t>(rq[k].br - rq[k].bl, (int64_t) rq[k].w.up - rq[k].w.lw + 3 * rows));
                const int b = rows < 8 ? 0 : rows < 16 ? 1 : rows < 32 ? 2 : rows < 64 ? 3 : rows < 128 ? 4 : 5;
                ++shape_n[b]; shape_steps[b] += cols + rows; shape_cells[b] += cols * rows;
            }
        }
        std::vector<SpdpAlignment> res(m);
        SpdpContext* lc = spdp_lane(ctx, lane);
        const int brc = rc < 0 ? -1 : spdh_run_requests(st, rq.data(), m, res.data(), lc);
        const int64_t us_dev = us_since(t0);
        t0 = std::chrono::steady_clock::now();
        if (brc < 0) { rc = -1; std::lock_guard<std::mutex> g(stats_mu); if (lc != ctx) ctx->err = lc->err; }                           // the walks still have to be let go: every request fails from here on
        for (int k = 0; k < m; ++k) {
            Parked& q = *take[k];
            if (brc < 0 || res[k].n_skl < 0) q.failed = true;
            else {
                q.score = res[k].score;
                if (res[k].n_skl > 0) q.rec.assign(res[k].skl, res[k].skl + res[k].n_skl);
            }
        }
        if (brc >= 0) spdp_free_alignments(res.data(), m);
        std::lock_guard<std::mutex> g(stats_mu);
        ++n_batches;
        lane_n[lane] += 1; lane_us[lane] += us_dev; lane_req[lane] += m;
        for (int k = 0; k < m; ++k) { ++n_kind[take[k]->kind & 3]; if (take[k]->cut[1] > take[k]->cut[0]) ++n_cut; }
        us_device += us_dev; us_hand += us_since(t0);
        if (!--busy_lanes) t_idle = std::chrono::steady_clock::now();
    };
    WalkScheduler ws;
    // the walks of long windows first: their gaps hold the long introns, i.e. the slow requests, and a call ends with its last walk
    ws.order.resize(n_probs);
    for (int i = 0; i < n_probs; ++i) ws.order[i] = i;
    std::stable_sort(ws.order.begin(), ws.order.end(), [&](int x, int y) {
        return probs[x].b_right - probs[x].b_left > probs[y].b_right - probs[y].b_left; });
    // latency classes: a sweep walks its columns one step at a time, 64 query rows per pass (~0.13 us a step); the scalar
    // engine (fewer than 8 rows, cut ranges) takes about eight times as long per step.  Two dispatchers for the short class:
    // its batches are bound by launch and read-back latency, not by the device.  (Columns count once, in nucleotides: weighted
    // three times, as the three phases of a protein column suggest, a third of all requests landed in the long class and waited
    // for its 30 ms batches -- 20 000 pairs 0.39 s against 0.29 s.)
    const int n_cls = class_of_lane.back() + 1;
    int64_t scalar_w = 8;
    if (const char* e = getenv("SPDP_SEED_SCALAR_W")) scalar_w = std::max(1, atoi(e));      // (tuning)
    auto cls = [n_cls, scalar_w](const Parked& q) {
        const int rows = q.s.ar - q.s.al;
        const int64_t cols = std::max<int64_t>(0, (int64_t) std::min(q.s.br - q.s.bl, q.w.up - q.w.lw + 3 * rows) - (q.cut[1] > q.cut[0] ? q.cut[1] - q.cut[0] : 0));
        const int64_t steps = (rows < 8 || q.kind != 0) ? scalar_w * (cols + rows) : (int64_t) ((rows + 63) / 64) * cols;
        static const int64_t thr[] = {1500, 6000, 18000, 45000};     // (five classes measured: no gain over three; the defaults give lanes to the first three)
        return latency_class(steps, thr, (int) (sizeof thr / sizeof thr[0]), n_cls);
    };
    cls_fn = cls;
    {
        const char* e = getenv("SPDP_SEED_SCOUT");
        const int mode = e ? atoi(e) : 1;       // (mode 2, every request handed over, costs this path 2.4 x: 10 000 proteins of the drop-in 8.1 -> 19.4 s; mode 1: 7.7 s)
        slow_class = (mode && n_cls >= 3 && sp->wilip && !(src && src->units) && !getenv("SPDP_SEED_TRACE")) ? (mode == 2 ? 0 : n_cls - 1) : -1;
        if (slow_class == 0) { ws.all_in_flight = true; ws.last_class_waits = true; }
        // every walk in flight when the mapping budget allows: a protein walk's requests are short and few-rowed (69 % of them 8 .. 15 query
        // rows), a batch lasts 30 - 80 ms whatever its size, and a call is a chain of such batches -- with half the walks in flight the
        // chain is twice as long (20 000 loci: alignment 2.6 -> 2.0 s)
        ws.all_in_flight = true;
    }
    if (!ws.run(n_probs, walk, device, class_of_lane, cls)) { ctx->err = "the seeded path could not allocate a stack for a walk"; rc = -1; }
    us_walks += us_since(t_idle);
    if (getenv("SPDP_SEED_VERBOSE"))
        for (int l = 0; l < n_lanes; ++l)
            fprintf(stderr, "[seeded] lane %d (class %d): %lld batches, %.2f ms each, %.0f requests each\n", l, class_of_lane[l], (long long) lane_n[l],
                    lane_n[l] ? lane_us[l] / 1e3 / lane_n[l] : 0.0, lane_n[l] ? (double) lane_req[l] / lane_n[l] : 0.0);
    if (getenv("SPDP_SEED_VERBOSE"))
        fprintf(stderr, "[seeded] %d problems: upload %.3f s, walks alone %.3f s, device batches %.3f s (summed over the lanes), in all %.3f s\n", n_probs,
                us_upload / 1e6, us_walks / 1e6, us_device / 1e6, us_since(t_begin) / 1e6);
    if (shape_stats) {
        static const char* name[6] = {"< 8", "8 .. 15", "16 .. 31", "32 .. 63", "64 .. 127", ">= 128"};
        for (int b = 0; b < 6; ++b)
            fprintf(stderr, "[seeded] requests of %s query rows: %lld, %.3g anti-diagonals, %.3g cells\n", name[b], (long long) shape_n[b], (double) shape_steps[b], (double) shape_cells[b]);
    }
    spdh_store_close(st);
    ctx->seed_stats[0] = n_batches; ctx->seed_stats[1] = n_kind[0]; ctx->seed_stats[2] = n_kind[1] + n_kind[3];
    ctx->seed_stats[3] = n_cut; ctx->seed_stats[4] = n_wilip.load(); ctx->seed_stats[5] = n_probs;
    ctx->seed_stats[6] = us_upload; ctx->seed_stats[7] = us_walks; ctx->seed_stats[8] = us_device; ctx->seed_stats[9] = us_hand;
    ctx->seed_stats[10] = us_since(t_begin);
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

extern "C" int spdp_seeded_phase_marks(const SpdpContext* ctx, int q, const SpdpPhaseMark** marks)
{
    if (marks) *marks = nullptr;
    if (!ctx || q < 0 || q >= (int) ctx->seed_marks.size() || ctx->seed_marks[q].empty()) return 0;
    if (marks) *marks = ctx->seed_marks[q].data();
    return (int) ctx->seed_marks[q].size();
}

extern "C" int spdp_wilip(const SpdpWilipModel* model, const SpdpProblem* p, const SpdpScoring* sc,
                          const SpdpProblemH* ph, const SpdpScoringH* sch,
                          int32_t level, const int32_t span[4], const int32_t exg[2], int32_t** flat)
{
    if (!model || !span || !flat || (!p == !ph) || (p && !sc) || (ph && !sch) || level < -1 || level > 2) return -1;
    spdp_hsp::Seqs pr; spdp_hsp::GapCosts gc;
    if (p) { pr = {p->a, p->a_len, span[0], span[1], exg ? exg[0] : 0, exg ? exg[1] : 0, p->b, p->b_len, span[2], span[3], 1, nullptr, nullptr, nullptr};
             gc = {sc->intpen, sc->intpen_len, sc->gop, sc->gep, sc->lgop, sc->lgep, sc->codonk1}; }
    else { pr = {ph->a, ph->a_len, span[0], span[1], exg ? exg[0] : 0, exg ? exg[1] : 0, ph->b, ph->b_len, span[2], span[3], 3, ph->sigS, ph->sigE, ph->sigT};
           gc = {sch->intpen, sch->intpen_len, sch->gop, sch->gep, sch->lgop, sch->lgep, sch->codonk1}; }
    if (!pr.a || !pr.b || !gc.intpen || gc.intpen_len <= 0) return -1;
    if (pr.a_left < 0 || pr.a_right > pr.a_len || pr.a_left > pr.a_right || pr.b_left < 0 || pr.b_right > pr.b_len || pr.b_left > pr.b_right) return -1;
    std::vector<spdp_hsp::Unit> us;
    spdp_hsp::search(model, pr, gc, level, us);
    std::vector<int32_t> f;
    spdp_hsp::flatten(us, f);
    *flat = (int32_t*) malloc(sizeof(int32_t) * f.size());
    if (!*flat) return -1;
    memcpy(*flat, f.data(), sizeof(int32_t) * f.size());
    return (int) f.size();
}
