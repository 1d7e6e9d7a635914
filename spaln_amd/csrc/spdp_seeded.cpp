// spdp_seeded.cpp -- alignS_ng with seeding on (algmode.qck = 1 .. 3): the walks of a batch of queries over their HSPs,
// with every DP call of every walk served by the device in common batches.
//
// What it mirrors (ogotoh/spaln v3.0.7): Aln2s1::globalS_ng -> seededS_ng -> interpolateS (src/fwd2s1.cc:2587-2694,
// 2405-2539); the walk itself is spdp_walk.h (Walk<CdnaPath>).  The reference runs one walk per worker thread and calls its DP
// engines synchronously from deep inside it (spaln -t, src/spaln.cc:1389-1468).  Here every walk of a call runs on a
// fiber, thousands in flight on the host's cores (spdp_seeded_rv.h); a walk that reaches lspS_ng / trcbkalignS_ng parks
// its request and yields; the parked requests, sorted by how long their sweeps will take, run as sets of device launches
// on the resident inputs of the batch (spdp_run_requests: the same rounds as spdp_align_s -- linear-space sweeps, slab
// tracebacks, walks), one dispatcher lane per latency class, and their owners become runnable again.  No DP cell of a
// request is computed on the host.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <ctime>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "spdp_internal.h"
#include "spdp_walk.h"
#include "spdp_hsp_host.h"
#include "spdp_seeded_rv.h"

namespace {
using namespace spdp_seed;

struct DeviceBackend : DpBackend {
    Fiber* fiber; int query; const SpdpHspSource* src;
    bool failed = false;
    int flags = 0;                              // of all DP calls of the walk
    std::atomic<int64_t>* n_wilip;
    std::atomic<int64_t>* ns_cb = nullptr;
    // scout pass (pass 0) and the walk proper (pass 1) over one cache; slow(request): the request belongs to the longest class
    RequestCache* cache = nullptr;
    int pass = 1;
    bool after_dummy = false;
    std::atomic<int64_t>* n_hit = nullptr; std::atomic<int64_t>* n_miss = nullptr;      // the walk proper: requests the scout had / had not asked for
    const std::function<bool(const Parked&)>* slow = nullptr;
    int take(const Parked& p, std::vector<SpdpSkl>& rec)
    {
        if (p.failed) { failed = true; return SPDP_NEVSEL; }
        flags |= p.flags;
        rec.insert(rec.end(), p.rec.begin(), p.rec.end());
        return p.score;
    }
    int park(int kind, const Span& s, const SpdpWindow& w, const int* cut, std::vector<SpdpSkl>& rec)
    {
        int key[14];
        if (cache) {
            RequestCache::make_key(key, kind, s, w, cut);
            if (RequestCache::Entry* e = cache->find(key)) {
                if (e->p.async && !e->p.done) {
                    if (pass == 0) return SPDP_NEVSEL;          // (the scout does not wait for what it handed over)
                    fiber->wait_for(&e->p);
                }
                if (pass && n_hit) ++*n_hit;
                return take(e->p, rec);
            }
            if (pass && n_miss) ++*n_miss;
        }
        if (cache && pass == 0) {
            // the scout's own doing: a gap whose lspS_ng it did not wait for "found nothing", and the walk bridges it by the cut-range
            // traceback of shortcutS_ng -- a request the walk proper will not make once the real result is in
            const bool bridge = after_dummy && kind == 2;
            after_dummy = false;
            if (bridge) return SPDP_NEVSEL;
            RequestCache::Entry* e = cache->add(key);
            Parked& p = e->p;
            p.query = query; p.kind = kind; p.s = s; p.w = w;
            if (cut) { p.cut[0] = cut[0]; p.cut[1] = cut[1]; }
            if ((*slow)(p)) { fiber->submit(&p); after_dummy = kind == 0; return SPDP_NEVSEL; }     // on its way; the scout goes on as if nothing had been found
            fiber->park(&p);
            return take(p, rec);
        }
        Parked p;
        p.query = query; p.kind = kind; p.s = s; p.w = w;
        if (cut) { p.cut[0] = cut[0]; p.cut[1] = cut[1]; }
        fiber->park(&p);                        // back when the request has been served
        return take(p, rec);
    }
    int lsp(const Span& s, const SpdpWindow& w, std::vector<SpdpSkl>& rec) override { return park(0, s, w, nullptr, rec); }
    int trcbk(const Span& s, const SpdpWindow& w, bool, const int* cut, std::vector<SpdpSkl>& rec) override
    {
        return park(cut ? 2 : 1, s, w, cut, rec);      // (the cDNA engines have no intron switch)
    }
    const SpdpWilipModel* wm = nullptr; const SpdpProblem* prob = nullptr; const SpdpScoring* scp = nullptr; int codonk1 = 0;
    bool wilip(int level, const Span& s, std::vector<Unit>& units) override
    {
        if ((!src || !src->units) && wm) {      // the library's own HSP search (spdp_hsp_host.h)
            int skey[9] = {level, s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr};
            if (cache) {
                for (auto& e : cache->searches) if (!memcmp(e->key, skey, sizeof skey)) { units = e->units; if (pass) ++*n_wilip; return e->ok; }
                const bool ok = own_search(level, s, units);
                cache->searches.emplace_back(new RequestCache::Search);
                memcpy(cache->searches.back()->key, skey, sizeof skey);
                cache->searches.back()->ok = ok; cache->searches.back()->units = units;
                if (pass) ++*n_wilip;
                return ok;
            }
            ++*n_wilip;
            return own_search(level, s, units);
        }
        return callback_search(level, s, units);
    }
    bool own_search(int level, const Span& s, std::vector<Unit>& units)
    {
        {
            const spdp_hsp::Seqs pr = {prob->a, prob->a_len, s.al, s.ar, s.a_exgl, s.a_exgr, prob->b, prob->b_len, s.bl, s.br, 1,
                                        nullptr, nullptr, nullptr};
            const spdp_hsp::GapCosts gc = {scp->intpen, scp->intpen_len, scp->gop, scp->gep, scp->lgop, scp->lgep, codonk1};
            std::vector<spdp_hsp::Unit> us;
            spdp_hsp::search(wm, pr, gc, level, us);
            std::vector<int32_t> flat;
            spdp_hsp::flatten(us, flat);
            return parse_units(flat.data(), (int32_t) flat.size(), units);
        }
    }
    bool callback_search(int level, const Span& s, std::vector<Unit>& units)
    {
        if (!src || !src->units) return false;
        const int32_t span[8] = {s.al, s.ar, s.bl, s.br, s.a_exgl, s.a_exgr, s.b_exgl, s.b_exgr};
        const int32_t* flat = nullptr; int32_t n = 0;
        ++*n_wilip;
        timespec t0, t1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0);
        const int urc = src->units(src->user, query, level, span, &flat, &n);
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1);
        if (ns_cb) *ns_cb += (int64_t) (t1.tv_sec - t0.tv_sec) * 1000000000 + (t1.tv_nsec - t0.tv_nsec);
        if (urc || !flat) return false;
        const bool ok = parse_units(flat, n, units);
        if (src->release) src->release(src->user, query, flat);
        return ok;
    }
};
}   // namespace

// the walks of `n_probs` (query, strand) pairs: scores[i] = what globalS_ng's seededS_ng returns, recs[i] = its record file
// (dummy record first), status[i] = 0, or 1 / 2 for a walk that was not served
// SPDP_SEED_DUMP=<file> SPDP_SEED_DUMP_HASH=<fnv1a of the query's codes>: what this call was handed for that query, as one flat
// file of named blocks (a development aid: two callers -- the reference's CLI on the library, spdp_map_align_s -- can be held
// against each other input by input)
static void dump_inputs(const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpProblem* probs, int n_probs,
                        const SpdpJuxt* const* hsps, const int32_t* n_hsps, const int32_t* lowest_level)
{
    const char* path = getenv("SPDP_SEED_DUMP"); const char* hs = getenv("SPDP_SEED_DUMP_HASH");
    if (!path || !hs) return;
    std::vector<uint32_t> wants;                        // (comma-separated)
    for (const char* c = hs; *c; ) { char* e; wants.push_back((uint32_t) strtoul(c, &e, 0)); c = *e ? e + 1 : e; }
    for (int q = 0; q < n_probs; ++q) {
        const SpdpProblem& P = probs[q];
        uint32_t h = 2166136261u;
        for (int i = 0; i < P.a_len; ++i) h = (h ^ P.a[i]) * 16777619u;
        if (std::find(wants.begin(), wants.end(), h) == wants.end()) continue;
        FILE* f = fopen(path, "ab");
        if (!f) return;
        auto blk = [&](const char* name, const void* p, size_t n) {
            char nm[32] = {0}; strncpy(nm, name, 31);
            const uint64_t len = p ? n : 0;
            fwrite(nm, 1, 32, f); fwrite(&len, 8, 1, f); if (len) fwrite(p, 1, len, f);
        };
        const size_t N = (size_t) P.b_len + 1;
        SpdpScoring s0 = *sc; s0.intpen = nullptr; s0.sigmodel = nullptr;
        SpdpSeedParams p0 = *sp; p0.wilip = nullptr;
        SpdpProblem q0 = P; q0.a = q0.b = nullptr; q0.sig5 = q0.sig3 = nullptr; q0.cano5 = q0.cano3 = q0.dinc = nullptr; q0.cip = nullptr; q0.phs5 = q0.phs3 = nullptr;
        blk("scoring", &s0, sizeof s0); blk("intpen", sc->intpen, 2 * (size_t) sc->intpen_len);
        blk("seed", &p0, sizeof p0); blk("problem", &q0, sizeof q0);
        blk("a", P.a, (size_t) P.a_len); blk("b", P.b, (size_t) P.b_len);
        blk("sig5", P.sig5, 2 * N); blk("sig3", P.sig3, 2 * N); blk("cano5", P.cano5, N); blk("cano3", P.cano3, N); blk("dinc", P.dinc, N);
        blk("phs5", P.phs5, N); blk("phs3", P.phs3, N);
        const int nh = (hsps && n_hsps && hsps[q]) ? n_hsps[q] : 0;
        blk("hsps", nh ? hsps[q] : nullptr, sizeof(SpdpJuxt) * (size_t) (nh + 1));
        const int lv = lowest_level ? lowest_level[q] : 0;
        blk("lowest", &lv, 4);
        if (sp->wilip) blk("wilip", sp->wilip, sizeof(SpdpWilipModel));
        fclose(f);
    }
}

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
    dump_inputs(sc, sp, probs, n_probs, hsps, n_hsps, lowest_level);
    const auto t_begin = std::chrono::steady_clock::now();
    auto us_since = [](std::chrono::steady_clock::time_point t) {
        return (int64_t) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count(); };
    DevStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    const int64_t us_upload = us_since(t_begin);
    int64_t us_walks = 0, us_device = 0, us_hand = 0;

    std::atomic<int64_t> n_wilip{0};
    std::atomic<int64_t> ns_bind{0}, ns_run{0}, ns_cb{0};
    auto cpu_ns = [] { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return (int64_t) t.tv_sec * 1000000000 + t.tv_nsec; };
    scores.assign(n_probs, SPDP_NEVSEL);
    recs.assign(n_probs, std::vector<SpdpSkl>());
    status.assign(n_probs, 0);                          // 1: the walk met a state it does not serve, 2: a request failed
    // A walk's slow requests -- its terminal stretches: a few query rows against tens of thousands of columns, 100 - 200 ms as
    // one wave each -- come one after the other, head first, tail last, with the inner gaps in between: the call
    // waits for their sum.  So a walk runs twice.  The SCOUT hands every slow request over without waiting (and goes on as if
    // it had found nothing there), waits for the fast ones as usual and keeps all results by what defines a request; the walk
    // proper then finds its requests answered or on their way -- head and tail are in flight together -- and whatever the scout
    // did not ask for (its path may differ behind a request it did not wait for) is asked for as before.  Results are those of
    // the second run alone.  SPDP_SEED_SCOUT=0: one run; = 1 / 2: see where the mode is chosen below.
    std::atomic<int64_t> n_hit{0}, n_miss{0};
    std::function<int(const Parked&)> cls_fn;         // (set below, before any walk starts)
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
                DeviceBackend be;
                be.fiber = &fb; be.query = q; be.src = src; be.n_wilip = &n_wilip; be.ns_cb = &ns_cb;
                be.wm = sp->wilip; be.prob = &probs[q]; be.scp = sc; be.codonk1 = sp->codonk1;       // (the walk's own GapPenalty reads it there)
                if (two) { be.cache = &cache; be.pass = pass; be.slow = &is_slow; be.n_hit = &n_hit; be.n_miss = &n_miss; }
                SeedWalk w;
                const int nh = (hsps && n_hsps && hsps[q]) ? n_hsps[q] : 0;
                const int64_t tb0 = cpu_ns();
                SpdpProblem bound = probs[q];
                if (two && pass && !cache.phs5.empty()) { bound.phs5 = cache.phs5.data(); bound.phs3 = cache.phs3.data(); }     // (derived by the scout)
                if (!bind_problem(w, sc, sp, &bound, nh ? hsps[q] : nullptr, nh, lowest_level ? lowest_level[q] : 0)) { status[q] = 1; break; }
                ns_bind += cpu_ns() - tb0;
                w.dp = &be;
                const SpdpProblem& p = probs[q];
                const Span whole = {p.a_left, p.a_right, p.b_left, p.b_right, p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr};
                const int score = w.run(whole);
                if (pass == 0) {                    // (marks the walk sets go to a list of edits, not into the arrays)
                    fb.flush();                     // (what the scout handed over goes to the dispatchers now, under one lock)
                   
                    if (!probs[q].phs5 && !w.phs5.own.empty()) { cache.phs5.swap(w.phs5.own); cache.phs3.swap(w.phs3.own); }
                    continue;
                }
                scores[q] = score;
                recs[q].swap(w.rec);
                status[q] = be.failed ? 2 : (w.unsupported ? 1 : 0);
                if (be.flags & SPDP_ALN_LEFT_EDGE) status[q] |= 16;
            } catch (...) { if (pass) status[q] = 2; }  // (out of memory inside one walk: that query comes back without an alignment)
        }
        // what the scout handed over and the walk never asked for is still written to when it is served: not before that may the cache go
        for (auto& e : cache.all) if (e->p.async && !e->p.done) fb.wait_for(&e->p);
    };

    std::atomic<int> rc{0};
    int64_t n_batches = 0, n_kind[3] = {0, 0, 0};
    auto t_idle = std::chrono::steady_clock::now();
    std::mutex stats_mu;
    const std::vector<int> class_of_lane = lanes_per_class(n_probs, {2, 1, 1});
    const int n_lanes = (int) class_of_lane.size();
    if (n_lanes > 1 && !spdp_lane(ctx, n_lanes - 1)) return -1;       // (created here, on one thread)
    int busy_lanes = 0;
    std::vector<int64_t> lane_n(n_lanes, 0), lane_us(n_lanes, 0), lane_req(n_lanes, 0);
    auto device = [&](std::vector<Parked*>& take, int lane) {
        (void) hipSetDevice(ctx->device);
        { std::lock_guard<std::mutex> g(stats_mu); if (!busy_lanes++) us_walks += us_since(t_idle); }
        auto t0 = std::chrono::steady_clock::now();
        if (getenv("SPDP_SEED_TIMELINE")) fprintf(stderr, "[seeded] t = %.3f s: lane %d starts %zu requests\n", us_since(t_begin) / 1e6, lane, take.size());
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
        }
        if (const char* df = getenv("SPDP_SEED_DUMP")) {      // request shapes of every batch, for tuning: batch kind rows cols lw up cut query a_left b_left lane
            std::lock_guard<std::mutex> g(stats_mu);
            if (FILE* f = fopen(df, "a")) {
                for (int k = 0; k < m; ++k) {
                    const Parked& q = *take[k];
                    fprintf(f, "%lld %d %d %d %d %d %d %d %d %d %d\n", (long long) n_batches, q.kind, q.s.ar - q.s.al, q.s.br - q.s.bl, q.w.lw, q.w.up, q.cut[1] - q.cut[0],
                            q.query, q.s.al, q.s.bl, lane);
                }
                fclose(f);
            }
        }
        std::vector<SpdpAlignment> res(m);
        const SpdpRequests rq = {parents.data(), wins.data(), kinds.data(), cuts.data()};
        SpdpContext* lc = spdp_lane(ctx, lane);
        int brc = rc < 0 ? -1 : spdp_run_requests(lc, &st, rp.data(), m, &rq, res.data());
        const int64_t us_dev = us_since(t0);
        t0 = std::chrono::steady_clock::now();
        if (brc < 0) { rc = -1; std::lock_guard<std::mutex> g(stats_mu); if (lc != ctx) ctx->err = lc->err; }    // the walks still have to be let go: every request fails from here on
        for (int k = 0; k < m; ++k) {
            Parked& q = *take[k];
            if (brc < 0 || res[k].n_skl < 0) q.failed = true;
            else {
                q.score = res[k].score; q.flags = res[k].flags;
                if (res[k].n_skl > 0) q.rec.assign(res[k].skl, res[k].skl + res[k].n_skl);
            }
        }
        if (brc >= 0) spdp_free_alignments(res.data(), m);
        std::lock_guard<std::mutex> g(stats_mu);
        ++n_batches;
        lane_n[lane] += 1; lane_us[lane] += us_dev; lane_req[lane] += m;
        for (int k = 0; k < m; ++k) ++n_kind[take[k]->kind];
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
    // its batches are bound by launch and read-back latency, not by the device
    const int n_cls = class_of_lane.back() + 1;
    int64_t scalar_w = 8;
    if (const char* e = getenv("SPDP_SEED_SCALAR_W")) scalar_w = std::max(1, atoi(e));      // (tuning)
    auto cls = [n_cls, scalar_w](const Parked& q) {
        const int rows = q.s.ar - q.s.al;
        const int64_t cols = std::max<int64_t>(0, (int64_t) std::min(q.s.br - q.s.bl, q.w.up - q.w.lw + rows) - (q.cut[1] > q.cut[0] ? q.cut[1] - q.cut[0] : 0));
        const int64_t steps = (rows < 8 || q.kind == 2) ? scalar_w * (cols + rows) : (int64_t) ((rows + 63) / 64) * cols;
        static const int64_t thr[] = {1500, 6000, 20000};            // (four classes measured: no gain over three; the defaults give lanes to the first three)
        return latency_class(steps, thr, (int) (sizeof thr / sizeof thr[0]), n_cls);
    };
    cls_fn = cls;
    {   // the scout pays where a call has slow requests at all: several classes, and the HSP searches the library's own (a
        // caller's callback would be asked twice)
        const char* e = getenv("SPDP_SEED_SCOUT");
        // 1: only the slow class is handed over; 2: every request is -- more than twice the requests (the scout's paths behind the
        // gaps it did not wait for), worth it while the host keeps up: 5 000 walks 0.94 -> 0.79 (1) -> 0.50 s (2), 20 000: 1.31 ->
        // 0.97 -> 1.00 s, 40 000: 1.79 -> 1.40 -> 2.10 s
        const int mode = e ? atoi(e) : 2;       // (with the fibers' requests handed over in one go and the long lane started from a thousand requests on:
                                                // 20 000 walks 0.96 (1) / 0.84 s (2), 40 000 walks 1.37 / 1.19 s)
        slow_class = (mode && n_cls >= 3 && sp->wilip && !(src && src->units)) ? (mode == 2 ? 0 : n_cls - 1) : -1;
        if (slow_class == 0) { ws.all_in_flight = true; ws.last_class_waits = true; }
    }
    if (!ws.run(n_probs, walk, device, class_of_lane, cls)) { ctx->err = "the seeded path could not allocate a stack for a walk"; rc = -1; }
    us_walks += us_since(t_idle);
    if (getenv("SPDP_SEED_VERBOSE"))
        fprintf(stderr, "[seeded] host CPU: walks %.1f ms (bind %.1f, HSP callback %.1f), %d worker threads\n", ws.cpu_ns / 1e6, ns_bind.load() / 1e6, ns_cb.load() / 1e6, ws.n_threads);
    if (getenv("SPDP_SEED_VERBOSE") && slow_class >= 0)
        fprintf(stderr, "[seeded] scout: the walks proper found %lld of their requests asked for already, %lld not\n", (long long) n_hit.load(), (long long) n_miss.load());
    if (getenv("SPDP_SEED_VERBOSE"))
        for (int l = 0; l < n_lanes; ++l)
            fprintf(stderr, "[seeded] lane %d: %lld batches, %.2f ms each, %.0f requests each\n", l, (long long) lane_n[l],
                    lane_n[l] ? lane_us[l] / 1e3 / lane_n[l] : 0.0, lane_n[l] ? (double) lane_req[l] / lane_n[l] : 0.0);
    ctx->seed_stats[0] = n_batches; ctx->seed_stats[1] = n_kind[0]; ctx->seed_stats[2] = n_kind[1] + n_kind[2];
    ctx->seed_stats[3] = n_kind[2]; ctx->seed_stats[4] = n_wilip.load(); ctx->seed_stats[5] = n_probs;
    ctx->seed_stats[6] = us_upload; ctx->seed_stats[7] = us_walks; ctx->seed_stats[8] = us_device; ctx->seed_stats[9] = us_hand;
    ctx->seed_stats[10] = us_since(t_begin);
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
    for (int i = 0; i < n && i < 12; ++i) out[i] = ctx->seed_stats[i];
    return 0;
}
