// spdp_collect.cpp -- SpdpCollector: turns the one-problem-at-a-time calls of many host threads into device batches.
//
// The reference's seeded path (seededS_ng -> interpolateS, src/fwd2s1.cc:2405-2672) walks the HSPs of ONE query on ONE
// CPU thread and calls lspS_ng / trcbkalignS_ng synchronously for every gap it cannot close by other means; spaln's
// thread pool (-t, src/spaln.cc:1560-1640) runs many such walks at once.  A shim that replaces lspS_ng by a device call
// would therefore issue thousands of single-problem launches.  The collector is the piece in between: every worker
// thread calls spdp_collector_align_s() with its one problem and blocks; a dispatcher thread gathers what has arrived
// (up to max_batch problems, or whatever is there max_wait_us after the first arrival), runs ONE batch through the
// ladder on the context it owns, hands each caller its result and wakes it.  Problems only borrow the caller's buffers
// for the duration of its call.
#include "spdp_internal.h"
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

struct SpdpCollector {
    SpdpContext* ctx = nullptr;
    SpdpScoring sc;
    std::vector<int16_t> intpen;                // own copy of the length-penalty table
    SpdpSignalModel model;                      // ... and of the signal model with its two matrices (SpdpScoring::sigmodel)
    std::vector<float> mtx5, mtx3;
    int in_flight = 0;                          // callers inside spdp_collector_align_s
    int max_batch = 256, max_wait_us = 200, raw = 0;
    bool protein = false;                       // spdp_collector_create_h: requests are SpdpProblemH, sch is the bundle
    SpdpScoringH sch;
    struct Req { const void* p; SpdpAlignment* out; int rc = 0; bool done = false; };
    int run(const SpdpProblem* probs, int n, SpdpAlignment* outs)
    {
        return raw ? spdp_lsp_s(ctx, &sc, probs, n, outs) : spdp_align_s(ctx, &sc, probs, n, outs);
    }
    int run(const SpdpProblemH* probs, int n, SpdpAlignment* outs)
    {
        return raw ? spdp_lsp_h(ctx, &sch, probs, n, outs) : spdp_align_h(ctx, &sch, probs, n, outs);
    }
    template <typename P> void serve(std::vector<Req*>& take, std::vector<SpdpAlignment>& outs, std::vector<int>& each, std::string& why)
    {
        std::vector<P> probs(take.size());
        for (size_t i = 0; i < take.size(); ++i) probs[i] = *static_cast<const P*>(take[i]->p);
        const int rc = run(probs.data(), (int) probs.size(), outs.data());
        each.assign(take.size(), rc);
        why = rc ? ctx->err : std::string();
        if (rc < 0 && take.size() > 1) {
            // one caller's malformed problem must not fail the callers that happened to share its batch: once more, one by one
            for (size_t i = 0; i < take.size(); ++i) {
                outs[i].score = SPDP_NEVSEL; outs[i].n_skl = 0; outs[i].skl = nullptr; outs[i].flags = 0; outs[i].reserved = 0;
                each[i] = run(&probs[i], 1, &outs[i]);
                if (each[i]) why = ctx->err;
            }
        }
    }
    std::mutex mu;
    std::condition_variable cv_req, cv_done;
    std::deque<Req*> queue;
    std::chrono::steady_clock::time_point first_arrival;
    bool stop = false;
    std::thread worker;
    std::string err;
    int64_t n_requests = 0, n_batches = 0, largest = 0;

    void loop()
    {
        (void) hipSetDevice(ctx->device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_req.wait(lk, [&] { return stop || !queue.empty(); });
            if (queue.empty()) { if (stop) return; continue; }
            // a batch closes when it is full, when its first request has waited long enough, or at shutdown
            const auto deadline = first_arrival + std::chrono::microseconds(max_wait_us);
            cv_req.wait_until(lk, deadline, [&] { return stop || (int) queue.size() >= max_batch; });
            std::vector<Req*> take;
            while (!queue.empty() && (int) take.size() < max_batch) { take.push_back(queue.front()); queue.pop_front(); }
            if (!queue.empty()) first_arrival = std::chrono::steady_clock::now();
            lk.unlock();
            std::vector<SpdpAlignment> outs(take.size());
            std::vector<int> each;
            std::string why;
            if (protein) serve<SpdpProblemH>(take, outs, each, why);
            else serve<SpdpProblem>(take, outs, each, why);
            lk.lock();
            if (!why.empty()) err = why;
            ++n_batches; n_requests += (int64_t) take.size(); largest = std::max<int64_t>(largest, (int64_t) take.size());
            for (size_t i = 0; i < take.size(); ++i) {
                *take[i]->out = outs[i];        // ownership of skl passes to the caller (spdp_free_alignments)
                // rc 1 = some queries of the batch came back without an alignment: only those report it
                take[i]->rc = each[i] < 0 ? -1 : ((each[i] == 1 && !outs[i].skl) ? 1 : 0);
                take[i]->done = true;
            }
            cv_done.notify_all();
        }
    }
};

extern "C" SpdpCollector* spdp_collector_create(SpdpContext* ctx, const SpdpScoring* sc, int max_batch, int max_wait_us, int raw_records)
{
    if (!ctx || !sc) return nullptr;
    SpdpCollector* c = new SpdpCollector;
    c->ctx = ctx; c->sc = *sc;
    if (sc->intpen && sc->intpen_len > 0) { c->intpen.assign(sc->intpen, sc->intpen + sc->intpen_len); c->sc.intpen = c->intpen.data(); }
    if (sc->sigmodel) {                         // the caller may free its model once the collector exists
        c->model = *sc->sigmodel;
        if (sc->sigmodel->mtx5) { c->mtx5.assign(sc->sigmodel->mtx5, sc->sigmodel->mtx5 + (size_t) c->model.cols5 * c->model.rows); c->model.mtx5 = c->mtx5.data(); }
        if (sc->sigmodel->mtx3) { c->mtx3.assign(sc->sigmodel->mtx3, sc->sigmodel->mtx3 + (size_t) c->model.cols3 * c->model.rows); c->model.mtx3 = c->mtx3.data(); }
        c->sc.sigmodel = &c->model;
    }
    c->max_batch = std::max(1, max_batch); c->max_wait_us = std::max(0, max_wait_us); c->raw = raw_records ? 1 : 0;
    c->worker = std::thread([c] { c->loop(); });
    return c;
}

extern "C" SpdpCollector* spdp_collector_create_h(SpdpContext* ctx, const SpdpScoringH* sc, int max_batch, int max_wait_us, int raw_records)
{
    if (!ctx || !sc) return nullptr;
    SpdpCollector* c = new SpdpCollector;
    c->ctx = ctx; c->protein = true; c->sch = *sc;
    if (sc->intpen && sc->intpen_len > 0) { c->intpen.assign(sc->intpen, sc->intpen + sc->intpen_len); c->sch.intpen = c->intpen.data(); }
    c->max_batch = std::max(1, max_batch); c->max_wait_us = std::max(0, max_wait_us); c->raw = raw_records ? 1 : 0;
    c->worker = std::thread([c] { c->loop(); });
    return c;
}

extern "C" void spdp_collector_destroy(SpdpCollector* c)
{
    if (!c) return;
    { std::lock_guard<std::mutex> g(c->mu); c->stop = true; }
    c->cv_req.notify_all();
    if (c->worker.joinable()) c->worker.join();
    {   // callers the last batch woke are still on their way out of cv_done.wait: the mutex must outlive them
        std::unique_lock<std::mutex> lk(c->mu);
        c->cv_done.wait(lk, [&] { return c->in_flight == 0; });
    }
    delete c;
}

static int collector_call(SpdpCollector* c, const void* p, SpdpAlignment* out);
extern "C" int spdp_collector_align_s(SpdpCollector* c, const SpdpProblem* p, SpdpAlignment* out)
{
    if (!c || c->protein) return -1;
    return collector_call(c, p, out);
}
extern "C" int spdp_collector_align_h(SpdpCollector* c, const SpdpProblemH* p, SpdpAlignment* out)
{
    if (!c || !c->protein) return -1;
    return collector_call(c, p, out);
}
static int collector_call(SpdpCollector* c, const void* p, SpdpAlignment* out)
{
    if (!c || !p || !out) return -1;
    SpdpCollector::Req r;
    r.p = p; r.out = out;
    std::unique_lock<std::mutex> lk(c->mu);
    if (c->stop) return -1;
    ++c->in_flight;
    if (c->queue.empty()) c->first_arrival = std::chrono::steady_clock::now();
    c->queue.push_back(&r);
    c->cv_req.notify_all();
    c->cv_done.wait(lk, [&] { return r.done; });
    if (--c->in_flight == 0) c->cv_done.notify_all();       // (spdp_collector_destroy may be waiting for the last caller)
    return r.rc;
}

extern "C" const char* spdp_collector_last_error(const SpdpCollector* c) { return c ? c->err.c_str() : "null collector"; }

extern "C" int spdp_collector_stats(SpdpCollector* c, int64_t* n_requests, int64_t* n_batches, int64_t* largest_batch)
{
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->mu);
    if (n_requests) *n_requests = c->n_requests;
    if (n_batches) *n_batches = c->n_batches;
    if (largest_batch) *largest_batch = c->largest;
    return 0;
}
