// spdp_async.cpp -- submit / wait form of the batched alignment calls (include/spdp.h).
//
// A worker thread runs the synchronous entry point and owns the context until spdp_wait() returns;
// callers overlap their own work (reading the next batch, building its SpdpProblem array, rescoring
// the previous one) with the GPU, or keep several contexts -- one per GPU, or several per GPU -- busy
// from one thread.  The reference has no counterpart: its worker threads each call alignS_ng
// synchronously (src/spaln.cc:1363-1387).
#include <hip/hip_runtime.h>
#include <atomic>
#include <thread>
#include "../../include/spdp.h"
#include "spdp_internal.h"

struct SpdpTicket {
    std::thread worker;
    std::atomic<int> done{0};
    int rc = -1;
};

template <typename F>
static SpdpTicket* submit(SpdpContext* ctx, F&& call)
{
    if (!ctx) return nullptr;
    SpdpTicket* t = new SpdpTicket();
    t->worker = std::thread([t, ctx, call]() {
        (void) hipSetDevice(ctx->device);
        t->rc = call();
        t->done.store(1, std::memory_order_release);
    });
    return t;
}

extern "C" {

SpdpTicket* spdp_submit_align_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                                SpdpAlignment* out)
{
    return submit(ctx, [=]() { return spdp_align_s(ctx, sc, probs, n_probs, out); });
}

SpdpTicket* spdp_submit_align_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                                SpdpAlignment* out)
{
    return submit(ctx, [=]() { return spdp_align_h(ctx, sc, probs, n_probs, out); });
}

SpdpTicket* spdp_submit_homscore_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                                   int32_t* scores)
{
    return submit(ctx, [=]() { return spdp_homscore_s(ctx, sc, probs, n_probs, scores); });
}

SpdpTicket* spdp_submit_homscore_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                                   int32_t* scores)
{
    return submit(ctx, [=]() { return spdp_homscore_h(ctx, sc, probs, n_probs, scores); });
}

int spdp_poll(const SpdpTicket* t) { return t ? t->done.load(std::memory_order_acquire) : 1; }

int spdp_wait(SpdpTicket* t)
{
    if (!t) return -1;
    if (t->worker.joinable()) t->worker.join();
    const int rc = t->rc;
    delete t;
    return rc;
}

}   // extern "C"
