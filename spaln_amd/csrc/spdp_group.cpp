// spdp_group.cpp -- several GPUs of one node behind one handle (include/spdp.h, "device groups").
//
// The reference is one multi-threaded process (spaln -t N: a master hands whole queries to worker threads,
// src/spaln.cc:1389-1468); the drop-in counterpart is one process that owns every GPU of the node.  A group
// holds one context per device; a batched call shards the query list into contiguous ranges (problems are
// independent, SURVEY.md 8e: no data-path exchange, no collective), runs every range on its own device from
// its own host thread, and the results land in the caller's arrays in query order.
#include <hip/hip_runtime.h>
#include <string>
#include <thread>
#include <vector>
#include "../../include/spdp.h"
#include "spdp_internal.h"

struct SpdpGroup {
    std::vector<SpdpContext*> ctx;
    std::string err;
};

namespace {
// contiguous, balanced slice of [0, n) for member r of w (the first members take the remainder): the same rule
// as spaln_amd/shard.py
void slice(int n, int r, int w, int* lo, int* cnt)
{
    const int base = n / w, rem = n % w;
    *lo = r * base + (r < rem ? r : rem);
    *cnt = base + (r < rem ? 1 : 0);
}

template <typename F>
int fan_out(SpdpGroup* g, int n, F&& call)
{
    if (!g || g->ctx.empty()) return -1;
    const int w = (int) g->ctx.size();
    std::vector<int> rc(w, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < w; ++r) {
        int lo, cnt;
        slice(n, r, w, &lo, &cnt);
        if (cnt == 0) continue;
        th.emplace_back([&, r, lo, cnt]() {
            (void) hipSetDevice(g->ctx[r]->device);
            rc[r] = call(g->ctx[r], lo, cnt);
        });
    }
    for (std::thread& t : th) t.join();
    int worst = 0;
    for (int r = 0; r < w; ++r) {
        if (rc[r] < 0) { g->err = "device " + std::to_string(g->ctx[r]->device) + ": " + g->ctx[r]->err; return -1; }
        worst |= rc[r];
    }
    return worst;
}
}   // namespace

extern "C" {

SpdpGroup* spdp_group_create(const int* devices, int n_devices)
{
    if (!devices || n_devices <= 0) return nullptr;
    SpdpGroup* g = new SpdpGroup();
    for (int i = 0; i < n_devices; ++i) {
        SpdpContext* c = spdp_create(devices[i]);
        if (!c) { spdp_group_destroy(g); return nullptr; }
        g->ctx.push_back(c);
    }
    return g;
}

void spdp_group_destroy(SpdpGroup* g)
{
    if (!g) return;
    for (SpdpContext* c : g->ctx) spdp_destroy(c);
    delete g;
}

int spdp_group_size(const SpdpGroup* g) { return g ? (int) g->ctx.size() : 0; }
const char* spdp_group_last_error(const SpdpGroup* g) { return g ? g->err.c_str() : "null group"; }

int spdp_group_homscore_s(SpdpGroup* g, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs, int32_t* scores)
{
    return fan_out(g, n_probs, [=](SpdpContext* c, int lo, int cnt) {
        return spdp_homscore_s(c, sc, probs + lo, cnt, scores + lo);
    });
}

int spdp_group_align_s(SpdpGroup* g, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs, SpdpAlignment* out)
{
    return fan_out(g, n_probs, [=](SpdpContext* c, int lo, int cnt) {
        return spdp_align_s(c, sc, probs + lo, cnt, out + lo);
    });
}

int spdp_group_homscore_h(SpdpGroup* g, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs, int32_t* scores)
{
    return fan_out(g, n_probs, [=](SpdpContext* c, int lo, int cnt) {
        return spdp_homscore_h(c, sc, probs + lo, cnt, scores + lo);
    });
}

int spdp_group_align_h(SpdpGroup* g, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs, SpdpAlignment* out)
{
    return fan_out(g, n_probs, [=](SpdpContext* c, int lo, int cnt) {
        return spdp_align_h(c, sc, probs + lo, cnt, out + lo);
    });
}

}   // extern "C"
