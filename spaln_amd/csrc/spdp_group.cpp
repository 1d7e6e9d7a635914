// spdp_group.cpp -- several GPUs of one node behind one handle (include/spdp.h, "device groups").
//
// The reference is one multi-threaded process (spaln -t N: a master hands whole queries to worker threads,
// src/spaln.cc:1389-1468); the drop-in counterpart is one process that owns every GPU of the node.  A group
// holds one context per device; a batched call shards the query list by DP cells (longest-processing-time rule;
// problems are independent, SURVEY.md 8e: no data-path exchange, no collective), runs every shard on its own device
// from its own host thread, and the results land in the caller's arrays in query order.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include "../../include/spdp.h"
#include "spdp_internal.h"

struct SpdpGroup {
    std::vector<SpdpContext*> ctx;
    std::string err;
    std::vector<int> shard_of;          // member that ran problem i in the last call (spdp_group_last_shards)
};

namespace {
// cost-balanced shards (longest-processing-time rule over the DP cells of every problem; the rule of
// spaln_amd/shard.py: balanced_shards): windows differ by a factor of two and more, and the slowest member bounds the
// call.  idx[r] = the problems of member r, in caller order.
std::vector<std::vector<int>> balanced(const std::vector<int64_t>& cost, int w)
{
    std::vector<int> order(cost.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int) i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost[x] > cost[y]; });
    std::vector<int64_t> load(w, 0);
    std::vector<std::vector<int>> idx(w);
    for (int i : order) {
        int r = 0;
        for (int k = 1; k < w; ++k) if (load[k] < load[r]) r = k;
        idx[r].push_back(i);
        load[r] += cost[i];
    }
    for (auto& v : idx) std::sort(v.begin(), v.end());
    return idx;
}

// P = SpdpProblem / SpdpProblemH, R = the per-problem result type: member r gets its problems gathered into one array
template <typename P, typename R, typename F>
int fan_out(SpdpGroup* g, const P* probs, int n, const std::vector<int64_t>& cost, R* res, F&& call)
{
    if (!g || g->ctx.empty()) return -1;
    const int w = (int) g->ctx.size();
    const std::vector<std::vector<int>> idx = balanced(cost, w);
    g->shard_of.assign(n, 0);
    std::vector<int> rc(w, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < w; ++r) {
        if (idx[r].empty()) continue;
        for (int i : idx[r]) g->shard_of[i] = r;
        th.emplace_back([&, r]() {
            (void) hipSetDevice(g->ctx[r]->device);
            const int cnt = (int) idx[r].size();
            std::vector<P> mine(cnt);
            std::vector<R> out(cnt);
            for (int k = 0; k < cnt; ++k) { mine[k] = probs[idx[r][k]]; out[k] = res[idx[r][k]]; }
            rc[r] = call(g->ctx[r], mine.data(), cnt, out.data());
            for (int k = 0; k < cnt; ++k) res[idx[r][k]] = out[k];
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

static std::vector<int64_t> costs_s(const SpdpScoring* sc, const SpdpProblem* probs, int n)
{
    std::vector<int64_t> c(std::max(n, 0));
    for (int i = 0; i < n; ++i) { SpdpWindow w; spdp_stripe(&probs[i], sc->sh, &w); c[i] = spdp_cells(&probs[i], &w); }
    return c;
}
static std::vector<int64_t> costs_h(const SpdpScoringH* sc, const SpdpProblemH* probs, int n)
{
    std::vector<int64_t> c(std::max(n, 0));
    for (int i = 0; i < n; ++i) { SpdpWindow w; spdp_stripe31(&probs[i], sc->sh, &w); c[i] = spdp_cells_h(&probs[i], &w); }
    return c;
}

int spdp_group_homscore_s(SpdpGroup* g, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs, int32_t* scores)
{
    return fan_out(g, probs, n_probs, costs_s(sc, probs, n_probs), scores, [=](SpdpContext* c, const SpdpProblem* p, int cnt, int32_t* o) {
        return spdp_homscore_s(c, sc, p, cnt, o);
    });
}

int spdp_group_align_s(SpdpGroup* g, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs, SpdpAlignment* out)
{
    return fan_out(g, probs, n_probs, costs_s(sc, probs, n_probs), out, [=](SpdpContext* c, const SpdpProblem* p, int cnt, SpdpAlignment* o) {
        return spdp_align_s(c, sc, p, cnt, o);
    });
}

int spdp_group_homscore_h(SpdpGroup* g, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs, int32_t* scores)
{
    return fan_out(g, probs, n_probs, costs_h(sc, probs, n_probs), scores, [=](SpdpContext* c, const SpdpProblemH* p, int cnt, int32_t* o) {
        return spdp_homscore_h(c, sc, p, cnt, o);
    });
}

int spdp_group_align_h(SpdpGroup* g, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs, SpdpAlignment* out)
{
    return fan_out(g, probs, n_probs, costs_h(sc, probs, n_probs), out, [=](SpdpContext* c, const SpdpProblemH* p, int cnt, SpdpAlignment* o) {
        return spdp_align_h(c, sc, p, cnt, o);
    });
}

// which member ran problem i in the last group call (cost-balanced: spdp_cells per problem, longest first)
int spdp_group_last_shards(const SpdpGroup* g, int32_t* member, int n)
{
    if (!g || !member) return -1;
    for (int i = 0; i < n && i < (int) g->shard_of.size(); ++i) member[i] = g->shard_of[i];
    return (int) g->shard_of.size();
}

}   // extern "C"
