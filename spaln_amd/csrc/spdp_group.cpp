// spdp_group.cpp -- several GPUs of one node behind one handle (include/spdp.h, "device groups").
//
// The reference is one multi-threaded process (spaln -t N: a master hands whole queries to worker threads,
// src/spaln.cc:1389-1468); the drop-in counterpart is one process that owns every GPU of the node.  A group
// holds one context per device; a batched call shards the query list by DP cells (longest-processing-time rule;
// problems are independent, SURVEY.md 8e: no data-path exchange, no collective), runs every shard on its own device
// from its own host thread, and the results land in the caller's arrays in query order.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
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

// ---- the calls round 3 / 4 added: seeded alignment, rescoring, the block vote -------------------------------------------
extern "C++" {
namespace {
// a member's HSP source: the caller's, asked with the CALLER's query numbers
struct RemapSource { const SpdpHspSource* src; const int* idx; };
int remap_units(void* user, int32_t query, int32_t level, const int32_t span[8], const int32_t** flat, int32_t* n_flat)
{
    const RemapSource* r = (const RemapSource*) user;
    return r->src->units(r->src->user, r->idx[query], level, span, flat, n_flat);
}
void remap_release(void* user, int32_t query, const int32_t* flat)
{
    const RemapSource* r = (const RemapSource*) user;
    if (r->src->release) r->src->release(r->src->user, r->idx[query], flat);
}

// member r runs call(ctx, its problem numbers in caller order); results are written by the call itself (it knows the layout)
template <typename F>
int fan_out_idx(SpdpGroup* g, int n, const std::vector<int64_t>& cost, F&& call)
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
        th.emplace_back([&, r]() { (void) hipSetDevice(g->ctx[r]->device); rc[r] = call(g->ctx[r], idx[r]); });
    }
    for (std::thread& t : th) t.join();
    int worst = 0;
    for (int r = 0; r < w; ++r) {
        if (rc[r] < 0) { g->err = "device " + std::to_string(g->ctx[r]->device) + ": " + g->ctx[r]->err; return -1; }
        worst |= rc[r];
    }
    return worst;
}

template <typename SC, typename P, typename CALL>
int group_seeded(SpdpGroup* g, const SC* sc, const P* probs, int n, const std::vector<int64_t>& cost,
                 const SpdpJuxt* const* hsps, const int32_t* n_hsps, const int32_t* lowest_level, const SpdpHspSource* src,
                 SpdpAlignment* out, CALL&& call)
{
    return fan_out_idx(g, n, cost, [&](SpdpContext* c, const std::vector<int>& idx) {
        const int cnt = (int) idx.size();
        std::vector<P> mine(cnt);
        std::vector<const SpdpJuxt*> h(cnt);
        std::vector<int32_t> nh(cnt), ll(cnt);
        std::vector<SpdpAlignment> o(cnt);
        for (int k = 0; k < cnt; ++k) {
            mine[k] = probs[idx[k]];
            h[k] = hsps ? hsps[idx[k]] : nullptr;
            nh[k] = n_hsps ? n_hsps[idx[k]] : 0;
            ll[k] = lowest_level ? lowest_level[idx[k]] : 0;
        }
        RemapSource rs = {src, idx.data()};
        SpdpHspSource local = {&rs, remap_units, remap_release};
        const int rc = call(c, sc, mine.data(), cnt, hsps ? h.data() : nullptr, n_hsps ? nh.data() : nullptr,
                            lowest_level ? ll.data() : nullptr, src ? &local : nullptr, o.data());
        for (int k = 0; k < cnt; ++k) out[idx[k]] = o[k];
        return rc;
    });
}
}   // namespace
}   // extern "C++"

// alignS_ng / alignH_ng with seeding on for a batch, sharded over the group's devices by DP cells of the whole windows (the
// reference's counterpart: the -t N master / worker loop, src/spaln.cc:1389-1468); every member runs its own fiber scheduler
// and dispatcher lanes; the HSP source is asked with the caller's query numbers
int spdp_group_align_s_seeded(SpdpGroup* g, const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpProblem* probs, int n_probs,
                              const SpdpJuxt* const* hsps, const int32_t* n_hsps, const int32_t* lowest_level,
                              const SpdpHspSource* src, SpdpAlignment* out)
{
    return group_seeded(g, sc, probs, n_probs, costs_s(sc, probs, n_probs), hsps, n_hsps, lowest_level, src, out,
                        [=](SpdpContext* c, const SpdpScoring* s, const SpdpProblem* p, int cnt, const SpdpJuxt* const* h, const int32_t* nh,
                            const int32_t* ll, const SpdpHspSource* so, SpdpAlignment* o) { return spdp_align_s_seeded(c, s, sp, p, cnt, h, nh, ll, so, o); });
}
int spdp_group_align_h_seeded(SpdpGroup* g, const SpdpScoringH* sc, const SpdpSeedParams* sp, const SpdpProblemH* probs, int n_probs,
                              const SpdpJuxt* const* hsps, const int32_t* n_hsps, const int32_t* lowest_level,
                              const SpdpHspSource* src, SpdpAlignment* out)
{
    return group_seeded(g, sc, probs, n_probs, costs_h(sc, probs, n_probs), hsps, n_hsps, lowest_level, src, out,
                        [=](SpdpContext* c, const SpdpScoringH* s, const SpdpProblemH* p, int cnt, const SpdpJuxt* const* h, const int32_t* nh,
                            const int32_t* ll, const SpdpHspSource* so, SpdpAlignment* o) { return spdp_align_h_seeded(c, s, sp, p, cnt, h, nh, ll, so, o); });
}

// skl_rngS_ng / skl_rngH_ng for a batch of alignments, sharded by alignment length (records)
int spdp_group_skl_rng_s(SpdpGroup* g, const SpdpScoring* sc, const SpdpRescoreParams* rp, const SpdpProblem* probs, int n_probs,
                         const SpdpAlignment* aln, SpdpRescored* out)
{
    std::vector<int64_t> cost(std::max(n_probs, 0));
    for (int i = 0; i < n_probs; ++i) cost[i] = 1 + std::max(aln[i].n_skl, 0) + (probs[i].a_right - probs[i].a_left);
    return fan_out_idx(g, n_probs, cost, [&](SpdpContext* c, const std::vector<int>& idx) {
        const int cnt = (int) idx.size();
        std::vector<SpdpProblem> mine(cnt);
        std::vector<SpdpAlignment> al(cnt);
        std::vector<SpdpRescored> o(cnt);
        for (int k = 0; k < cnt; ++k) { mine[k] = probs[idx[k]]; al[k] = aln[idx[k]]; o[k] = out[idx[k]]; }
        const int rc = spdp_skl_rng_s(c, sc, rp, mine.data(), cnt, al.data(), o.data());
        for (int k = 0; k < cnt; ++k) out[idx[k]] = o[k];
        return rc;
    });
}
int spdp_group_skl_rng_h(SpdpGroup* g, const SpdpScoringH* sc, const SpdpRescoreParamsH* rp, const SpdpProblemH* probs, int n_probs,
                         const SpdpAlignment* aln, SpdpRescored* out)
{
    std::vector<int64_t> cost(std::max(n_probs, 0));
    for (int i = 0; i < n_probs; ++i) cost[i] = 1 + std::max(aln[i].n_skl, 0) + (probs[i].a_right - probs[i].a_left);
    return fan_out_idx(g, n_probs, cost, [&](SpdpContext* c, const std::vector<int>& idx) {
        const int cnt = (int) idx.size();
        std::vector<SpdpProblemH> mine(cnt);
        std::vector<SpdpAlignment> al(cnt);
        std::vector<SpdpRescored> o(cnt);
        for (int k = 0; k < cnt; ++k) { mine[k] = probs[idx[k]]; al[k] = aln[idx[k]]; o[k] = out[idx[k]]; }
        const int rc = spdp_skl_rng_h(c, sc, rp, mine.data(), cnt, al.data(), o.data());
        for (int k = 0; k < cnt; ++k) out[idx[k]] = o[k];
        return rc;
    });
}

// the block vote for a batch of queries: every member holds its own copy of the index (ix[r] created on the group's r-th
// context: spdp_group_context), the queries are sharded by length
SpdpContext* spdp_group_context(SpdpGroup* g, int member) { return (g && member >= 0 && member < (int) g->ctx.size()) ? g->ctx[member] : nullptr; }
int spdp_group_blk_vote(SpdpGroup* g, const SpdpBlkIndex* const* ix, const uint8_t* codes, const int64_t* offs,
                        const int32_t* left, const int32_t* right, const int32_t* stop_at, int32_t n, int32_t* out, int32_t out_cap)
{
    if (!ix || !codes || !offs || !left || !right || !out) { if (g) g->err = "spdp_group_blk_vote: null argument"; return -1; }
    std::vector<int64_t> cost(std::max(n, 0));
    for (int i = 0; i < n; ++i) cost[i] = 1 + right[i] - left[i];
    return fan_out_idx(g, n, cost, [&](SpdpContext* c, const std::vector<int>& idx) {
        int member = 0;
        while (g->ctx[member] != c) ++member;
        const int cnt = (int) idx.size();
        std::vector<int64_t> o2(cnt + 1, 0);
        for (int k = 0; k < cnt; ++k) o2[k + 1] = o2[k] + (offs[idx[k] + 1] - offs[idx[k]]);
        std::vector<uint8_t> cd((size_t) o2[cnt]);
        std::vector<int32_t> l(cnt), r(cnt), st(cnt), rec((size_t) cnt * out_cap);
        for (int k = 0; k < cnt; ++k) {
            memcpy(cd.data() + o2[k], codes + offs[idx[k]], (size_t) (o2[k + 1] - o2[k]));
            l[k] = left[idx[k]]; r[k] = right[idx[k]]; st[k] = stop_at ? stop_at[idx[k]] : 0;
        }
        const int rc = spdp_blk_vote(c, ix[member], cd.data(), o2.data(), l.data(), r.data(), stop_at ? st.data() : nullptr, cnt,
                                     rec.data(), out_cap, nullptr);
        if (rc == 0)
            for (int k = 0; k < cnt; ++k) memcpy(out + (size_t) idx[k] * out_cap, rec.data() + (size_t) k * out_cap, (size_t) out_cap * 4);
        return rc;
    });
}

int spdp_group_map_align_s(SpdpGroup* g, const SpdpBlkIndex* const* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                           const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpSignalModel* sigmodel,
                           const SpdpBlkFindParams* fprm, const SpdpRescoreParams* rp,
                           const uint8_t* codes, const int64_t* offs, int32_t n, int32_t ori,
                           SpdpMapGene* genes, SpdpMapExon** exons)
{
    if (!ix || !codes || !offs || !genes || !exons) { if (g) g->err = "spdp_group_map_align_s: null argument"; return -1; }
    *exons = nullptr;
    std::vector<int64_t> cost(std::max(n, 0));
    for (int i = 0; i < n; ++i) cost[i] = 1 + offs[i + 1] - offs[i];
    const int w = g ? (int) g->ctx.size() : 0;
    std::vector<std::vector<SpdpMapExon>> part(std::max(w, 1));        // a member's exons, its genes' exon_off relative to them
    std::vector<std::vector<int>> whose(std::max(w, 1));
    const int rc = fan_out_idx(g, n, cost, [&](SpdpContext* c, const std::vector<int>& idx) {
        int member = 0;
        while (g->ctx[member] != c) ++member;
        const int cnt = (int) idx.size();
        std::vector<int64_t> o2(cnt + 1, 0);
        for (int k = 0; k < cnt; ++k) o2[k + 1] = o2[k] + (offs[idx[k] + 1] - offs[idx[k]]);
        std::vector<uint8_t> cd((size_t) o2[cnt]);
        for (int k = 0; k < cnt; ++k) memcpy(cd.data() + o2[k], codes + offs[idx[k]], (size_t) (o2[k + 1] - o2[k]));
        std::vector<SpdpMapGene> gn(cnt);
        SpdpMapExon* ex = nullptr;
        const int r = spdp_map_align_s(c, ix[member], hix, genome, sc, sp, sigmodel, fprm, rp, cd.data(), o2.data(), cnt, ori, gn.data(), &ex, nullptr);
        if (r >= 0) {
            size_t ne = 0;
            for (int k = 0; k < cnt; ++k) { genes[idx[k]] = gn[k]; ne = std::max<size_t>(ne, (size_t) gn[k].exon_off + gn[k].n_exons); }
            if (ex) part[member].assign(ex, ex + ne);
            whose[member] = idx;
        }
        free(ex);
        return r;
    });
    if (rc < 0) return rc;
    size_t total = 0;
    for (int r = 0; r < w; ++r) total += part[r].size();
    *exons = (SpdpMapExon*) malloc(sizeof(SpdpMapExon) * std::max<size_t>(total, 1));
    if (!*exons) { g->err = "spdp_group_map_align_s: out of memory"; return -1; }
    size_t base = 0;
    for (int r = 0; r < w; ++r) {
        if (!part[r].empty()) memcpy(*exons + base, part[r].data(), sizeof(SpdpMapExon) * part[r].size());
        for (int i : whose[r]) genes[i].exon_off += (int64_t) base;
        base += part[r].size();
    }
    return rc;
}

// which member ran problem i in the last group call (cost-balanced: spdp_cells per problem, longest first)
int spdp_group_last_shards(const SpdpGroup* g, int32_t* member, int n)
{
    if (!g || !member) return -1;
    for (int i = 0; i < n && i < (int) g->shard_of.size(); ++i) member[i] = g->shard_of[i];
    return (int) g->shard_of.size();
}

}   // extern "C"
