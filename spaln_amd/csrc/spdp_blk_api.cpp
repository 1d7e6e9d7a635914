// spdp_blk_api.cpp -- host side of the block search's vote (include/spdp.h, "block search"): the index goes to the device
// once, a call uploads its queries, runs spdp_blk_vote_wave (a wave per query, waves persistent) and brings the records back.
#include "spdp_internal.h"
#include "spdp_blk_dev.h"
#include "spdp_hsp_dev.h"
#include "spdp_loci.h"
#include "spdp_hsp_host.h"
#include "spdp_region.h"
#include "spdp_hostcpus.h"
#include <atomic>
#include <chrono>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

struct SpdpBlkIndex {
    SpdpContext* ctx = nullptr;
    BlkDev dev;
    std::vector<void*> bufs;                    // device copies of the index arrays
    uint8_t* slabs = nullptr; uint32_t* next = nullptr;
    // the genome, resident for the HSP searches (uploaded by the first spdp_blk_find that names it)
    uint8_t* d_genome = nullptr; const uint8_t* genome_host = nullptr; int64_t genome_len = 0;
    uint8_t* d_tron = nullptr; int32_t* d_mtx = nullptr;
    BlkDev* d_dev = nullptr;                    // `dev` as the kernels read it
    size_t slab_bytes = 0;
    int n_waves = 0, hh_in_lds = 0;
    uint32_t lds_bytes = 0;
    ~SpdpBlkIndex()
    {
        for (void* p : bufs) if (p) (void) hipFree(p);
        if (slabs) (void) hipFree(slabs);
        if (next) (void) hipFree(next);
        if (d_genome) (void) hipFree(d_genome);
        if (d_tron) (void) hipFree(d_tron);
        if (d_mtx) (void) hipFree(d_mtx);
        if (d_dev) (void) hipFree(d_dev);
    }
};

namespace {

template <class T>
const T* to_device(SpdpBlkIndex* ix, const T* src, size_t n, hipError_t& e)
{
    void* d = nullptr;
    if (e != hipSuccess) return nullptr;
    e = hipMalloc(&d, std::max<size_t>(n * sizeof(T), 16));
    if (e != hipSuccess) return nullptr;
    ix->bufs.push_back(d);
    e = hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice);
    return (const T*) d;
}

// smallest prime-ish size the reference's Dhash(n, ..) picks is not restated here: the caller passes the geometry its own
// containers have (SpdpBlkIndexDesc::hh_size ...); only the second step has a fixed default (SecondHS, src/clib.h:76)
int or_default(int v, int d) { return v > 0 ? v : d; }

// waves of a launch: as many as the LDS of every CU holds (the run hash and the queues of a wave live there), as many slabs as
// fit the budget
int pick_waves(const SpdpContext* ctx, size_t slab_bytes, uint32_t lds_bytes)
{
    size_t budget = (size_t) 64 << 30;
    if (const char* e = getenv("SPDP_BLK_SLAB_GB")) budget = (size_t) std::max(1, atoi(e)) << 30;
    int waves_per_cu = (int) std::max<size_t>(1, std::min<size_t>(16, ((size_t) 160 << 10) / std::max<uint32_t>(lds_bytes, 1)));
    if (const char* e = getenv("SPDP_BLK_WAVES_PER_CU")) waves_per_cu = std::max(1, std::min(atoi(e), 32));
    size_t waves = (size_t) std::max(1, ctx->n_cu) * waves_per_cu;
    waves = std::min(waves, std::max<size_t>(1, budget / std::max<size_t>(slab_bytes, 1)));
    return (int) waves;
}

}  // namespace

extern "C" SpdpBlkIndex* spdp_blk_index_create(SpdpContext* ctx, const SpdpBlkIndexDesc* d)
{
    if (!ctx) return nullptr;
    if (!d || !d->convtab || !d->nblk || !d->wscr || !d->blkp || !d->blkb || !d->rscrtab || !d->chr || !d->bitpat) {
        ctx->err = "spdp_blk_index_create: index arrays missing"; return nullptr;
    }
    if (d->kk < 1 || d->kk > 3 || d->nshift < 1 || d->nshift > SPDP_BLK_MAX_SHIFT || d->tabsize < 1 || d->nseg < 2 ||
        d->ncand < 1 || d->nascr < 1 || d->hh_size < 2 || d->hb_size < 2 || d->ha_size < 2 || d->maxmmc < 1 || d->n_chr < 1) {
        ctx->err = "spdp_blk_index_create: parameter out of range (kk 1..3, Nshift <= 32, table geometries given)"; return nullptr;
    }
    // findblock serves nucleotide queries on a nucleotide index (DRNA, Nalpha = 4) and protein queries on the amino-acid words
    // of a translated genome (-KP; src/blksrc.cc:2184-2187): the two must agree
    if ((d->drna != 0) != (d->nalpha == 4) || d->nalpha < 2 || d->nalpha > 32) {
        ctx->err = "spdp_blk_index_create: drna = 1 goes with nalpha = 4 (nucleotide index), drna = 0 with an amino-acid alphabet";
        return nullptr;
    }
    (void) hipSetDevice(ctx->device);
    SpdpBlkIndex* ix = new SpdpBlkIndex;
    ix->ctx = ctx;
    BlkDev& v = ix->dev;
    memset(&v, 0, sizeof v);
    v.nalpha = d->nalpha; v.tabsize = d->tabsize; v.nshift = d->nshift; v.nbitpat = d->nbitpat; v.convts = d->convts;
    v.n_chr = d->n_chr; v.kk = d->kk; v.drna = d->drna; v.maxmmc = d->maxmmc; v.nseg = d->nseg; v.minsigpr = d->minsigpr;
    v.ncand = d->ncand; v.nascr = d->nascr; v.maxblock = d->maxblock; v.extblock = d->extblock; v.extblockl = d->extblockl; v.shortquery = d->shortquery;
    v.maxlist = std::max(1, d->maxblk);
    v.hh_size1 = d->hh_size; v.hh_size2 = or_default(d->hh_step, 8);
    v.hb_size1 = d->hb_size; v.hb_size2 = or_default(d->hb_step, 8);
    v.ha_size1 = d->ha_size; v.ha_size2 = or_default(d->ha_step, 8);
    blk_fill_hash_levels(v);
    v.gdb = d->gdb; v.rbscoef = d->rbscoef; v.rbscons = d->rbscons;
    v.bclw = d->bclw; v.bcup = d->bcup; v.bcce = d->bcce;
    v.app_c = d->kk > 1 ? pow((double) d->nbitpat, d->cfact) : 1.;
    int at = 0;
    for (int k = 0; k < d->kk; ++k) {
        if (at + 3 > d->n_bitpat || d->bitpat[at] < 1 || at + 3 + 2 * d->bitpat[at] > d->n_bitpat) {
            ctx->err = "spdp_blk_index_create: bit-pattern record too short"; delete ix; return nullptr;
        }
        v.pat_off[k] = at;
        at += 3 + 2 * d->bitpat[at];
    }
    // everything the kernel later uses as an INDEX is checked here, once, on the host: a mismatched or corrupt index would
    // otherwise write into other lanes' score slabs without a sign
    // posting lists must stay inside blkb
    for (int64_t w = 0; w < d->tabsize; ++w)
        if (d->blkp[w] && ((int64_t) d->blkp[w] - 1 + d->nblk[w] > d->n_words || d->blkp[w] < 0)) {
            ctx->err = "spdp_blk_index_create: a posting list runs past blkb"; delete ix; return nullptr;
        }
    // block numbers index the per-lane slabs (acr[blk], rscr[blk])
    for (int64_t i = 0; i < d->n_words; ++i)
        if ((int64_t) d->blkb[i] >= (int64_t) d->nseg) {
            ctx->err = "spdp_blk_index_create: a block number in blkb is not below nseg"; delete ix; return nullptr;
        }
    // words index blkp / wscr / nblk: the table must have nalpha ^ weight entries for every pattern; exam offsets stay inside the pattern
    for (int k = 0; k < d->kk; ++k) {
        const int32_t* bp = d->bitpat + v.pat_off[k];
        const int weight = bp[0], width = bp[1];
        int64_t words = 1;
        for (int i = 0; i < weight && words <= (int64_t) d->tabsize; ++i) words *= d->nalpha;
        if (words != (int64_t) d->tabsize || width < weight) {
            ctx->err = "spdp_blk_index_create: tabsize is not nalpha ^ weight of a bit pattern"; delete ix; return nullptr;
        }
        for (int i = 0; i < 2 * weight; ++i)
            if (bp[3 + i] < 0 || bp[3 + i] >= width) {
                ctx->err = "spdp_blk_index_create: an exam offset of a bit pattern lies outside its width"; delete ix; return nullptr;
            }
    }
    // the chromosome table is searched by block number: first blocks ascend and stay below nseg
    for (int c = 0; c <= d->n_chr; ++c) {
        const int64_t fb = d->chr[2 * c + 1];
        if (fb < 0 || fb >= d->nseg + 1 || (c && fb < d->chr[2 * c - 1])) {
            ctx->err = "spdp_blk_index_create: the chromosome table's first blocks do not ascend inside nseg"; delete ix; return nullptr;
        }
    }
    hipError_t e = hipSuccess;
    v.convtab = to_device(ix, d->convtab, d->convts, e);
    v.nblk = to_device(ix, d->nblk, d->tabsize, e);
    v.wscr = to_device(ix, d->wscr, d->tabsize, e);
    v.blkp = to_device(ix, d->blkp, d->tabsize, e);
    v.blkb = to_device(ix, d->blkb, (size_t) d->n_words, e);
    v.rscrtab = to_device(ix, d->rscrtab, 128, e);
    v.chr = to_device(ix, d->chr, 2 * ((size_t) d->n_chr + 1), e);
    v.bitpat = to_device(ix, d->bitpat, d->n_bitpat, e);
    if (e != hipSuccess) { ctx->err = std::string("spdp_blk_index_create: ") + hipGetErrorString(e); delete ix; return nullptr; }
    // the longest posting list the kernel may meet (its staging area holds two): the header's MaxBlk, checked against the table
    for (int64_t w = 0; w < d->tabsize; ++w) if (d->blkp[w]) v.maxlist = std::max<int32_t>(v.maxlist, d->nblk[w]);
    if (v.hh_size2 < 1 || v.hb_size2 < 1 || v.ha_size2 < 1 || v.hh_size2 > v.hh_size1 || v.hb_size2 > v.hb_size1 || v.ha_size2 > v.ha_size1) {
        ctx->err = "spdp_blk_index_create: a table's second step exceeds its size"; delete ix; return nullptr;
    }
    // the run hash of a wave lives in LDS when its first level fits beside the queues (8 KB: some ten waves per CU)
    ix->hh_in_lds = (size_t) v.hh_sizes[0] * 8 <= (size_t) 8 << 10;
    if (const char* e = getenv("SPDP_BLK_HH_LDS")) ix->hh_in_lds = atoi(e) != 0 && (size_t) v.hh_sizes[0] * 8 <= (size_t) 48 << 10;
    ix->lds_bytes = spdp_blk_vote_lds_bytes(&v, ix->hh_in_lds);
    if (ix->lds_bytes > (64u << 10)) { ctx->err = "spdp_blk_index_create: the queues of a query do not fit the LDS of a workgroup (ncand / nascr too large)"; delete ix; return nullptr; }
    ix->slab_bytes = spdp_blk_vote_slab_bytes(&v, ix->hh_in_lds);
    if (getenv("SPDP_BLK_VERBOSE")) fprintf(stderr, "[blk] nseg %d, longest list %d, run hash %d slots (%s), LDS %u B per wave, slab %.2f MB per wave\n",
                                            v.nseg, v.maxlist, v.hh_sizes[0], ix->hh_in_lds ? "LDS" : "HBM", ix->lds_bytes, ix->slab_bytes / 1048576.);
    return ix;
}

extern "C" void spdp_blk_index_destroy(SpdpBlkIndex* ix)
{
    if (!ix) return;
    (void) hipSetDevice(ix->ctx->device);
    delete ix;
}

static int ensure_waves(SpdpContext* ctx, SpdpBlkIndex* ix, int n)
{
    int want = pick_waves(ctx, ix->slab_bytes, ix->lds_bytes);
    want = std::min(want, std::max(1, n));
    if (!ix->next) { HIPCHK(hipMalloc((void**) &ix->next, 16)); }
    if (want <= ix->n_waves) return 0;
    if (ix->slabs) { (void) hipFree(ix->slabs); ix->slabs = nullptr; }
    ix->n_waves = 0;
    HIPCHK(hipMalloc((void**) &ix->slabs, (size_t) want * ix->slab_bytes));
    HIPCHK(hipMemsetAsync(ix->slabs, 0, (size_t) want * ix->slab_bytes, ctx->stream));     // (tags of no query: every score slot reads as zero)
    ix->n_waves = want;
    return 0;
}

extern "C" int spdp_blk_vote_resident(SpdpContext* ctx, const SpdpBlkIndex* cix, const uint8_t* d_codes, const int64_t* d_offs,
                                      const int32_t* d_left, const int32_t* d_right, const int32_t* d_stop_at, int32_t n,
                                      int32_t* d_out, int32_t out_cap, float* kernel_ms)
{
    if (!ctx) return -1;
    SpdpBlkIndex* ix = const_cast<SpdpBlkIndex*>(cix);
    if (!ix || ix->ctx != ctx) { ctx->err = "spdp_blk_vote: the index belongs to another context"; return -1; }
    if (n <= 0) return 0;
    if (out_cap < 3) { ctx->err = "spdp_blk_vote: out_cap < 3"; return -1; }
    (void) hipSetDevice(ctx->device);
    if (ensure_waves(ctx, ix, n)) return -1;
    BlkVoteArgs A;
    if (!ix->d_dev) {
        HIPCHK(hipMalloc((void**) &ix->d_dev, sizeof(BlkDev)));
        HIPCHK(hipMemcpy(ix->d_dev, &ix->dev, sizeof(BlkDev), hipMemcpyHostToDevice));
    }
    A.ix = ix->d_dev;
    A.codes = d_codes; A.offs = d_offs; A.left = d_left; A.right = d_right; A.stop_at = d_stop_at;
    A.out = d_out; A.out_cap = out_cap; A.n = n;
    A.slabs = ix->slabs; A.slab_bytes = ix->slab_bytes; A.next = ix->next;
    A.n_waves = std::min(ix->n_waves, n);
    A.hh_in_lds = ix->hh_in_lds; A.lds_bytes = ix->lds_bytes; A.res_cap = SPDP_BLK_RES_CAP;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(spdp_blk_vote_launch(&A, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) HIPCHK(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    if (getenv("SPDP_BLK_VERBOSE")) {                  // how often the run hash met a full table (one-lane path) and grew, over the slabs' lifetime
        uint64_t slow = 0, grown = 0;
        for (int w = 0; w < A.n_waves; ++w) {
            uint32_t h[4];
            HIPCHK(hipMemcpy(h, ix->slabs + (size_t) w * ix->slab_bytes, sizeof h, hipMemcpyDeviceToHost));
            slow += h[1]; grown += h[2];
        }
        fprintf(stderr, "[blk] %d queries on %d waves: so far %llu entries took the one-lane path, %llu table growths\n", n, A.n_waves,
                (unsigned long long) slow, (unsigned long long) grown);
    }
    return 0;
}

extern "C" int spdp_blk_vote(SpdpContext* ctx, const SpdpBlkIndex* ix, const uint8_t* codes, const int64_t* offs,
                             const int32_t* left, const int32_t* right, const int32_t* stop_at, int32_t n,
                             int32_t* out, int32_t out_cap, float* kernel_ms)
{
    if (!ctx) return -1;
    if (n <= 0) return 0;
    if (!codes || !offs || !left || !right || !out) { ctx->err = "spdp_blk_vote: null argument"; return -1; }
    for (int i = 0; i < n; ++i) {
        const int64_t len = offs[i + 1] - offs[i];
        if (len < 0 || len > INT32_MAX || left[i] < 0 || right[i] > len || right[i] < left[i]) {
            ctx->err = "spdp_blk_vote: bad query range"; return -1;
        }
    }
    (void) hipSetDevice(ctx->device);
    struct Dev { void* p = nullptr; ~Dev() { if (p) (void) hipFree(p); } } d_codes, d_offs, d_l, d_r, d_s, d_out;
    const size_t nb = (size_t) (offs[n] - offs[0]);
    HIPCHK(hipMalloc(&d_codes.p, std::max<size_t>(nb, 16)));
    HIPCHK(hipMalloc(&d_offs.p, ((size_t) n + 1) * 8));
    HIPCHK(hipMalloc(&d_l.p, (size_t) n * 4)); HIPCHK(hipMalloc(&d_r.p, (size_t) n * 4));
    if (stop_at) HIPCHK(hipMalloc(&d_s.p, (size_t) n * 4));
    HIPCHK(hipMalloc(&d_out.p, (size_t) n * out_cap * 4));
    std::vector<int64_t> rel(n + 1);
    for (int i = 0; i <= n; ++i) rel[i] = offs[i] - offs[0];
    HIPCHK(hipMemcpyAsync(d_codes.p, codes + offs[0], nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_offs.p, rel.data(), ((size_t) n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_l.p, left, (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_r.p, right, (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (stop_at) HIPCHK(hipMemcpyAsync(d_s.p, stop_at, (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));          // (rel is a local: the copies must have read it)
    if (spdp_blk_vote_resident(ctx, ix, (const uint8_t*) d_codes.p, (const int64_t*) d_offs.p, (const int32_t*) d_l.p,
                               (const int32_t*) d_r.p, (const int32_t*) d_s.p, n, (int32_t*) d_out.p, out_cap, kernel_ms)) return -1;
    HIPCHK(hipMemcpy(out, d_out.p, (size_t) n * out_cap * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ---- from the vote to candidate loci -------------------------------------------------------------------------------------------
// The vote runs on the device call after call (stop_at); between two votes every query that reached a TestOutput call is a machine
// (spdp_loci.h) that names the regions it needs searched.  The searches of all machines go to the device as batches (spdp_hsp.hip:
// the first regions of all pairs in one launch, a protein query's second looks in further ones), their HSPs are chained on the
// host threads (spdp_hsp_chain.h), the machines advance.
namespace {

template <class F> void on_threads(int n, F f)
{
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    auto work = [&] { try { for (int k; (k = next++) < n; ) f(k); } catch (...) { failed = true; } };
    const int nt = std::max(1, std::min(spdp_host_cpus(), n));
    std::vector<std::thread> th;
    try { for (int t = 1; t < nt; ++t) th.emplace_back(work); } catch (...) {}      // (fewer threads: the caller's does the rest)
    work();
    for (std::thread& t : th) t.join();
    if (failed) throw std::bad_alloc();
}

struct SearchTask { int machine, pair; spdp_loci::Region r; std::vector<spdp_hsp::Unit> units; };

struct HspBatch {                                       // what a call's searches share
    SpdpContext* ctx; SpdpBlkIndex* ix; const SpdpGenome* genome; const SpdpWilipModel* model; const SpdpScoring* sc;
    const spdp_loci::Params* P;
    const uint8_t* codes; const int64_t* offs; const int32_t* left; const int32_t* right;       // the call's queries (host)
    uint8_t* d_codes = nullptr;
    ~HspBatch() { if (d_codes) (void) hipFree(d_codes); }

    int prepare(int n)
    {
        // the genome goes to the device once per index (the caller's array is the key: the same pointer and length again = resident)
        const int64_t glen = genome->chr_off[genome->n_chr];
        if (ix->genome_host != genome->codes || ix->genome_len != glen) {
            if (ix->d_genome) { (void) hipFree(ix->d_genome); ix->d_genome = nullptr; }
            HIPCHK(hipMalloc((void**) &ix->d_genome, (size_t) std::max<int64_t>(glen, 16)));
            HIPCHK(hipMemcpy(ix->d_genome, genome->codes, (size_t) glen, hipMemcpyHostToDevice));
            ix->genome_host = genome->codes; ix->genome_len = glen;
        }
        if (!ix->d_tron) {
            uint8_t mid[32], tron_of[64];
            spdp_genetic_code_tables(mid, tron_of);
            HIPCHK(hipMalloc((void**) &ix->d_tron, 64));
            HIPCHK(hipMemcpy(ix->d_tron, tron_of, 64, hipMemcpyHostToDevice));
        }
        if (!ix->d_mtx) HIPCHK(hipMalloc((void**) &ix->d_mtx, sizeof model->mtx));
        HIPCHK(hipMemcpy(ix->d_mtx, model->mtx, sizeof model->mtx, hipMemcpyHostToDevice));
        const size_t nb = (size_t) (offs[n] - offs[0]);
        HIPCHK(hipMalloc((void**) &d_codes, std::max<size_t>(nb, 16)));
        HIPCHK(hipMemcpy(d_codes, codes + offs[0], nb, hipMemcpyHostToDevice));
        return 0;
    }
    spdp_hsp::ChainCost cost(int a_len) const
    {
        int vthr = model->level[0].vthr;
        if (a_len < model->shortquery) vthr = vthr * a_len / model->shortquery;       // (as the search scaled it)
        return {model, sc->intpen, sc->intpen_len, sc->gop, sc->gep, sc->lgop, sc->lgep, sc->codonk1, P->bbt == 3 ? 3 : 1, vthr};
    }
    // the host's form, for a task the device hands back
    void on_host(SearchTask& t, int q) const
    {
        std::vector<uint8_t> reg;
        spdp_region::materialize(genome->codes, genome->chr_off, t.r.chr, t.r.base, t.r.len, t.r.rvs != 0, P->bbt == 3, reg);
        const int a_len = (int) (offs[q + 1] - offs[q]);
        const spdp_hsp::Seqs s = {codes + offs[q], a_len, left[q], right[q], P->a_exgl, P->a_exgr, reg.data(), t.r.len, 0, t.r.len,
                                  P->bbt == 3 ? 3 : 1, nullptr, nullptr, nullptr};
        const spdp_hsp::GapCosts gc = {sc->intpen, sc->intpen_len, sc->gop, sc->gep, sc->lgop, sc->lgep, sc->codonk1};
        spdp_hsp::search(model, s, gc, -1, t.units);
    }
    // all tasks: device search, then the chains on the host threads.  query_of[machine] = the query
    int run(std::vector<SearchTask>& tasks, const std::vector<int>& query_of)
    {
        const int n = (int) tasks.size();
        if (!n) return 0;
        std::vector<HspTask> ht(n);
        int longest = 1;
        for (int k = 0; k < n; ++k) {
            const int q = query_of[tasks[k].machine];
            HspTask& T = ht[k];
            T.a_off = offs[q] - offs[0]; T.a_len = (int32_t) (offs[q + 1] - offs[q]); T.a_left = left[q]; T.a_right = right[q];
            T.g_off = genome->chr_off[tasks[k].r.chr] + tasks[k].r.base; T.b_len = tasks[k].r.len; T.rvs = tasks[k].r.rvs;
            T.a_exgl = P->a_exgl; T.a_exgr = P->a_exgr; T.pad = 0;
            longest = std::max(longest, T.a_right - T.a_left);
        }
        HspArgs A;
        memset(&A, 0, sizeof A);
        A.codes = d_codes; A.genome = ix->d_genome; A.n_tasks = n;
        A.level = model->level[0];
        A.mtx = ix->d_mtx; A.mtx_rows = model->mtx_rows; A.mtx_cols = model->mtx_cols; A.tron_of = ix->d_tron;
        A.bbt = P->bbt == 3 ? 3 : 1; A.shortquery = model->shortquery; A.end_bonus = model->end_bonus; A.crs = model->crs;
        A.ser = model->ser; A.ser2 = model->ser2;
        // the wave's LDS: query words (twice their number, a power of two), shared words, segments -- sized for the batch's longest query,
        // capped where one wave per CU is left; what does not fit comes back flagged
        int slots = 256;
        while (slots < 2 * longest && slots < 8192) slots <<= 1;
        A.hash_slots = slots; A.hit_cap = slots <= 1024 ? 4096 : 8192; A.seg_cap = 256; A.out_cap = 96;
        if (const char* e = getenv("SPDP_HSP_HITS")) A.hit_cap = std::max(256, atoi(e));
        const uint32_t lds = spdp_hsp_lds_bytes(&A);
        struct Dev { void* p = nullptr; ~Dev() { if (p) (void) hipFree(p); } } d_tasks, d_out, d_counts;
        HIPCHK(hipMalloc(&d_tasks.p, sizeof(HspTask) * (size_t) n));
        HIPCHK(hipMalloc(&d_out.p, sizeof(int32_t) * 8 * (size_t) A.out_cap * n));
        HIPCHK(hipMalloc(&d_counts.p, sizeof(int32_t) * 2 * (size_t) n));
        HIPCHK(hipMemcpyAsync(d_tasks.p, ht.data(), sizeof(HspTask) * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
        A.tasks = (const HspTask*) d_tasks.p; A.out = (int32_t*) d_out.p; A.counts = (int32_t*) d_counts.p;
        const int per_cu = (int) std::max<uint32_t>(1, std::min<uint32_t>(8, (160u << 10) / std::max<uint32_t>(lds, 1)));
        const int waves = std::min(n, std::max(1, ctx->n_cu) * per_cu);
        HIPCHK(spdp_hsp_launch(&A, waves, ctx->stream));
        std::vector<int32_t> counts(2 * (size_t) n), out(8 * (size_t) A.out_cap * n);
        HIPCHK(hipMemcpyAsync(counts.data(), d_counts.p, sizeof(int32_t) * counts.size(), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(out.data(), d_out.p, sizeof(int32_t) * out.size(), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        on_threads(n, [&](int k) {
            SearchTask& t = tasks[k];
            const int q = query_of[t.machine];
            if (counts[2 * k + 1]) { on_host(t, q); return; }
            struct Rec { int v[8]; };
            const Rec* r = (const Rec*) (out.data() + 8 * (size_t) A.out_cap * k);
            std::vector<Rec> recs(r, r + counts[2 * k]);
            std::sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) { return x.v[5] != y.v[5] ? x.v[5] < y.v[5] : x.v[6] < y.v[6]; });    // scan order: diagonal, position
            std::vector<spdp_hsp::Hsp> hsps;
            for (const Rec& x : recs) hsps.push_back({x.v[0], x.v[1], x.v[2], x.v[3], x.v[4]});
            spdp_hsp::chain(hsps, cost(ht[k].a_len), left[q], right[q], 0, t.r.len, t.units);
        });
        return 0;
    }
};

}   // namespace

static int blk_find(SpdpContext* ctx, const SpdpBlkIndex* cix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                    const SpdpWilipModel* model, const SpdpScoring* sc, const SpdpBlkFindParams* prm,
                    const uint8_t* codes, const int64_t* offs, const int32_t* left, const int32_t* right, int32_t n,
                    SpdpLocus** loci, int32_t* n_loci, SpdpJuxt** hsps, int32_t* status)
{
    SpdpBlkIndex* ix = const_cast<SpdpBlkIndex*>(cix);
    spdp_loci::Params P;
    P.vthr = prm->vthr; P.drop_rate = prm->drop_rate; P.max_out = prm->max_out; P.max_out2 = prm->max_out2; P.min_agap = prm->min_agap;
    // protein queries against the translated index (-KP): SrchBlk::bbt = 3 (src/blksrc.cc:2218), the DvsP = 1 branch of FindHsp
    // with its NoRetry = 2 (:34) searches on a grown region
    P.bbt = model->dvsp == 1 ? 3 : 1; P.dvsp = model->dvsp; P.no_retry = 2;
    P.blklen = hix->blklen; P.ext_block = hix->extblock; P.ext_block_l = hix->extblockl; P.phase1t = prm->phase1t;
    P.a_exgl = prm->a_exgl; P.a_exgr = prm->a_exgr;
    if (P.max_out < 1 || P.max_out2 < P.max_out || P.blklen < 1) { ctx->err = "spdp_blk_find: max_out / max_out2 / blklen out of range"; return -1; }
    const spdp_loci::Chromosomes G = {genome->chr_off, genome->n_chr, hix->chr};
    const spdp_loci::RandomScore rnd = {hix->rscrtab, hix->rbscoef, hix->rbscons, hix->gdb};
    (void) hipSetDevice(ctx->device);
    HspBatch B = {ctx, ix, genome, model, sc, &P, codes, offs, left, right};
    if (B.prepare(n)) return -1;
    int out_cap = 4096;                                 // (a record that does not fit makes the round run again with room for it)
    const bool verbose = getenv("SPDP_FIND_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    double t_vote = 0, t_machines = 0, t_search = 0, t_advance = 0; int n_batches = 0; size_t n_tasks = 0;
    std::vector<int> active(n), stop(n, 0), crit(n, 0), calls(n, 0);
    for (int i = 0; i < n; ++i) active[i] = i;
    std::vector<std::vector<spdp_loci::Locus>> found(n);
    std::vector<int32_t> rec;
    while (!active.empty()) {
        const int m = (int) active.size();
        std::vector<int64_t> o(m + 1, 0);
        std::vector<int32_t> l(m), r(m), st(m);
        for (int k = 0; k < m; ++k) { const int q = active[k]; o[k + 1] = o[k] + (offs[q + 1] - offs[q]); l[k] = left[q]; r[k] = right[q]; st[k] = stop[q]; }
        std::vector<uint8_t> cd((size_t) o[m]);
        for (int k = 0; k < m; ++k) memcpy(cd.data() + o[k], codes + offs[active[k]], (size_t) (o[k + 1] - o[k]));
        auto t0 = now();
        for (;;) {
            rec.assign((size_t) m * out_cap, 0);
            if (spdp_blk_vote(ctx, ix, cd.data(), o.data(), l.data(), r.data(), st.data(), m, rec.data(), out_cap, nullptr)) return -1;
            bool cut = false;
            for (int k = 0; k < m && !cut; ++k) cut = (rec[(size_t) k * out_cap + 2] & SPDP_BLK_CUT) != 0;
            if (!cut || out_cap >= (1 << 20)) break;
            out_cap *= 8;
        }
        auto t1 = now(); t_vote += secs(t0, t1);
        // ---- a machine per query that reached its call
        std::vector<spdp_loci::Call> mach(m);
        std::vector<int> verdict(m, 0);                 // > 0 loci, 0 go on, -1 ended, -2 record cut / table full
        std::vector<char> live(m, 0);
        on_threads(m, [&](int k) {
            const int32_t* rc = rec.data() + (size_t) k * out_cap;
            if (!(rc[2] & SPDP_BLK_REACHED)) { verdict[k] = -1; return; }            // findblock ended before this call
            if (rc[2] & (SPDP_BLK_CUT | SPDP_BLK_TABLE)) { verdict[k] = -2; return; }
            spdp_loci::Call& c = mach[k];
            const int q = active[k];
            c.P = &P; c.G = &G; c.rnd = rnd;
            c.q = {(int) (offs[q + 1] - offs[q]), left[q], right[q]};
            int j = 3;
            memcpy(c.mmct, rc + j + 4, sizeof c.mmct);
            j += 20;
            for (int d = 0; d < 4; ++d) j += 1 + 2 * rc[j];
            const int np = rc[j++];
            for (int i = 0; i < np; ++i, j += 9)
                c.pairs.push_back({rc[j], rc[j + 1], 0, (uint32_t) rc[j + 2], (uint32_t) rc[j + 3], (uint32_t) rc[j + 4], (uint32_t) rc[j + 5],
                                   (uint32_t) rc[j + 6], (uint32_t) rc[j + 7], rc[j + 8]});
            const int nr = rc[j++];
            for (int i = 0; i < nr; ++i) c.runs.emplace_back((uint32_t) rc[j + 2 * i], rc[j + 2 * i + 1]);
            c.forced = (rc[2] & SPDP_BLK_FORCED) != 0;
            c.critjscr = crit[q];
            c.begin();
            live[k] = 1;
        });
        // ---- searches in batches: first every pair's first region, then what the machines ask for
        std::vector<SearchTask> tasks;
        for (int k = 0; k < m; ++k) {
            if (!live[k]) continue;
            std::vector<std::pair<int, spdp_loci::Region>> fr;
            mach[k].first_regions(fr);
            for (auto& x : fr) tasks.push_back({k, x.first, x.second, {}});
        }
        std::vector<std::vector<SearchTask>> answers(m);
        auto t2 = now(); t_machines += secs(t1, t2);
        while (!tasks.empty()) {
            auto t3 = now();
            ++n_batches; n_tasks += tasks.size();
            if (B.run(tasks, active)) return -1;
            auto t4 = now(); t_search += secs(t3, t4);
            for (SearchTask& t : tasks) answers[t.machine].push_back(std::move(t));
            tasks.clear();
            std::vector<SearchTask> more(m);
            std::vector<char> asks(m, 0);
            on_threads(m, [&](int k) {
                if (!live[k] || mach[k].done) return;
                spdp_loci::Call& c = mach[k];
                int pi; spdp_loci::Region rg;
                while (c.needs(pi, rg)) {
                    SearchTask* have = nullptr;
                    for (SearchTask& a : answers[k])
                        if (a.pair == pi && a.r.chr == rg.chr && a.r.rvs == rg.rvs && a.r.base == rg.base && a.r.len == rg.len) { have = &a; break; }
                    if (!have) { more[k] = {k, pi, rg, {}}; asks[k] = 1; return; }
                    std::vector<spdp_hsp::Unit> u = have->units;        // (a region may be asked for again: the answer stays)
                    c.take(u);
                }
            });
            for (int k = 0; k < m; ++k) if (asks[k]) tasks.push_back(std::move(more[k]));
            t_advance += secs(t4, now());
        }
        std::vector<int> again;
        for (int k = 0; k < m; ++k) {
            const int q = active[k];
            if (live[k]) {
                verdict[k] = mach[k].result;
                crit[q] = mach[k].critjscr;
                calls[q] = stop[q] + 1;
                if (verdict[k] > 0) found[q].assign(mach[k].loci.begin(), mach[k].loci.begin() + verdict[k]);
            }
            if (verdict[k] == 0) { ++stop[q]; again.push_back(q); }
            else if (verdict[k] == -2) { ctx->err = "spdp_blk_find: a vote record was cut or a hash table of the reference's size ran full"; return -1; }
            else if (verdict[k] < 0 && status) status[q] = -(calls[q] ? calls[q] : stop[q] + 1);
        }
        active.swap(again);
    }
    if (verbose) fprintf(stderr, "[find] %d queries: votes %.3f s, machines set up %.3f s, %d search batches of %zu tasks %.3f s (device + chains), machines advanced %.3f s\n",
                         n, t_vote, t_machines, n_batches, n_tasks, t_search, t_advance);
    size_t nl = 0, nh = 0;
    for (int q = 0; q < n; ++q) for (const spdp_loci::Locus& g : found[q]) { ++nl; nh += g.hsp.size(); }
    *loci = (SpdpLocus*) malloc(sizeof(SpdpLocus) * std::max<size_t>(nl, 1));
    *hsps = (SpdpJuxt*) malloc(sizeof(SpdpJuxt) * std::max<size_t>(nh, 1));
    if (!*loci || !*hsps) { free(*loci); free(*hsps); *loci = nullptr; *hsps = nullptr; ctx->err = "spdp_blk_find: out of memory"; return -1; }
    size_t a = 0, b = 0;
    for (int q = 0; q < n; ++q) {
        if (status && !found[q].empty()) status[q] = calls[q];
        for (const spdp_loci::Locus& g : found[q]) {
            SpdpLocus& L = (*loci)[a++];
            L.query = q; L.chr = g.at.chr; L.rvs = g.at.rvs; L.base = g.at.base; L.len = g.at.len; L.left = g.left; L.right = g.right;
            L.jscr = g.jscr; L.n_hsp = (int32_t) g.hsp.size() - 1; L.hsp_off = (int64_t) b;
            for (const spdp_hsp::Hsp& t : g.hsp) { (*hsps)[b++] = {t.jx, t.jy, t.jlen, t.nid, t.jscr}; }
        }
    }
    *n_loci = (int32_t) nl;
    return 0;
}

extern "C" int spdp_blk_find(SpdpContext* ctx, const SpdpBlkIndex* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                             const SpdpWilipModel* model, const SpdpScoring* sc, const SpdpBlkFindParams* prm,
                             const uint8_t* codes, const int64_t* offs, const int32_t* left, const int32_t* right, int32_t n,
                             SpdpLocus** loci, int32_t* n_loci, SpdpJuxt** hsps, int32_t* status)
{
    if (!ctx) return -1;
    if (!ix || !hix || !genome || !model || !sc || !prm || !loci || !n_loci || !hsps) { ctx->err = "spdp_blk_find: null argument"; return -1; }
    *loci = nullptr; *hsps = nullptr; *n_loci = 0;
    if (n <= 0) return 0;
    if (!sc->intpen || sc->intpen_len <= 0 || !genome->codes || !genome->chr_off || genome->n_chr != hix->n_chr || !hix->chr || !hix->rscrtab) {
        ctx->err = "spdp_blk_find: needs SpdpScoring.intpen, the genome of the index's chromosomes and the host index's tables"; return -1;
    }
    if ((model->dvsp == 0) != (hix->drna != 0)) { ctx->err = "spdp_blk_find: the block table is incompatible with the query type (src/blksrc.cc:2186)"; return -1; }
    if (ix->ctx != ctx) { ctx->err = "spdp_blk_find: the index belongs to another context"; return -1; }
    try { return blk_find(ctx, ix, hix, genome, model, sc, prm, codes, offs, left, right, n, loci, n_loci, hsps, status); }
    catch (...) {                                       // (nothing of C++ crosses the C boundary; the worker threads have been joined)
        if (*loci) { free(*loci); *loci = nullptr; }
        if (*hsps) { free(*hsps); *hsps = nullptr; }
        *n_loci = 0;
        ctx->err = "spdp_blk_find: out of host memory (or no thread could be started)";
        return -1;
    }
}
