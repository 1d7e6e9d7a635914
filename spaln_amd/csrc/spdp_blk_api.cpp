// spdp_blk_api.cpp -- host side of the block search's vote (include/spdp.h, "block search"): the index goes to the device
// once, a call uploads its queries, runs spdp_blk_vote_wave (a wave per query, waves persistent) and brings the records back.
#include "spdp_internal.h"
#include "spdp_blk_dev.h"
#include "spdp_blk_find.h"
#include "spdp_hostcpus.h"
#include <atomic>
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
    size_t slab_bytes = 0;
    int n_waves = 0, hh_in_lds = 0;
    uint32_t lds_bytes = 0;
    ~SpdpBlkIndex()
    {
        for (void* p : bufs) if (p) (void) hipFree(p);
        if (slabs) (void) hipFree(slabs);
        if (next) (void) hipFree(next);
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
    // the run hash of a wave lives in LDS when its first level fits beside the queues (16 KB: ten waves per CU)
    ix->hh_in_lds = (size_t) v.hh_sizes[0] * 8 <= (size_t) 16 << 10;
    if (const char* e = getenv("SPDP_BLK_HH_LDS")) ix->hh_in_lds = atoi(e) != 0 && (size_t) v.hh_sizes[0] * 8 <= (size_t) 48 << 10;
    ix->lds_bytes = spdp_blk_vote_lds_bytes(&v, ix->hh_in_lds);
    if (ix->lds_bytes > (64u << 10)) { ctx->err = "spdp_blk_index_create: the queues of a query do not fit the LDS of a workgroup (ncand / nascr too large)"; delete ix; return nullptr; }
    ix->slab_bytes = spdp_blk_vote_slab_bytes(&v, ix->hh_in_lds);
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
    A.ix = ix->dev;
    A.codes = d_codes; A.offs = d_offs; A.left = d_left; A.right = d_right; A.stop_at = d_stop_at;
    A.out = d_out; A.out_cap = out_cap; A.n = n;
    A.slabs = ix->slabs; A.slab_bytes = ix->slab_bytes; A.next = ix->next;
    A.n_waves = std::min(ix->n_waves, n);
    A.hh_in_lds = ix->hh_in_lds; A.lds_bytes = ix->lds_bytes;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(spdp_blk_vote_launch(&A, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) HIPCHK(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
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

// ---- from the vote to candidate loci (spdp_blk_find.h) ---------------------------------------------------------------------
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
    // the host's view of the index: what TestOutput / FindHsp read (random-score table, chromosome table, block geometry)
    BlkDev hv;
    memset(&hv, 0, sizeof hv);
    hv.gdb = hix->gdb; hv.rbscoef = hix->rbscoef; hv.rbscons = hix->rbscons; hv.rscrtab = hix->rscrtab; hv.nseg = hix->nseg;
    blk_find::Params P;
    P.vthr = prm->vthr; P.drop_rate = prm->drop_rate; P.max_out = prm->max_out; P.max_out2 = prm->max_out2; P.min_agap = prm->min_agap;
    // protein queries against the translated index (-KP): SrchBlk::bbt = 3 (src/blksrc.cc:2218), the DvsP = 1 branch of FindHsp
    // with its NoRetry = 2 (:34) searches on a grown region
    P.bbt = model->dvsp == 1 ? 3 : 1; P.dvsp = model->dvsp; P.no_retry = 2;
    if ((model->dvsp == 0) != (hix->drna != 0)) { ctx->err = "spdp_blk_find: the block table is incompatible with the query type (src/blksrc.cc:2186)"; return -1; }
    P.blklen = hix->blklen; P.ext_block = hix->extblock; P.ext_block_l = hix->extblockl; P.phase1t = prm->phase1t;
    P.a_exgl = prm->a_exgl; P.a_exgr = prm->a_exgr;
    if (P.max_out < 1 || P.max_out2 < P.max_out || P.blklen < 1) { ctx->err = "spdp_blk_find: max_out / max_out2 / blklen out of range"; return -1; }
    const blk_find::Genome G = {genome->codes, genome->chr_off, genome->n_chr};
    int out_cap = 4096;                                 // (a record that does not fit makes the round run again with room for it)
    std::vector<int> active(n), stop(n, 0), crit(n, 0), calls(n, 0);
    for (int i = 0; i < n; ++i) active[i] = i;
    std::vector<std::vector<blk_find::Locus>> found(n);
    std::vector<int32_t> rec;
    while (!active.empty()) {
        const int m = (int) active.size();
        // this round's queries, packed
        std::vector<int64_t> o(m + 1, 0);
        std::vector<int32_t> l(m), r(m), st(m);
        for (int k = 0; k < m; ++k) { const int q = active[k]; o[k + 1] = o[k] + (offs[q + 1] - offs[q]); l[k] = left[q]; r[k] = right[q]; st[k] = stop[q]; }
        std::vector<uint8_t> cd((size_t) o[m]);
        for (int k = 0; k < m; ++k) memcpy(cd.data() + o[k], codes + offs[active[k]], (size_t) (o[k + 1] - o[k]));
        for (;;) {
            rec.assign((size_t) m * out_cap, 0);
            if (spdp_blk_vote(ctx, ix, cd.data(), o.data(), l.data(), r.data(), st.data(), m, rec.data(), out_cap, nullptr)) return -1;
            bool cut = false;
            for (int k = 0; k < m && !cut; ++k) cut = (rec[(size_t) k * out_cap + 2] & SPDP_BLK_CUT) != 0;
            if (!cut || out_cap >= (1 << 20)) break;
            out_cap *= 8;
        }
        std::vector<int> verdict(m, 0);                 // > 0 loci, 0 go on, -1 ended, -2 record cut / table full
        std::atomic<int> next{0};
        auto work = [&] {
            blk_find::Searcher S;
            S.ix = &hv; S.P = &P; S.G = &G; S.M = model; S.intpen = sc->intpen; S.intpen_len = sc->intpen_len;
            S.gop = sc->gop; S.gep = sc->gep; S.lgop = sc->lgop; S.lgep = sc->lgep; S.codonk1 = sc->codonk1; S.chr_tab = hix->chr;
            for (int k; (k = next++) < m; ) {
                const int q = active[k];
                const int32_t* rc = rec.data() + (size_t) k * out_cap;
                if (!(rc[2] & SPDP_BLK_REACHED)) { verdict[k] = -1; continue; }         // findblock ended before this call
                if (rc[2] & (SPDP_BLK_CUT | SPDP_BLK_TABLE)) { verdict[k] = -2; continue; }
                int j = 3;
                const int32_t* mmct = rc + j + 4;
                j += 20;
                for (int d = 0; d < 4; ++d) j += 1 + 2 * rc[j];
                const int np = rc[j++];
                std::vector<blk_find::Pair> pairs(np);
                for (int i = 0; i < np; ++i, j += 9)
                    pairs[i] = {rc[j], rc[j + 1], 0, (uint32_t) rc[j + 2], (uint32_t) rc[j + 3], (uint32_t) rc[j + 4], (uint32_t) rc[j + 5],
                                (uint32_t) rc[j + 6], (uint32_t) rc[j + 7], rc[j + 8]};
                S.n_runs = rc[j++]; S.runs = rc + j;
                const blk_find::Query Q = {codes + offs[q], (int) (offs[q + 1] - offs[q]), left[q], right[q]};
                S.q = &Q; S.critjscr = crit[q];
                const int res = S.test_output(pairs, mmct, (rc[2] & SPDP_BLK_FORCED) != 0);
                crit[q] = S.critjscr;
                calls[q] = stop[q] + 1;
                verdict[k] = res;
                if (res > 0) found[q].assign(S.gener.begin(), S.gener.begin() + res);
            }
        };
        const int nt = std::max(1, std::min(spdp_host_cpus(), m));
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (std::thread& t : th) t.join();
        std::vector<int> again;
        for (int k = 0; k < m; ++k) {
            const int q = active[k];
            if (verdict[k] == 0) { ++stop[q]; again.push_back(q); }
            else if (verdict[k] == -2) { ctx->err = "spdp_blk_find: a vote record was cut or a hash table of the reference's size ran full"; return -1; }
            else if (verdict[k] < 0 && status) status[q] = -(calls[q] ? calls[q] : stop[q] + 1);
        }
        active.swap(again);
    }
    size_t nl = 0, nh = 0;
    for (int q = 0; q < n; ++q) for (const blk_find::Locus& g : found[q]) { ++nl; nh += g.jxt.size(); }
    *loci = (SpdpLocus*) malloc(sizeof(SpdpLocus) * std::max<size_t>(nl, 1));
    *hsps = (SpdpJuxt*) malloc(sizeof(SpdpJuxt) * std::max<size_t>(nh, 1));
    if (!*loci || !*hsps) { free(*loci); free(*hsps); *loci = nullptr; *hsps = nullptr; ctx->err = "spdp_blk_find: out of memory"; return -1; }
    size_t a = 0, b = 0;
    for (int q = 0; q < n; ++q) {
        if (status && !found[q].empty()) status[q] = calls[q];
        for (const blk_find::Locus& g : found[q]) {
            SpdpLocus& L = (*loci)[a++];
            L.query = q; L.chr = g.chr; L.rvs = g.rvs; L.base = g.base; L.len = g.len; L.left = g.left; L.right = g.right;
            L.jscr = g.jscr; L.n_hsp = (int32_t) g.jxt.size() - 1; L.hsp_off = (int64_t) b;
            for (const spdp_wl::Juxt& t : g.jxt) { (*hsps)[b++] = {t.jx, t.jy, t.jlen, t.nid, t.jscr}; }
        }
    }
    *n_loci = (int32_t) nl;
    return 0;
}
