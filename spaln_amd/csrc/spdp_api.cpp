// spdp_api.cpp -- host side of libspdp_hip.so: the extern "C" surface declared in
// include/spdp.h.  Packs problems into the HBM layout of spdp_dev.h, launches
// the sweeps of spdp_kernels.hip, and carries the reference's dispatch logic
// around them (Aln2s1::lspS_ng & co., src/fwd2s1.cc:1667-1897) -- see
// spdp_host.cpp for that part.  There is no CPU compute path here: without a
// HIP device spdp_create() fails.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

#include "../../include/spdp.h"
#include "spdp_dev.h"
#include "spdp_internal.h"

#define HIPCHK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

// ---- small helpers --------------------------------------------------------------
void spdp_stripe(const SpdpProblem* p, int sh, SpdpWindow* w)
{   // stripe(), src/aln2.cc:156-176 (cmode 3)
    if (sh < 0) {
        int shorter = std::min(p->a_right - p->a_left, p->b_right - p->b_left);
        sh = -sh * shorter / 100;
    }
    w->up = p->b_right - p->a_right;
    w->lw = p->b_left - p->a_left;
    if (w->up < w->lw) std::swap(w->up, w->lw);
    w->up += sh;
    w->lw -= sh;
    int q;
    if ((q = p->b_right - p->a_left) < w->up) w->up = q;
    if ((q = p->b_left - p->a_right) > w->lw) w->lw = q;
    w->width = w->up - w->lw + 3;
}

int64_t spdp_cells(const SpdpProblem* p, const SpdpWindow* w)
{   // cells visited by the reference loops, src/fwd2s1.cc:249-256
    int64_t c = 0;
    for (int m = p->a_left + 1; m <= p->a_right; ++m) {
        int n1 = std::max(m + w->lw, p->b_left);
        int n9 = std::min(m + w->up + 1, p->b_right);
        if (n9 > n1) c += n9 - n1;
    }
    return c;
}

// ---- context ----------------------------------------------------------------------
SpdpContext* spdp_create(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "spdp_create: no HIP device (this library has no CPU path)\n");
        return nullptr;
    }
    if (device < 0 || device >= ndev) return nullptr;
    SpdpContext* ctx = new SpdpContext();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return nullptr;
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, device);
    ctx->n_cu = prop.multiProcessorCount;
    ctx->name = prop.name;
    hipEventCreate(&ctx->ev0);
    hipEventCreate(&ctx->ev1);
    return ctx;
}

void spdp_destroy(SpdpContext* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* spdp_last_error(const SpdpContext* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int spdp_device_name(const SpdpContext* ctx, char* buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0) return -1;
    snprintf(buf, buflen, "%s (%d CUs)", ctx->name.c_str(), ctx->n_cu);
    return 0;
}

// ---- device batch ---------------------------------------------------------------------
static void to_dev_scoring(const SpdpScoring* sc, DevScoring* d)
{
    memset(d, 0, sizeof *d);
    d->mtx_dim = sc->mtx_dim;
    d->gop = sc->gop; d->gep = sc->gep;
    d->spj = sc->spj; d->llmt = sc->llmt; d->nquant = std::max(1, std::min(sc->nquant, SPDP_MAX_QUANT));
    d->local = sc->local ? 1 : 0;
    for (int j = 0; j < SPDP_MAX_QUANT; ++j) { d->qm_len[j] = sc->qm_len[j]; d->qm_pen[j] = sc->qm_pen[j]; }
    for (int i = 1; i < sc->mtx_dim && i < 32; ++i)
        for (int j = 1; j < sc->mtx_dim && j < 32; ++j)
            d->mtx[i * 32 + j] = sc->mtx[i * sc->mtx_dim + j];
}

static int stripe_blocks(const DevProblem& P, int s, bool forward)
{
    const int ml = P.a_left + s * SPDP_NELEM;
    const int j9 = std::min(SPDP_NELEM, P.a_right - ml);
    const int n_start = std::max(P.b_left, P.lw + ml);
    const int n9 = std::min(P.b_right, P.up + (ml + j9) + 1) + j9;
    const int len = std::max(0, n9 + (forward ? 1 : 0) - n_start);
    return (len + 15) >> 4;
}

void DevBatch::release()
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    void* ptrs[] = {d_sc, d_probs, d_a, d_cols, d_bnd, d_tb, d_imd, d_res, d_queue, d_skl, d_nskl,
                    d_cpos, d_ranges, d_scores};
    for (void* p : ptrs) if (p) hipFree(p);
    d_sc = nullptr; d_probs = nullptr; d_a = nullptr; d_cols = nullptr; d_bnd = nullptr; d_tb = nullptr;
    d_imd = nullptr; d_res = nullptr; d_queue = nullptr; d_skl = nullptr; d_nskl = nullptr;
    d_cpos = nullptr; d_ranges = nullptr; d_scores = nullptr;
}

// Packs and uploads a batch.  `wdws` may be null (then stripe(sh) per problem);
// n_im: per-problem number of UDH intermediates (null = none); flavour decides
// which work buffers are allocated.
int DevBatch::build(SpdpContext* c, const SpdpScoring* sc, const SpdpProblem* probs, int n,
                    const SpdpWindow* wdws, const int* n_im, int flav)
{
    ctx = c; flavour = flav; n_probs = n; local = sc->local ? 1 : 0;
    hipSetDevice(ctx->device);
    if (sc->noll != 2) { ctx->err = "only affine gaps (Noll = 2) are implemented"; return -1; }
    if (flav == 2 && sc->local) { ctx->err = "local UDH is not implemented"; return -1; }
    DevScoring hsc;
    to_dev_scoring(sc, &hsc);
    h_probs.assign(n, DevProblem());
    int64_t a_tot = 0, col_tot = 0, bnd_tot = 0, tb_tot = 0, imd_tot = 0;
    total_cells = 0; max_n_im = 0; max_skl = 0;
    for (int i = 0; i < n; ++i) {
        const SpdpProblem& p = probs[i];
        DevProblem& P = h_probs[i];
        SpdpWindow w;
        if (wdws) w = wdws[i]; else spdp_stripe(&p, sc->sh, &w);
        P.a_left = p.a_left; P.a_right = p.a_right; P.b_left = p.b_left; P.b_right = p.b_right;
        P.lw = w.lw; P.up = w.up; P.width = w.width;
        P.buf_size = w.width + 2 * SPDP_NELEM;
        P.flags = (p.a_exgl ? 1 : 0) | (p.a_exgr ? 2 : 0) | (p.b_exgl ? 4 : 0) | (p.b_exgr ? 8 : 0);
        P.n_im = n_im ? n_im[i] : 0;
        if (p.a_right > p.a_len || p.b_right > p.b_len || p.a_left < 0 || p.b_left < 0 ||
            p.a_right < p.a_left || p.b_right < p.b_left || w.width < 3) {
            ctx->err = "bad problem ranges"; return -1;
        }
        P.a_off = a_tot; a_tot += (p.a_len + 15) & ~15ll;
        P.col_off = col_tot; col_tot += (int64_t) (p.b_right - p.b_left + 1) + SPDP_COL_PAD;
        P.bnd_off = bnd_tot; bnd_tot += (int64_t) P.buf_size + SPDP_BND_PAD;
        P.tb_off = tb_tot;
        if (flav == 1) {
            const int ns = (p.a_right - p.a_left + SPDP_NELEM - 1) / SPDP_NELEM;
            for (int s = 0; s < ns; ++s) tb_tot += 256ll * stripe_blocks(P, s, true);
        }
        P.imd_off = imd_tot; imd_tot += (int64_t) P.n_im * 4 * w.width;
        P.cells = spdp_cells(&p, &w);
        total_cells += P.cells;
        max_n_im = std::max(max_n_im, P.n_im);
        max_skl = std::max(max_skl, (p.a_right - p.a_left) + (p.b_right - p.b_left) + 8);
    }
    // host staging
    std::vector<uint8_t> ha(std::max<int64_t>(a_tot, 16), 0);
    std::vector<int32_t> hc(2 * std::max<int64_t>(col_tot, 1), 0);
    const int ipen = sc->ipen;
    for (int i = 0; i < n; ++i) {
        const SpdpProblem& p = probs[i];
        const DevProblem& P = h_probs[i];
        memcpy(ha.data() + P.a_off, p.a, p.a_len);
        int32_t* c = hc.data() + 2 * P.col_off;
        for (int nn = p.b_left; nn <= p.b_right; ++nn, c += 2) {
            const uint16_t s5 = (uint16_t) (int16_t) (p.sig5[nn] + ipen);
            const uint16_t s3 = (uint16_t) p.sig3[nn];
            c[0] = sc->spj ? (int32_t) ((uint32_t) s5 | ((uint32_t) s3 << 16)) : 0;
            c[1] = (nn > p.b_left) ? p.b[nn - 1] : 0;
        }
    }
    const int bw = (flav == 2) ? 4 : 2;
    SpdpContext* ctx = c;
    HIPCHK(hipMalloc(&d_sc, sizeof(DevScoring)));
    HIPCHK(hipMalloc(&d_probs, sizeof(DevProblem) * std::max(n, 1)));
    HIPCHK(hipMalloc(&d_a, ha.size()));
    HIPCHK(hipMalloc(&d_cols, hc.size() * sizeof(int32_t)));
    HIPCHK(hipMalloc(&d_bnd, sizeof(int32_t) * bw * std::max<int64_t>(bnd_tot, 1)));
    HIPCHK(hipMalloc(&d_res, sizeof(DevResult) * std::max(n, 1)));
    HIPCHK(hipMalloc(&d_queue, sizeof(int)));
    if (flav == 1) {
        HIPCHK(hipMalloc(&d_tb, std::max<int64_t>(tb_tot, 16)));
        HIPCHK(hipMalloc(&d_skl, sizeof(int2) * (int64_t) max_skl * std::max(n, 1)));
        HIPCHK(hipMalloc(&d_nskl, sizeof(int) * std::max(n, 1)));
    }
    if (flav == 2) {
        HIPCHK(hipMalloc(&d_imd, sizeof(int32_t) * std::max<int64_t>(imd_tot, 1)));
        HIPCHK(hipMalloc(&d_cpos, sizeof(int32_t) * 10 * (max_n_im + 1) * std::max(n, 1)));
        HIPCHK(hipMalloc(&d_ranges, sizeof(int32_t) * 4 * std::max(n, 1)));
        HIPCHK(hipMalloc(&d_scores, sizeof(int32_t) * std::max(n, 1)));
    }
    HIPCHK(hipMemcpyAsync(d_sc, &hsc, sizeof hsc, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_probs, h_probs.data(), sizeof(DevProblem) * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_a, ha.data(), ha.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_cols, hc.data(), hc.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    tb_bytes = tb_tot;
    return 0;
}

// one sweep over the resident batch; kernel_ms = HIP-event time of the DP kernel alone
int DevBatch::run(float* kernel_ms)
{
    hipSetDevice(ctx->device);
    SweepArgs A;
    A.sc = (const DevScoring*) d_sc; A.probs = (const DevProblem*) d_probs; A.n_probs = n_probs;
    A.a_codes = (const uint8_t*) d_a; A.cols = (const int2*) d_cols; A.bnd = (int*) d_bnd;
    A.tb = (uint8_t*) d_tb; A.imd = (int*) d_imd; A.res = (DevResult*) d_res; A.queue = (int*) d_queue;
    A.dbg = nullptr;
    static int* dbg_host = nullptr;
    if (getenv("SPDP_DEBUG")) {
        if (!dbg_host) { hipHostMalloc((void**) &dbg_host, 64 * sizeof(int), hipHostMallocMapped); }
        memset(dbg_host, 0, 64 * sizeof(int));
        int* dp = nullptr; hipHostGetDevicePointer((void**) &dp, dbg_host, 0);
        A.dbg = dp;
    }
    HIPCHK(hipMemsetAsync(d_queue, 0, sizeof(int), ctx->stream));
    const int grid = std::max(1, std::min((n_probs + 3) / 4, ctx->n_cu * 8));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(spdp_launch_sweep(flavour, local, &A, grid, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (flavour == 1) {
        WalkArgs W;
        W.probs = A.probs; W.n_probs = n_probs; W.tb = (const uint8_t*) d_tb; W.res = (const DevResult*) d_res;
        W.skl = (int2*) d_skl; W.n_skl = (int*) d_nskl; W.skl_cap = max_skl;
        HIPCHK(spdp_launch_walk(&W, ctx->stream));
    }
    if (flavour == 2) {
        CposArgs C;
        C.probs = A.probs; C.n_probs = n_probs; C.imd = (const int*) d_imd; C.res = (const DevResult*) d_res;
        C.cpos = (int*) d_cpos; C.ranges = (int*) d_ranges; C.scores = (int*) d_scores;
        C.cpos_stride = 10 * (max_n_im + 1);
        HIPCHK(spdp_launch_cpos(&C, ctx->stream));
    }
    if (A.dbg) {
        for (int it = 0; it < 100; ++it) {
            if (hipStreamQuery(ctx->stream) == hipSuccess) break;
            usleep(100000);
            if (it % 10 == 9) {
                fprintf(stderr, "[spdp dbg] t=%.1fs:", 0.1 * (it + 1));
                for (int q = 0; q < 32; ++q) fprintf(stderr, " %d", dbg_host[q]);
                fprintf(stderr, "\n");
            }
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) HIPCHK(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    return 0;
}

int DevBatch::fetch_results(std::vector<DevResult>& out)
{
    out.resize(n_probs);
    HIPCHK(hipMemcpy(out.data(), d_res, sizeof(DevResult) * n_probs, hipMemcpyDeviceToHost));
    return 0;
}

// ---- engine-level entry points ----------------------------------------------------------
int spdp_wip_scoreonly(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                       int n_probs, int32_t* scores)
{
    if (!ctx) return -1;
    if (n_probs <= 0) return 0;
    DevBatch bt;
    if (bt.build(ctx, sc, probs, n_probs, nullptr, nullptr, 0)) return -1;
    if (bt.run(nullptr)) return -1;
    std::vector<DevResult> r;
    if (bt.fetch_results(r)) return -1;
    for (int i = 0; i < n_probs; ++i) scores[i] = r[i].score;
    return 0;
}

int spdp_homscore_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                    int n_probs, int32_t* scores)
{   // HomScoreS_ng for simd > 1: stripe(alprm.sh) + scoreonlyS1_wip, src/fwd2s1.cc:2696-2712
    return spdp_wip_scoreonly(ctx, sc, probs, n_probs, scores);
}

int spdp_wip_forward_w(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                       const SpdpWindow* wdws, int n_probs, SpdpAlignment* out)
{
    if (!ctx) return -1;
    for (int i = 0; i < n_probs; ++i) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; }
    if (n_probs <= 0) return 0;
    DevBatch bt;
    if (bt.build(ctx, sc, probs, n_probs, wdws, nullptr, 1)) return -1;
    if (bt.run(nullptr)) return -1;
    std::vector<DevResult> r;
    if (bt.fetch_results(r)) return -1;
    std::vector<int> nskl(n_probs);
    std::vector<SpdpSkl> skl((size_t) n_probs * bt.max_skl);
    HIPCHK(hipMemcpy(nskl.data(), bt.d_nskl, sizeof(int) * n_probs, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(skl.data(), bt.d_skl, sizeof(SpdpSkl) * skl.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n_probs; ++i) {
        out[i].score = r[i].score;
        if (nskl[i] < 0) { ctx->err = "traceback walk failed"; return -1; }
        out[i].n_skl = nskl[i];
        out[i].skl = (SpdpSkl*) malloc(sizeof(SpdpSkl) * std::max(1, nskl[i]));
        memcpy(out[i].skl, skl.data() + (size_t) i * bt.max_skl, sizeof(SpdpSkl) * nskl[i]);
    }
    return 0;
}

int spdp_wip_forward(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                     int n_probs, SpdpAlignment* out)
{
    return spdp_wip_forward_w(ctx, sc, probs, nullptr, n_probs, out);
}

int spdp_wip_udh_w(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                   const SpdpWindow* wdws, int n_probs, const int* n_im, int cpos_rows,
                   int32_t* scores, int32_t* cpos, int32_t* ranges)
{
    if (!ctx) return -1;
    if (n_probs <= 0) return 0;
    DevBatch bt;
    if (bt.build(ctx, sc, probs, n_probs, wdws, n_im, 2)) return -1;
    if (bt.run(nullptr)) return -1;
    const int stride = 10 * (bt.max_n_im + 1);
    std::vector<int32_t> hc((size_t) stride * n_probs);
    HIPCHK(hipMemcpy(hc.data(), bt.d_cpos, sizeof(int32_t) * hc.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ranges, bt.d_ranges, sizeof(int32_t) * 4 * n_probs, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(scores, bt.d_scores, sizeof(int32_t) * n_probs, hipMemcpyDeviceToHost));
    for (int i = 0; i < n_probs; ++i)
        memcpy(cpos + (size_t) i * 10 * cpos_rows, hc.data() + (size_t) i * stride,
               sizeof(int32_t) * 10 * std::min(cpos_rows, n_im[i] + 1));
    return 0;
}

int spdp_wip_udh(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                 int n_im, int32_t* scores, int32_t* cpos, int32_t* ranges)
{
    std::vector<int> v(std::max(n_probs, 1), n_im);
    return spdp_wip_udh_w(ctx, sc, probs, nullptr, n_probs, v.data(), n_im + 1, scores, cpos, ranges);
}

void spdp_free_alignments(SpdpAlignment* out, int n)
{
    for (int i = 0; i < n; ++i) { free(out[i].skl); out[i].skl = nullptr; out[i].n_skl = 0; }
}

// ---- resident batches -----------------------------------------------------------------------
struct SpdpBatch {
    DevBatch score;          // HomScoreS_ng leg
    SpdpScoring sc;
    std::vector<SpdpProblem> probs;
};

SpdpBatch* spdp_batch_upload(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n)
{
    if (!ctx || n <= 0) return nullptr;
    SpdpBatch* b = new SpdpBatch();
    b->sc = *sc;
    b->probs.assign(probs, probs + n);
    if (b->score.build(ctx, sc, probs, n, nullptr, nullptr, 0)) { delete b; return nullptr; }
    return b;
}

void spdp_batch_free(SpdpBatch* bt) { delete bt; }
int64_t spdp_batch_cells(const SpdpBatch* bt) { return bt ? bt->score.total_cells : 0; }

int spdp_batch_homscore(SpdpBatch* bt, int32_t* scores, float* kernel_ms)
{
    if (!bt) return -1;
    if (bt->score.run(kernel_ms)) return -1;
    if (scores) {
        std::vector<DevResult> r;
        if (bt->score.fetch_results(r)) return -1;
        for (size_t i = 0; i < r.size(); ++i) scores[i] = r[i].score;
    }
    return 0;
}
